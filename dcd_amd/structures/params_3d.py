"""Per-image label container with the reference's interface (DGDE/structures/params_3d.py:6-57)."""
import torch


class ParamsList:
    """A bag of named fields for one image.  Tensors (and anything with `.to`) follow `.to(device)`;
    strings and calibration objects are kept as they are; everything else goes through `torch.as_tensor`."""

    def __init__(self, image_size, is_train=True):
        self.size = image_size
        self.is_train = is_train
        self.extra_fields = {}

    def add_field(self, field, field_data):
        keep = isinstance(field_data, (torch.Tensor, str)) or hasattr(field_data, "project_image_to_rect")
        self.extra_fields[field] = field_data if keep else torch.as_tensor(field_data)

    def get_field(self, field):
        return self.extra_fields[field]

    def has_field(self, field):
        return field in self.extra_fields

    def fields(self):
        return list(self.extra_fields)

    def _copy_extra_fields(self, target):
        self.extra_fields.update(target.extra_fields)

    def to(self, device):
        moved = ParamsList(self.size, self.is_train)
        for name, value in self.extra_fields.items():
            moved.add_field(name, value.to(device) if hasattr(value, "to") else value)
        return moved

    def __len__(self):
        if not self.is_train:
            return 0
        return int(torch.count_nonzero(self.extra_fields["reg_mask"]))

    def __repr__(self):
        return "ParamsList(regress_number=%d, image_width=%s, image_height=%s)" % (len(self), self.size[0], self.size[1])
