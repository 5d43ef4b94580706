"""Per-image label container with the reference's interface (DGDE/structures/params_3d.py:6-57): `ParamsList(image_size,
is_train)`, `add_field / get_field / has_field / fields / to`, `len()` = number of regressed objects.  The attribute names
`size`, `is_train` and `extra_fields` are read by callers and kept."""
import torch


def _keep_as_is(value):
    """Tensors, strings and calibration objects (anything that can project image points to the rectified frame) are stored
    unchanged; numbers, lists and numpy arrays become tensors."""
    return torch.is_tensor(value) or isinstance(value, str) or hasattr(value, "project_image_to_rect")


class ParamsList:
    def __init__(self, image_size, is_train=True):
        self.size, self.is_train = image_size, is_train
        self.extra_fields = {}

    # ---- field access ------------------------------------------------------------------------------------------------
    def add_field(self, field, field_data):
        self.extra_fields[field] = field_data if _keep_as_is(field_data) else torch.as_tensor(field_data)

    def get_field(self, field):
        return self.extra_fields[field]

    def has_field(self, field):
        return field in self.extra_fields

    def fields(self):
        return [*self.extra_fields]

    def _copy_extra_fields(self, target):
        self.extra_fields.update(target.extra_fields)

    # ---- device placement: every field that knows `.to` follows, the rest is shared ---------------------------------
    def to(self, device):
        clone = type(self)(self.size, self.is_train)
        clone.extra_fields = {k: (v.to(device) if hasattr(v, "to") else v) for k, v in self.extra_fields.items()}
        return clone

    def __len__(self):
        return int(torch.count_nonzero(self.extra_fields["reg_mask"])) if self.is_train else 0

    def __repr__(self):
        width, height = self.size[0], self.size[1]
        return f"{type(self).__name__}(regress_number={len(self)}, image_width={width}, image_height={height})"
