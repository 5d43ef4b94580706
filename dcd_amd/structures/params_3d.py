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


def stack_field(targets, name):
    """`torch.stack([t.get_field(name) for t in targets])` -- without the copy when the per-image tensors are the consecutive
    slices of ONE batched tensor, which is how `dcd_amd.data.target_encoder.encode_targets` (device encoder) and `collate_fields`
    hand them out.  The reference stacks per-image fields on every step (detector_loss.py:106-146, predictor :172-176)."""
    ts = [t.get_field(name) for t in targets]
    t0 = ts[0]
    base = getattr(t0, "_base", None)
    if (base is not None and base.dim() == t0.dim() + 1 and base.shape[0] == len(ts) and base.is_contiguous()
            and tuple(base.shape[1:]) == tuple(t0.shape) and not base.requires_grad):
        step = base.stride(0) * base.element_size() if base.dim() > 0 and base.shape[0] > 1 else 0
        p0 = base.data_ptr()
        if all(getattr(t, "_base", None) is base and t.data_ptr() == p0 + i * step and t.shape == t0.shape for i, t in enumerate(ts)):
            return base
    return torch.stack(ts)


def collate_fields(targets):
    """Re-home every tensor field of a list of ParamsList in one batched tensor per field (what a collate function of the data
    loader does once per batch); the per-image fields become views, so later stacking is free (`stack_field`)."""
    if not targets:
        return targets
    for name in targets[0].fields():
        vals = [t.get_field(name) for t in targets]
        if all(torch.is_tensor(v) for v in vals) and all(v.shape == vals[0].shape and v.dtype == vals[0].dtype and
                                                           v.device == vals[0].device for v in vals):
            base = torch.stack(vals)
            for i, t in enumerate(targets):
                t.add_field(name, base[i])
    return targets
