"""Batched-image holder with the reference's interface (DGDE/structures/image_list.py:6-70): `ImageList(tensors, image_sizes)`
with `.to(...)`, and `to_image_list(x, size_divisible=0)` that accepts what the reference accepts."""
import torch


class ImageList:
    """`tensors`: (N, C, H, W) batch, possibly zero padded; `image_sizes`: the (h, w) of every image before padding."""

    def __init__(self, tensors, image_sizes):
        self.tensors, self.image_sizes = tensors, image_sizes

    def to(self, *args, **kwargs):
        return ImageList(self.tensors.to(*args, **kwargs), self.image_sizes)


def _round_up(v, multiple):
    return v if multiple <= 0 else -(-v // multiple) * multiple


def _stack_padded(images, size_divisible):
    """Zero-pad (C, Hi, Wi) images to a common (C, H, W) -- H, W rounded up to `size_divisible` -- and stack them."""
    shapes = torch.tensor([tuple(img.shape) for img in images])
    c, h, w = (int(v) for v in shapes.max(dim=0).values)
    h, w = _round_up(h, size_divisible), _round_up(w, size_divisible)
    batch = images[0].new_zeros((len(images), c, h, w))
    for slot, img in zip(batch, images):
        ci, hi, wi = img.shape
        slot[:ci, :hi, :wi] = img
    return ImageList(batch, [img.shape[-2:] for img in images])


def to_image_list(tensors, size_divisible=0):
    """ImageList -> itself; a (C,H,W) / (N,C,H,W) tensor -> wrapped as is (split and padded when `size_divisible` > 0);
    a list / tuple of (C,Hi,Wi) tensors -> zero padded to the largest size."""
    if isinstance(tensors, ImageList):
        return tensors
    if torch.is_tensor(tensors):
        if size_divisible > 0:
            return _stack_padded([tensors], size_divisible)
        batch = tensors.unsqueeze(0) if tensors.dim() == 3 else tensors
        if batch.dim() != 4:
            raise ValueError("expected a 3-D or 4-D image tensor")
        return ImageList(batch, [img.shape[-2:] for img in batch])
    if isinstance(tensors, (tuple, list)):
        return _stack_padded(list(tensors), size_divisible)
    raise TypeError("Unsupported type for to_image_list: {}".format(type(tensors)))
