"""Heat-map helpers with the reference's names (DGDE/model/layers/utils.py:22-145) on the HIP kernels."""
from dcd_amd import ops


class Converter_key2channel(object):
    """Maps a regression-head key to its channel slice in the concatenated output (utils.py:22-37)."""

    def __init__(self, keys, channels):
        self.keys = [k for group in keys for k in group]
        self.channels = [c for group in channels for c in group]

    def __call__(self, key):
        i = self.keys.index(key)
        start = sum(self.channels[:i])
        return slice(start, start + self.channels[i], 1)


def sigmoid_hm(hm_features):
    """In-place sigmoid then clamp to [1e-4, 1-1e-4] (utils.py:39-43)."""
    return hm_features.sigmoid_().clamp(min=1e-4, max=1 - 1e-4)


def nms_hm(heat_map, kernel=3, reso=1):
    """heat_map * (maxpool3x3(heat_map) == heat_map), one launch (utils.py:45-58)."""
    return ops.nms_hm(heat_map, kernel, reso)


def select_topk(heat_map, K=100, fuse_nms=False):
    """(scores, inds, clses, ys, xs), each (B,K) (utils.py:61-100); fuse_nms folds nms_hm into the same launch."""
    return ops.select_topk(heat_map, K, fuse_nms=fuse_nms)


def select_point_of_interest(batch, index, feature_maps):
    """(B,M,C) features at the given points / linear indices, gathered straight from NCHW (utils.py:120-145)."""
    return ops.select_point_of_interest(batch, index, feature_maps)


def _gather_feat(feat, ind):
    """feat (B,N,C), ind (B,K) -> (B,K,C) (utils.py:103-117)."""
    ind = ind.unsqueeze(-1).expand(ind.size(0), ind.size(1), feat.size(-1))
    return feat.gather(1, ind)
