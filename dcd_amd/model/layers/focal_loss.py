"""Penalty-reduced focal loss module (API of DGDE/model/layers/focal_loss.py:29-86) on the HIP kernel."""
from torch import nn

from dcd_amd import ops


class FocalLoss(nn.Module):
    def __init__(self, alpha=2, beta=4, cfg=None):
        super().__init__()
        self.alpha = alpha   # focusing exponent on hard / easy examples
        self.beta = beta     # down-weighting of negatives near a centre
        self.eps = 1e-10
        self.cls_num = cfg.DATASETS.MAX_CLASSES_NUM if cfg is not None else None

    def forward(self, prediction, target):
        """-> (loss summed over all elements, number of positives); one fused launch (+ one for backward)."""
        return ops.focal_loss(prediction, target, self.alpha, self.beta)
