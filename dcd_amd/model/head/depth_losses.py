"""Depth-related regression losses (DGDE/model/head/depth_losses.py:30-105)."""
import torch
import torch.nn.functional as F
from torch import nn


class RegWeightedL1Loss(nn.Module):
    """L1 over (x,y) of each keypoint, re-weighted by object depth: d<5 -> 0.01*d, else log10(d-4)+0.1
    (depth_losses.py:50-67).  pred/target (k,n,2), dep (k) -> (k,n)."""

    def forward(self, pred, target, dep):
        dep = dep.detach()
        w = torch.where(dep < 5, dep * 0.01, torch.log10(torch.clamp(dep, min=5) - 4) + 0.1)
        return F.l1_loss(pred, target, reduction='none').sum(dim=-1) * w.unsqueeze(-1)


class Berhu_Loss(nn.Module):
    """Reverse Huber: L1 below c = 0.2*max|diff|, scaled L2 above (depth_losses.py:30-48; the reference version
    drops into pdb first and is not used by DGDE.yaml)."""

    def __init__(self):
        super().__init__()
        self.c = 0.2

    def forward(self, prediction, target, reduction='none'):
        differ = (prediction - target).abs()
        c = torch.clamp(differ.max() * self.c, min=1e-4)
        return torch.where(differ <= c, differ, (differ ** 2 / c + c) / 2)


class Inverse_Sigmoid_Loss(nn.Module):
    def forward(self, prediction, target, weight=None, reduction='none'):
        loss = F.l1_loss(1 / torch.sigmoid(target) - 1, target, reduction='none')   # sic: the reference transforms `target`
        return loss if weight is None else loss * weight


class Log_L1_Loss(nn.Module):
    def forward(self, prediction, target, weight=None, reduction='none'):
        loss = F.l1_loss(torch.log(prediction), torch.log(target), reduction='none')
        return loss if weight is None else loss * weight
