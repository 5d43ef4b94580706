"""IoU-family loss on (l, t, r, b) distances (API of DGDE/model/layers/iou_loss.py:7-49).

'giou' (the DGDE configuration) runs the fused HIP kernel; 'iou' / 'linear_iou' derive from its IoU output.
`get_iou_3d` (iou_loss.py:99-136) is a logging-only metric built on shapely polygons in the reference; here it is
one HIP launch (convex-polygon clip per object pair), so the train step needs no CPU round trip.
"""
import torch
from torch import nn

from dcd_amd import ops


class IOULoss(nn.Module):
    def __init__(self, loss_type="iou"):
        super().__init__()
        if loss_type not in ("iou", "linear_iou", "giou"):
            raise NotImplementedError(loss_type)
        self.loss_type = loss_type

    def forward(self, pred, target, weight=None):
        losses, ious = ops.giou_loss(pred, target)
        if self.loss_type == "giou":
            return losses, ious
        # the two plain-IoU variants are not used by DGDE.yaml; computed from the same quantities with torch ops
        p, t = pred.float(), target.float()
        inter = (torch.min(p[:, 0], t[:, 0]) + torch.min(p[:, 2], t[:, 2])) * \
                (torch.min(p[:, 3], t[:, 3]) + torch.min(p[:, 1], t[:, 1]))
        union = (t[:, 0] + t[:, 2]) * (t[:, 1] + t[:, 3]) + (p[:, 0] + p[:, 2]) * (p[:, 1] + p[:, 3]) - inter
        iou = (inter + 1.0) / (union + 1.0)
        return (-torch.log(iou) if self.loss_type == "iou" else 1 - iou), iou


def get_iou_3d(pred_corners, target_corners):
    """(N,8,3) corner sets -> (N) 3-D IoU: BEV polygon overlap x height overlap (iou_loss.py:99-136).
    Logging only (called under no_grad); one HIP launch instead of a shapely loop on the host."""
    return ops.iou_3d(pred_corners, target_corners)
