"""BatchNorm2d with the residual add and ReLU that follow it fused in (csrc/norm.hip).

Same parameters, buffers and state-dict keys as `torch.nn.BatchNorm2d` (the reference's `BatchNorm`,
DGDE/model/backbone/dla_dcn.py:17-18), so checkpoints are interchangeable.  `forward(x, residual=None)` computes
`act(bn(x) + residual)` with `act` = ReLU when `fuse_relu` is set.  Where the reference has `[BN, ReLU]` inside an
`nn.Sequential`, the ReLU slot is kept as `nn.Identity()` so module indices (and therefore keys) do not move.

Device tensors always take the HIP kernels (and raise if libdcd_hip.so is missing).  CPU tensors (the host-logic tests)
run `torch.nn.BatchNorm2d.forward`, the very op the reference calls; synchronised statistics exist on the device only --
the world-size-2 gloo tests plug their own torch restatement into `BatchNorm2d.host_sync_stats` (tests/cpu_syncbn.py).
`sync_group`: set by `engine.trainer.wrap_distributed` when MODEL.USE_SYNC_BN; statistics are then all-reduced.
"""
import torch
from torch import nn
from torch.nn import functional as F

from dcd_amd import ops


class BatchNorm2d(nn.BatchNorm2d):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, fuse_relu=False):
        super().__init__(num_features, eps=eps, momentum=momentum)
        self.fuse_relu = fuse_relu
        self.sync_group = None

    def extra_repr(self):
        return super().extra_repr() + ", fuse_relu={}".format(self.fuse_relu)

    host_sync_stats = None      # test hook: f(x, weight, bias, eps, group) -> (y, mean, var); never set by product code

    def _stock(self, x, residual):
        if self.training and self.sync_group is not None:
            if self.host_sync_stats is None:
                raise RuntimeError("synchronised BatchNorm2d needs device tensors (the statistics exchange is a HIP + RCCL path)")
            y, mean, var = self.host_sync_stats(x, self.weight, self.bias, self.eps, self.sync_group)
            with torch.no_grad():
                n = x.numel() // x.shape[1] * torch.distributed.get_world_size(self.sync_group)
                self.running_mean.mul_(1 - self.momentum).add_(mean.float(), alpha=self.momentum)
                self.running_var.mul_(1 - self.momentum).add_((var * n / max(n - 1, 1)).float(), alpha=self.momentum)
                self.num_batches_tracked.add_(1)
        else:
            y = super().forward(x)
        if residual is not None:
            y = y + residual
        return F.relu(y) if self.fuse_relu else y

    def forward(self, x, residual=None):
        if not x.is_cuda or x.dtype != torch.float32:
            return self._stock(x, residual)
        if self.training:
            return ops.batch_norm_act(x, residual, self.weight, self.bias, self.running_mean, self.running_var,
                                      self.num_batches_tracked, self.momentum, self.eps, self.fuse_relu, self.sync_group)
        if torch.is_grad_enabled() and (x.requires_grad or (residual is not None and residual.requires_grad)):
            return self._stock(x, residual)      # gradients through frozen statistics: stock autograd
        return ops.batch_norm_act_eval(x, residual, self.weight, self.bias, self.running_mean, self.running_var, self.eps,
                                       self.fuse_relu)

    def forward_at(self, x, pos):
        """Training-mode BN (+ReLU) whose output is needed only at `pos` (B,N) linear pixel indices: returns (B,N,C).
        Statistics and running buffers are those of the dense op.  CPU tensors (tests) take the dense stock path + gather."""
        if x.is_cuda and x.dtype == torch.float32 and self.training and self.momentum is not None:
            return ops.batch_norm_act_at(x, pos, self.weight, self.bias, self.running_mean, self.running_var,
                                         self.num_batches_tracked, self.momentum, self.eps, self.fuse_relu, self.sync_group)
        y = self.forward(x)
        b, c = y.shape[0], y.shape[1]
        return y.flatten(2).gather(2, pos.long().unsqueeze(1).expand(b, c, pos.shape[1])).transpose(1, 2)
