"""Test infrastructure: SyncBN statistics on CPU tensors over gloo (torch's SyncBatchNorm is GPU-only and the product's
synchronised BatchNorm2d is a HIP + RCCL path).  The same two-phase scheme as csrc/norm.hip, written with torch ops;
`install()` plugs it into `BatchNorm2d.host_sync_stats` for the world-size-2 CPU tests."""
import torch


class _SyncStatsCPU(torch.autograd.Function):
    """SyncBN on CPU tensors over gloo (torch's SyncBatchNorm is GPU-only): the same two-phase scheme as the HIP path,
    written with torch ops.  Test infrastructure for the world-size-2 CPU tests."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, group):
        import torch.distributed as dist
        dims = [0] + list(range(2, x.dim()))
        xd = x.double()
        stats = torch.stack((xd.sum(dims), (xd * xd).sum(dims)), 1)
        dist.all_reduce(stats, group=group)
        count = x.numel() // x.shape[1] * dist.get_world_size(group)
        mean = stats[:, 0] / count
        var = (stats[:, 1] / count - mean * mean).clamp_min(0)
        invstd = (var + eps).rsqrt()
        shape = [1, -1] + [1] * (x.dim() - 2)
        xhat = ((xd - mean.view(shape)) * invstd.view(shape)).float()
        ctx.save_for_backward(xhat, weight, invstd.float())
        ctx.group, ctx.count = group, count
        ctx.mark_non_differentiable(mean, var)
        return xhat * weight.view(shape) + bias.view(shape), mean, var

    @staticmethod
    def backward(ctx, gy, _gm, _gv):
        import torch.distributed as dist
        xhat, weight, invstd = ctx.saved_tensors
        dims = [0] + list(range(2, gy.dim()))
        shape = [1, -1] + [1] * (gy.dim() - 2)
        sums = torch.stack((gy.double().sum(dims), (gy.double() * xhat.double()).sum(dims)), 1)
        gw, gb = sums[:, 1].float(), sums[:, 0].float()
        dist.all_reduce(sums, group=ctx.group)
        m = (sums / ctx.count).float()
        gx = (gy - m[:, 0].view(shape) - xhat * m[:, 1].view(shape)) * (invstd * weight).view(shape)
        return gx, gw, gb, None, None


def install():
    from dcd_amd.model.layers.norm import BatchNorm2d
    BatchNorm2d.host_sync_stats = staticmethod(_SyncStatsCPU.apply)
