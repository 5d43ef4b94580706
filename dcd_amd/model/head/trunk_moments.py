"""The eleven regression trunks of the head (3x3 conv 64->256, BatchNorm, ReLU; DGDE/model/head/detector_predictor.py:78-101)
evaluated in TRAINING without ever forming their dense outputs.

The training loss reads the regression heads at <= MAX_OBJECTS centres per image (plus the border cells for one head,
detector_loss.py:231-233), so round 1 already evaluated BN + ReLU + the 1x1 heads at those positions only -- but the 3x3
convolutions stayed dense, because BatchNorm needs the batch statistics of their outputs over every pixel: 11 x 72.5 GFLOP
forward, the same again for the input gradient and for the weight gradient (the largest block of the step: 106.9 GFLOP per
image forward).  Those statistics do not need the outputs.  With P_p the 9C-vector of the zero-padded
3x3 patch of the shared input x at pixel p,  y_o(p) = w_o . P_p,  hence

    sum_p y_o(p)   = w_o . S1,        S1 = sum_p P_p              (9C values)
    sum_p y_o(p)^2 = w_o^T G w_o,     G  = sum_p P_p P_p^T        (9C x 9C Gram matrix, ONE for all eleven trunks)

and the trunk outputs themselves are needed at the listed positions only: a (positions x 9C) x (9C x 256) product per trunk.
G costs 2 (9C)^2 HW flops per image (20 GFLOP at C = 64, 96x320) instead of 11 x 9 GFLOP x 3 passes, and its backward is one
more product of the same size (d/dP_p = 2 Q P_p, folded back onto x); everything else is tiny.  Same mathematics as the
reference's dense conv + BatchNorm2d (biased batch variance, running estimates with momentum and the unbiased variance), other
order of summation: products in fp32 like the convolution's, partial Gram matrices of <= 4096 pixels summed in fp64, the
256-channel statistics in fp64.  With MODEL.USE_SYNC_BN the per-channel sums (not G) are all-reduced, exactly like SyncBN.

Pure torch ops (GEMMs on the matrix pipe via the BLAS library, differentiable end to end), so the host-logic tests run it on the
CPU against the reference fixture.  `DCD_TRUNK_MOMENTS=0` restores the dense trunks (A/B timing)."""
import os

import torch
from torch.nn import functional as F

ENABLED = os.environ.get("DCD_TRUNK_MOMENTS", "1") != "0"


class _Gram(torch.autograd.Function):
    """G = sum over batches of  U_b U_b^T  (fp32 products, fp64 sum over the batches); backward = (Q + Q^T) U_b."""

    @staticmethod
    def forward(ctx, ub):
        ctx.save_for_backward(ub)
        return torch.bmm(ub, ub.transpose(1, 2)).double().sum(0)

    @staticmethod
    def backward(ctx, dg):
        ub, = ctx.saved_tensors
        q = (dg + dg.t()).to(ub.dtype)
        return torch.matmul(q, ub)


def _row_block(H, W, limit=4096):
    """Rows per block: the largest divisor of H whose block (rows x W pixels) stays within `limit` pixels."""
    best = 1
    for r in range(1, H + 1):
        if H % r == 0 and r * W <= limit:
            best = r
    return best


def patch_moments(x):
    """x (B,C,H,W) -> S1 (9C,) fp64 and G (9C, 9C) fp64 over every pixel's zero-padded 3x3 patch [index c*9 + tap, the order of
    `weight.view(out, -1)`].  The image is cut into blocks of whole rows (with a one-row halo) BEFORE the patches are formed, so
    the patch matrix comes out block-major -- (B * blocks, 9C, pixels per block), contiguous -- and each block's Gram matrix is
    one batch entry of a single bmm; blocks are summed in fp64."""
    B, C, H, W = x.shape
    r = _row_block(H, W)
    nb = H // r
    xb = F.pad(x, (0, 0, 1, 1)).unfold(2, r + 2, r)                    # (B, C, nb, W, r + 2): row blocks with their halo rows
    xb = xb.permute(0, 2, 1, 4, 3).reshape(B * nb, C, r + 2, W)
    ub = F.unfold(xb, 3, padding=(0, 1))                                # (B * nb, 9C, r * W)
    S1 = ub.sum(2).double().sum(0)
    return S1, _Gram.apply(ub)


def usable(trunks, x):
    """The eleven-trunk shape this path implements: Sequential(3x3 conv stride 1 pad 1 without bias, fused-ReLU BatchNorm2d, Identity)."""
    from dcd_amd.model.layers.norm import BatchNorm2d
    if not ENABLED or x.dim() != 4 or x.dtype not in (torch.float32, torch.float64):
        return False
    for t in trunks:
        conv, bn = t[0], t[1]
        if not (isinstance(bn, BatchNorm2d) and bn.fuse_relu and bn.training and bn.momentum is not None and bn.track_running_stats
                and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1)
                and conv.groups == 1 and conv.bias is None and conv.in_channels == x.shape[1]):
            return False
    return True


_TAPS = {}


def _taps(W, device):
    """Offsets of the nine taps inside the padded plane, cached per (width, device): no host-to-device copy per step."""
    key = (W, str(device))
    if key not in _TAPS:
        _TAPS[key] = torch.tensor([dy * (W + 2) + dx for dy in range(3) for dx in range(3)], device=device)
    return _TAPS[key]


def trunks_at(x, trunks, centers, extra=None):
    """BN + ReLU outputs of every trunk at `centers` (B, M) linear pixel indices -> list of (B, M, Cout); `extra` = (trunk index,
    positions (B, Ke)) appends that trunk's outputs at further positions (the border cells of the edge-fusion branch).  Updates the
    BatchNorm running estimates like a dense training forward."""
    B, C, H, W = x.shape
    T = len(trunks)
    S1, G = patch_moments(x)
    K = S1.shape[0]
    Wall = torch.stack([t[0].weight.reshape(t[0].out_channels, K) for t in trunks])            # (T, O, 9C)
    Wd = Wall.double()
    sums = torch.stack((Wd @ S1, ((Wd @ G) * Wd).sum(-1)), dim=-1)                              # (T, O, 2): sum y, sum y^2
    n = B * H * W
    group = trunks[0][1].sync_group
    if group is not None:
        import torch.distributed as dist
        import torch.distributed.nn.functional as distf
        sums = distf.all_reduce(sums, group=group)
        n = n * dist.get_world_size(group)
    mean = sums[..., 0] / n
    var = (sums[..., 1] / n - mean * mean).clamp_min(0)
    gamma = torch.stack([t[1].weight for t in trunks]).double()
    beta = torch.stack([t[1].bias for t in trunks]).double()
    eps = trunks[0][1].eps
    scale = gamma * torch.rsqrt(var + eps)                                                     # (T, O) fp64
    shift = beta - mean * scale
    with torch.no_grad():
        unbiased = var * (n / max(n - 1, 1))
        bns = [t[1] for t in trunks]
        mom = bns[0].momentum
        if all(bn.momentum == mom and bn.running_mean.dtype == bns[0].running_mean.dtype for bn in bns):
            # one multi-tensor launch per operation instead of five small kernels per trunk
            rm, rv = [bn.running_mean for bn in bns], [bn.running_var for bn in bns]
            dt = rm[0].dtype
            torch._foreach_mul_(rm, 1 - mom)
            torch._foreach_add_(rm, list(mean.to(dt).unbind(0)), alpha=mom)
            torch._foreach_mul_(rv, 1 - mom)
            torch._foreach_add_(rv, list(unbiased.to(dt).unbind(0)), alpha=mom)
            torch._foreach_add_([bn.num_batches_tracked for bn in bns], 1)
        else:
            for i, bn in enumerate(bns):
                bn.running_mean.mul_(1 - bn.momentum).add_(mean[i].to(bn.running_mean.dtype), alpha=bn.momentum)
                bn.running_var.mul_(1 - bn.momentum).add_(unbiased[i].to(bn.running_var.dtype), alpha=bn.momentum)
                bn.num_batches_tracked.add_(1)
    scale, shift = scale.to(x.dtype), shift.to(x.dtype)
    # patches at the listed positions straight from the (zero-padded) input, not from U: their backward is then a scatter of a
    # few thousand values into dx instead of a zero-filled (B, 9C, HW) tensor, a scatter into it and one more full-size addition
    xp = F.pad(x, (1, 1, 1, 1)).flatten(2)                                                     # (B, C, (H+2)(W+2))
    taps = _taps(W, x.device)

    def patches(pos):                                                                          # (B, n) -> (B, 9C, n), row c*9 + tap
        pos = pos.long()
        base = (pos // W) * (W + 2) + pos % W                                                  # top-left of the 3x3 window in xp
        n = pos.shape[1]
        idx = (base.unsqueeze(1) + taps.view(1, 9, 1)).reshape(B, 1, 9 * n).expand(B, C, 9 * n)
        return xp.gather(2, idx).reshape(B, K, n)                                              # (B, C, 9, n) -> (B, 9C, n)
    Xc = patches(centers)                                                                      # (B, 9C, M)
    y = torch.einsum('bkm,tok->tbmo', Xc, Wall)
    out = list(torch.relu(y * scale.view(T, 1, 1, -1) + shift.view(T, 1, 1, -1)).unbind(0))
    if extra is not None:
        i, pos = extra
        Xe = patches(pos)
        ye = torch.einsum('bkm,ok->bmo', Xe, Wall[i])
        out[i] = torch.cat((out[i], torch.relu(ye * scale[i].view(1, 1, -1) + shift[i].view(1, 1, -1))), dim=1)
    return out
