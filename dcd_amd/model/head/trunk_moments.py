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


def patch_moments_gram(x):
    """x (B,C,H,W) -> S1 (9C,) fp64 and G (9C, 9C) fp64 over every pixel's zero-padded 3x3 patch [index c*9 + tap, the order of
    `weight.view(out, -1)`].  The image is cut into blocks of whole rows (with a one-row halo) BEFORE the patches are formed, so
    the patch matrix comes out block-major -- (B * blocks, 9C, pixels per block), contiguous -- and each block's Gram matrix is
    one batch entry of a single bmm; blocks are summed in fp64.  (First form of this path: 2 (9C)^2 HW flops per image; kept for
    A/B runs, DCD_TRUNK_GRAM=bmm.)"""
    B, C, H, W = x.shape
    r = _row_block(H, W)
    nb = H // r
    xb = F.pad(x, (0, 0, 1, 1)).unfold(2, r + 2, r)                    # (B, C, nb, W, r + 2): row blocks with their halo rows
    xb = xb.permute(0, 2, 1, 4, 3).reshape(B * nb, C, r + 2, W)
    ub = F.unfold(xb, 3, padding=(0, 1))                                # (B * nb, 9C, r * W)
    S1 = ub.sum(2).double().sum(0)
    return S1, _Gram.apply(ub)


# ---- autocorrelation form ---------------------------------------------------------------------------------------------------
# G[(c,t),(c',t')] = sum_p X[c, p+t-1] X[c', p+t'-1] depends on the taps only through d = t' - t, up to border terms:
#     G(t,t') = R_d - (terms of the one image row / column that the zero padding cuts off for tap t)
#     R_d[c,c'] = sum_u X[c,u] X[c',u+d]                 (X = 0 outside the image; d in [-2,2]^2, R_-d = R_d^T)
# 13 products of size C x C x BHW instead of (9C)^2 x BHW: 6x fewer flops, and no patch matrix at all -- the shifted operand is a
# pointer offset into the zero-padded image (dcd_sgemm_shifted).  The border terms need the three outermost rows / columns only.
HALF = [(dy, dx) for dy in range(0, 3) for dx in range(-2, 3) if dy > 0 or dx >= 0]           # 13 shifts, (0,0) first
ALL25 = [(dy, dx) for dy in range(-2, 3) for dx in range(-2, 3)]
_CONST = {}


def _const(key, build):
    if key not in _CONST:
        _CONST[key] = build()
    return _CONST[key]


class _ShiftCorr(torch.autograd.Function):
    """x (B,C,H,W) -> R (13,C,C) fp64 [HALF order], total (C,) fp64 = sum of x per channel, and copies of the border bands
    (top / bottom three rows, left / right three columns) through which autograd reaches the border terms.
    Device tensors: two launches of the shifted-view GEMM (forward: A = padded x, B = its 13 shifts, split-K partials summed in
    fp64; backward: dX = K Xshift over all 25 shifts with the gradient of `total` as the row bias).  Host tensors (the fp64
    host-logic tests): the same sums written with torch slices."""

    @staticmethod
    def forward(ctx, x, positions=None):
        B, C, H, W = x.shape
        Hp, Wp = H + 4, (W + 4 + 3) // 4 * 4
        Lp = Hp * Wp
        guard = 2 * Wp + 8
        buf = x.new_zeros(2 * guard + B * C * Lp)
        xp = buf[guard:guard + B * C * Lp].view(B, C, Hp, Wp)
        xp[:, :, 2:2 + H, 2:2 + W] = x
        ctx.geom = (B, C, H, W, Hp, Wp, Lp, guard)
        ctx.save_for_backward(buf)
        if x.is_cuda:
            from dcd_amd import _lib
            L = _lib.lib()
            dev = x.device
            off = _const(("fwd", C, Lp, Wp, str(dev)), lambda: torch.tensor(
                [c * Lp + dy * Wp + dx for (dy, dx) in HALF for c in range(C)], dtype=torch.int64, device=dev))
            nsplit = max(1, min(16, Lp // 1024))         # fp32 partial sums of ~2000 products, summed in fp64; 832 workgroups at 96x320
            part = torch.empty((B * nsplit, C, len(HALF) * C), dtype=torch.float32, device=dev)
            st = L.dcd_sgemm_shifted(_lib.stream_of(x), xp.data_ptr(), Lp, C * Lp, xp.data_ptr(), off.data_ptr(), C * Lp, 1, None,
                                     part.data_ptr(), len(HALF) * C, nsplit * C * len(HALF) * C, C * len(HALF) * C,
                                     C, len(HALF) * C, Lp, B, nsplit)
            _lib.check(st, "dcd_sgemm_shifted")
            R = part.double().sum(0).view(C, len(HALF), C).transpose(0, 1).contiguous()
        else:
            R = torch.stack([torch.einsum('bcyx,bkyx->ck', xp[:, :, 2:2 + H, 2:2 + W], xp[:, :, 2 + dy:2 + dy + H, 2 + dx:2 + dx + W])
                             for (dy, dx) in HALF]).double()
        total = x.sum(3).double().sum((0, 2))
        outs = (R, total, x[:, :, 0:3, :].clone(), x[:, :, H - 3:H, :].clone(), x[:, :, :, 0:3].clone(), x[:, :, :, W - 3:W].clone())
        ctx.pidx = None
        if positions is not None:
            # the 3x3 patches at listed cells (B, n) -> (B, 9C, n) [row c*9 + tap] read from the padded buffer: their gradient is
            # scatter-added into this node's dx in the backward (no zero-filled map, no extra full-size addition)
            pos = positions.long()
            n = pos.shape[1]
            taps = _const(("ptaps", Wp, str(x.device)), lambda: torch.tensor(
                [(dy + 1) * Wp + dx + 1 for dy in range(3) for dx in range(3)], device=x.device))
            base = (pos // W) * Wp + pos % W                                                     # (B, n): window's corner - (1, 1)
            idx = base.unsqueeze(1) + taps.view(1, 9, 1)                                        # (B, 9, n) into the padded plane
            idx = idx.reshape(B, 1, 9 * n)
            ctx.pidx = idx
            ctx.pbase = (base + Wp + 1).contiguous()                                            # top-left element of the 3x3 window
            outs = outs + (xp.reshape(B, C, Lp).gather(2, idx.expand(B, C, 9 * n)).reshape(B, 9 * C, n),)
        return outs

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dR, dtotal, dT, dBt, dL, dRr, dP=None):
        buf, = ctx.saved_tensors
        B, C, H, W, Hp, Wp, Lp, guard = ctx.geom
        xp = buf[guard:guard + B * C * Lp].view(B, C, Hp, Wp)
        dR = dR.to(buf.dtype)
        # dX[c,u] = sum_d sum_c' ( dR_d[c,c'] X[c',u+d] + dR_d[c',c] X[c',u-d] ) + dtotal[c]
        ia, ma, ib, mb = _const(("k1", str(buf.device)), lambda: (
            torch.tensor([HALF.index(e) if e in HALF else 0 for e in ALL25], device=buf.device),
            torch.tensor([float(e in HALF) for e in ALL25], device=buf.device).view(25, 1, 1),
            torch.tensor([HALF.index((-e[0], -e[1])) if (-e[0], -e[1]) in HALF else 0 for e in ALL25], device=buf.device),
            torch.tensor([float((-e[0], -e[1]) in HALF) for e in ALL25], device=buf.device).view(25, 1, 1)))
        K1 = (dR[ia] * ma.to(dR.dtype) + dR[ib].transpose(1, 2) * mb.to(dR.dtype)).transpose(0, 1).contiguous()   # (C, 25, C)
        if buf.is_cuda:
            from dcd_amd import _lib
            L = _lib.lib()
            dev = buf.device
            off = _const(("bwd", C, Lp, Wp, str(dev)), lambda: torch.tensor(
                [c * Lp + dy * Wp + dx for (dy, dx) in ALL25 for c in range(C)], dtype=torch.int64, device=dev))
            dxp = torch.empty((B, C, Hp, Wp), dtype=torch.float32, device=dev)
            bias = dtotal.to(torch.float32).contiguous()
            K1 = K1.view(C, 25 * C)
            st = L.dcd_sgemm_shifted(_lib.stream_of(buf), K1.data_ptr(), 25 * C, 0, xp.data_ptr(), off.data_ptr(), C * Lp, 0,
                                     bias.data_ptr(), dxp.data_ptr(), Lp, C * Lp, 0, C, Lp, 25 * C, B, 1)
            _lib.check(st, "dcd_sgemm_shifted")
        else:
            dxp = buf.new_zeros((B, C, Hp, Wp))
            dxi = dtotal.to(buf.dtype).view(1, C, 1, 1).expand(B, C, H, W).clone()
            for j, (dy, dx_) in enumerate(ALL25):
                dxi += torch.einsum('ck,bkyx->bcyx', K1[:, j, :], xp[:, :, 2 + dy:2 + dy + H, 2 + dx_:2 + dx_ + W])
            dxp[:, :, 2:2 + H, 2:2 + W] = dxi
        if ctx.pidx is not None and dP is not None:
            n9 = ctx.pidx.shape[2]
            if dxp.is_cuda:
                from dcd_amd import _lib
                gp = dP.to(dxp.dtype).contiguous()
                st = _lib.lib().dcd_patch_scatter_add(_lib.stream_of(dxp), gp.data_ptr(), ctx.pbase.data_ptr(), B, C, Lp, Wp, n9 // 9,
                                                      dxp.data_ptr())
                _lib.check(st, "dcd_patch_scatter_add")
            else:
                dxp.view(B, C, Lp).scatter_add_(2, ctx.pidx.expand(B, C, n9), dP.reshape(B, C, n9).to(dxp.dtype))
        dx = dxp[:, :, 2:2 + H, 2:2 + W].contiguous()
        dx[:, :, 0:3, :] += dT
        dx[:, :, H - 3:H, :] += dBt
        dx[:, :, :, 0:3] += dL
        dx[:, :, :, W - 3:W] += dRr
        return dx, None


def _border_tables(device):
    """Constant tables of the assembly: E[t,t'] = index of d = t' - t in ALL25; Mk[t, s] = coefficient of border term s
    (bottom row, top row, right column, left column, corners BR, BL, TR, TL) that tap t subtracts."""
    def build():
        E = torch.tensor([[ALL25.index((t2 // 3 - t1 // 3, t2 % 3 - t1 % 3)) for t2 in range(9)] for t1 in range(9)], device=device)
        Mk = torch.zeros(9, 8, dtype=torch.float64)
        for t in range(9):
            ty, tx = t // 3, t % 3
            # tap t reads u = p + t - 1: over all p it misses the LAST row when ty == 0, the FIRST row when ty == 2 (same for columns)
            Mk[t, 0] = ty == 0; Mk[t, 1] = ty == 2; Mk[t, 2] = tx == 0; Mk[t, 3] = tx == 2
            Mk[t, 4] = -(ty == 0 and tx == 0); Mk[t, 5] = -(ty == 0 and tx == 2)
            Mk[t, 6] = -(ty == 2 and tx == 0); Mk[t, 7] = -(ty == 2 and tx == 2)
        half_of = torch.tensor([HALF.index(d) if d in HALF else HALF.index((-d[0], -d[1])) for d in ALL25], device=device)
        flip = torch.tensor([d not in HALF for d in ALL25], device=device)
        return E, Mk.to(device), half_of, flip
    return _const(("tables", str(device)), build)


def patch_moments(x, positions=None):
    """S1 (9C,) and G (9C, 9C) in fp64 [index c*9 + tap] from the autocorrelation matrices of x (see above); with `positions`
    (B, n) also the 3x3 patches at those cells (B, 9C, n) from the same node, else None as third value."""
    mode = os.environ.get("DCD_TRUNK_GRAM", "auto")
    # auto: the autocorrelation form pays from ~100k pixels per rank on (its border terms are ~60 small launches: at one
    # 96x320 image per rank the step is launch-bound and the single bmm is 1 ms faster; at eight images it is 1.7 ms slower)
    small = x.is_cuda and x.shape[0] * x.shape[2] * x.shape[3] < 100000
    if mode == "bmm" or (mode == "auto" and small) or x.shape[2] < 5 or x.shape[3] < 5:
        return patch_moments_gram(x) + (None,)
    B, C, H, W = x.shape
    res = _ShiftCorr.apply(x, positions)
    R13, total, T, Bt, Lb, Rb = res[:6]
    P = res[6] if positions is not None else None
    E, Mk, half_of, flip = _border_tables(x.device)
    R25 = R13[half_of]
    R25 = torch.where(flip.view(25, 1, 1), R25.transpose(1, 2), R25)                           # R_-d = R_d^T
    T, Bt, Lb, Rb = T.double(), Bt.double(), Lb.double(), Rb.double()

    # Border terms, two bands per call: the bottom / right band is mirrored so that its strip line sits at index 0 like the
    # top / left band's (a mirrored axis turns d into -d along it: the result is mirrored back).
    rows = torch.stack((T, Bt.flip(2)))                                                         # (2,B,C,3,W), strip row = index 0
    nbr = F.pad(rows, (2, 2, 2, 0))                                                             # rows -2..2 around the strip row
    # (one index per tensor and `unbind` for pairs: every separate integer index is a zero-filled tensor, a copy and an
    # accumulation in the backward)
    strip_rows = rows[:, :, :, 0, :]                                                            # (2,B,C,W)
    row_t = torch.einsum('sbcx,sbkijx->sijck', strip_rows, nbr.unfold(4, W, 1))                 # (2,5,5,C,C): [band, dy, dx]
    cols = torch.stack((Lb, Rb.flip(3)))                                                        # (2,B,C,H,3), strip column = index 0
    nbc = F.pad(cols, (2, 0, 2, 2))
    col_t = torch.einsum('sbcy,sbkijy->sijck', cols[:, :, :, :, 0], nbc.unfold(3, H, 1))        # (2,5(dy),5(dx),C,C)
    # corners: the strip row's first / last pixel against its 5x5 neighbourhood
    cor_l = torch.einsum('sbc,sbkij->sijck', strip_rows[..., 0], nbr[..., 0:5])                 # column 0:    (top-left, bottom-left)
    cor_r = torch.einsum('sbc,sbkij->sijck', strip_rows[..., W - 1], nbr[..., W - 1:W + 4])     # column W-1:  (top-right, bottom-right)
    top_row, bot_row = row_t.unbind(0)
    left_col, right_col = col_t.unbind(0)
    cor_l_top, cor_l_bot = cor_l.unbind(0)
    cor_r_top, cor_r_bot = cor_r.unbind(0)
    strips = torch.stack((bot_row.flip(0), top_row, right_col.flip(1), left_col, cor_r_bot.flip(0), cor_l_bot.flip(0), cor_r_top,
                          cor_l_top)).reshape(8, 25, C, C)
    corr = (Mk @ strips.view(8, -1)).view(9, 25, C, C)
    Gt = (R25.unsqueeze(0) - corr)[torch.arange(9, device=x.device).view(9, 1), E]            # (9, 9, C, C): [t, t', c, c']
    G = Gt.permute(2, 0, 3, 1).reshape(9 * C, 9 * C)
    # S1[c,t] = sum over u in the image minus the row / column tap t never reads (+ the corner counted twice)
    bot_line, top_line = Bt[:, :, 2, :].sum(0), T[:, :, 0, :].sum(0)                            # (C, W): outermost rows, summed over the batch
    sums = torch.stack((bot_line.sum(1), top_line.sum(1), Rb[:, :, :, 2].sum((0, 2)), Lb[:, :, :, 0].sum((0, 2)),
                        bot_line[:, W - 1], bot_line[:, 0], top_line[:, W - 1], top_line[:, 0]))                            # (8, C)
    S1 = (total.view(1, C) - Mk @ sums).t().reshape(9 * C)
    return S1, G, P


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


def trunks_at(x, trunks, centers, extra=None, stacked=False):
    """BN + ReLU outputs of every trunk at `centers` (B, M) linear pixel indices -> list of (B, M, Cout); `extra` = (trunk index,
    positions (B, Ke)) appends that trunk's outputs at further positions (the border cells of the edge-fusion branch).  Updates the
    BatchNorm running estimates like a dense training forward.  stacked: return (all trunks at the centres as ONE (T, B, M, Cout)
    tensor, the extra trunk's outputs at its extra positions (B, Ke, Cout) or None) instead of the list."""
    B, C, H, W = x.shape
    T = len(trunks)
    pos_all = centers.long() if extra is None else torch.cat((centers.long(), extra[1].long()), dim=1)
    S1, G, P_all = patch_moments(x, pos_all if os.environ.get("DCD_TRUNK_PATCH_NODE", "1") != "0" else None)   # 0: separate gather (A/B)
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
    sum_y, sum_yy = sums.unbind(-1)
    mean = sum_y / n
    var = (sum_yy / n - mean * mean).clamp_min(0)
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
    if P_all is not None:                                      # gathered inside the correlation node (its backward scatters into dx)
        M0 = centers.shape[1]

        def patches(pos, first):
            return P_all[:, :, :M0] if first else P_all[:, :, M0:]
    else:
        xp = F.pad(x, (1, 1, 1, 1)).flatten(2)                                                 # (B, C, (H+2)(W+2))
        taps = _taps(W, x.device)

        def patches(pos, first):                                                               # (B, n) -> (B, 9C, n), row c*9 + tap
            pos = pos.long()
            base = (pos // W) * (W + 2) + pos % W                                              # top-left of the 3x3 window in xp
            n = pos.shape[1]
            idx = (base.unsqueeze(1) + taps.view(1, 9, 1)).reshape(B, 1, 9 * n).expand(B, C, 9 * n)
            return xp.gather(2, idx).reshape(B, K, n)                                          # (B, C, 9, n) -> (B, 9C, n)
    Xc = patches(centers, True)                                                                # (B, 9C, M)
    y = torch.einsum('bkm,tok->tbmo', Xc, Wall)
    at_centres = torch.relu(y * scale.view(T, 1, 1, -1) + shift.view(T, 1, 1, -1))
    at_extra = None
    if extra is not None:
        i, pos = extra
        Xe = patches(pos, False)
        ye = torch.einsum('bkm,ok->bmo', Xe, Wall[i])
        at_extra = torch.relu(ye * scale[i].view(1, 1, -1) + shift[i].view(1, 1, -1))
    if stacked:
        return at_centres, at_extra
    out = list(at_centres.unbind(0))
    if at_extra is not None:
        out[extra[0]] = torch.cat((out[extra[0]], at_extra), dim=1)
    return out


def frozen(trunks, x):
    """The eleven-trunk shape with the BatchNorm layers in EVAL mode (running statistics: the --generate_for_GMW pass freezes
    them, DGDE/engine/trainer.py:62-67, and inference uses them): `trunks_at_frozen` then needs no statistics at all."""
    from dcd_amd.model.layers.norm import BatchNorm2d
    if x.dim() != 4 or x.dtype not in (torch.float32, torch.float64):
        return False
    for t in trunks:
        conv, bn = t[0], t[1]
        if not (isinstance(bn, BatchNorm2d) and bn.fuse_relu and not bn.training and bn.track_running_stats
                and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1)
                and conv.groups == 1 and conv.bias is None and conv.in_channels == x.shape[1] and isinstance(t[2], torch.nn.Identity)):
            return False
    return True


def trunks_at_frozen(x, trunks, centers, extra=None):
    """`trunks_at(..., stacked=True)` for BatchNorm layers that normalise with their RUNNING statistics: conv3x3 + BN + ReLU of every
    trunk at `centers` (B, M) linear pixel indices only -- the 3x3 patches gathered from the zero-padded input, one einsum against the
    stacked weights -- instead of eleven dense convolutions whose outputs are read at a few dozen cells (config 4: 16 images x 50
    detections of 30 720 cells).  Returns ((T, B, M, Cout), the `extra` trunk at its positions (B, Ke, Cout) or None)."""
    B, C, H, W = x.shape
    T, K = len(trunks), 9 * C
    Wall = torch.stack([t[0].weight.reshape(t[0].out_channels, K) for t in trunks])            # (T, O, 9C)
    rm = torch.stack([t[1].running_mean for t in trunks]).to(x.dtype)
    rv = torch.stack([t[1].running_var for t in trunks]).to(x.dtype)
    gamma = torch.stack([t[1].weight for t in trunks]).to(x.dtype)
    beta = torch.stack([t[1].bias for t in trunks]).to(x.dtype)
    scale = gamma * torch.rsqrt(rv + trunks[0][1].eps)
    shift = beta - rm * scale
    xp = F.pad(x, (1, 1, 1, 1)).flatten(2)                                                     # (B, C, (H+2)(W+2))
    taps = _taps(W, x.device)

    def patches(pos):                                                                          # (B, n) -> (B, 9C, n), row c*9 + tap
        pos = pos.long()
        base = (pos // W) * (W + 2) + pos % W
        n = pos.shape[1]
        idx = (base.unsqueeze(1) + taps.view(1, 9, 1)).reshape(B, 1, 9 * n).expand(B, C, 9 * n)
        return xp.gather(2, idx).reshape(B, K, n)
    y = torch.einsum('bkm,tok->tbmo', patches(centers), Wall)
    at_centres = torch.relu(y * scale.view(T, 1, 1, -1) + shift.view(T, 1, 1, -1))
    at_extra = None
    if extra is not None:
        i, pos = extra
        ye = torch.einsum('bkm,ok->bmo', patches(pos), Wall[i])
        at_extra = torch.relu(ye * scale[i].view(1, 1, -1) + shift[i].view(1, 1, -1))
    return at_centres, at_extra
