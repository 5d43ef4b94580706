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
# Summed over the EXTENDED pixel range p in [-1, H] x [-1, W] (one ring around the image), every in-image u = p + t - 1 occurs
# exactly once for every tap t, so
#     G_ext[(c,t),(c',t')] = sum_u X[c,u] X[c',u+d] = R_d[c,c'],   d = t' - t in [-2,2]^2,  R_-d = R_d^T   (X = 0 outside the image)
# depends on the taps only through d: 13 products of size C x C x BHW instead of (9C)^2 x BHW -- 6x fewer flops, and no patch
# matrix at all: the shifted operand is a pointer offset into the zero-padded image (dcd_sgemm_shifted).  The true sums run over
# the image's own pixels, i.e. G = G_ext - sum over the ring's 2 (W + 2) + 2 H pixels of P_p P_p^T (their zero-padded patches only
# touch the two outermost rows / columns), and S1 = total - sum over the ring of P_p.  The ring's patches are gathered by the
# correlation node (like the patches at the listed positions) and their Gram matrix is one small batched product.
# (Rounds 2-3 wrote the same correction as ~60 small tensor operations on the border bands -- row / column / corner terms per
# tap pair -- and ~100 more in their backward: the longest run of tiny launches of the step.)
HALF = [(dy, dx) for dy in range(0, 3) for dx in range(-2, 3) if dy > 0 or dx >= 0]           # 13 shifts, (0,0) first
ALL25 = [(dy, dx) for dy in range(-2, 3) for dx in range(-2, 3)]
_CONST = {}


def _const(key, build):
    if key not in _CONST:
        _CONST[key] = build()
    return _CONST[key]


def _ring_bases(H, W, Wp, device):
    """Padded-plane index of (window corner - (1, 1)) for the ring pixels p = (py, px), py in {-1, H} or px in {-1, W}: the
    3x3 window of p covers padded rows py + 1 .. py + 3, columns px + 1 .. px + 3 (image (y, x) sits at padded (y + 2, x + 2))."""
    def build():
        ring = [(-1, px) for px in range(-1, W + 1)] + [(H, px) for px in range(-1, W + 1)]
        ring += [(py, -1) for py in range(H)] + [(py, W) for py in range(H)]
        return torch.tensor([py * Wp + px for py, px in ring], dtype=torch.int64, device=device)
    return _const(("ring", H, W, Wp, str(device)), build)


class _ShiftCorr(torch.autograd.Function):
    """x (B,C,H,W) -> R (13,C,C) fp64 [HALF order], total (C,) fp64 = sum of x per channel, the zero-padded 3x3 patches of the
    ring pixels (B, 9C, 2 (W + 2) + 2 H) [row c*9 + tap] and, with `positions` (B, n), the patches at those cells (B, 9C, n).
    Device tensors: two launches of the shifted-view GEMM (forward: A = padded x, B = its 13 shifts, split-K partials summed in
    fp64; backward: dX = K Xshift over all 25 shifts with the gradient of `total` as the row bias); the patch gradients are
    scatter-added into dX by this node's backward (no zero-filled map, no extra full-size addition).  Host tensors (the fp64
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
        taps = _const(("ptaps", Wp, str(x.device)), lambda: torch.tensor(
            [(dy + 1) * Wp + dx + 1 for dy in range(3) for dx in range(3)], device=x.device))
        flat = xp.reshape(B, C, Lp)

        def gather(base):                                   # base (Bx, n): window corner - (1, 1) -> (B, 9C, n), row c*9 + tap
            n = base.shape[1]
            idx = (base.unsqueeze(1) + taps.view(1, 9, 1)).reshape(base.shape[0], 1, 9 * n)
            return flat.gather(2, idx.expand(B, C, 9 * n)).reshape(B, 9 * C, n)

        rbase = _ring_bases(H, W, Wp, x.device).view(1, -1)
        ctx.rbase = (rbase + Wp + 1).expand(B, -1).contiguous()                              # top-left element of each 3x3 window
        outs = (R, total, gather(rbase))
        ctx.pbase = None
        if positions is not None:
            pos = positions.long()
            base = (pos // W) * Wp + pos % W
            ctx.pbase = (base + Wp + 1).contiguous()
            outs = outs + (gather(base),)
        return outs

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dR, dtotal, dRing, dP=None):
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
        # patch gradients (ring, listed positions): scatter-added onto the padded plane; what lands on the padding is dropped below
        for grad, pbase in ((dRing, ctx.rbase), (dP, ctx.pbase)):
            if grad is None or pbase is None:
                continue
            n = pbase.shape[1]
            if dxp.is_cuda:
                from dcd_amd import _lib
                gp = grad.to(dxp.dtype).contiguous()
                st = _lib.lib().dcd_patch_scatter_add(_lib.stream_of(dxp), gp.data_ptr(), pbase.data_ptr(), B, C, Lp, Wp, n, dxp.data_ptr())
                _lib.check(st, "dcd_patch_scatter_add")
            else:
                taps = torch.tensor([dy * Wp + dx for dy in range(3) for dx in range(3)], device=dxp.device)
                idx = (pbase.unsqueeze(1) + taps.view(1, 9, 1)).reshape(B, 1, 9 * n)
                dxp.view(B, C, Lp).scatter_add_(2, idx.expand(B, C, 9 * n), grad.reshape(B, C, 9 * n).to(dxp.dtype))
        return dxp[:, :, 2:2 + H, 2:2 + W].contiguous(), None


def _assembly_tables(device):
    """E[t,t'] = index of d = t' - t in ALL25; half_of / flip: R_d from the 13 stored matrices (R_-d = R_d^T)."""
    def build():
        E = torch.tensor([[ALL25.index((t2 // 3 - t1 // 3, t2 % 3 - t1 % 3)) for t2 in range(9)] for t1 in range(9)], device=device)
        half_of = torch.tensor([HALF.index(d) if d in HALF else HALF.index((-d[0], -d[1])) for d in ALL25], device=device)
        flip = torch.tensor([d not in HALF for d in ALL25], device=device)
        return E, half_of, flip
    return _const(("tables", str(device)), build)


def patch_moments(x, positions=None):
    """S1 (9C,) and G (9C, 9C) in fp64 [index c*9 + tap] from the autocorrelation matrices of x (see above); with `positions`
    (B, n) also the 3x3 patches at those cells (B, 9C, n) from the same node, else None as third value."""
    mode = os.environ.get("DCD_TRUNK_GRAM", "auto")
    # auto: the autocorrelation form from one 96x320 image per rank on (round 5, one image: 12.50 vs 12.70 ms per step; with the
    # border terms as ~60 small launches, rounds 2-3, the threshold was 100k pixels, in round 4 two images)
    small = x.is_cuda and x.shape[0] * x.shape[2] * x.shape[3] < 20000
    if mode == "bmm" or (mode == "auto" and small) or x.shape[2] < 5 or x.shape[3] < 5:
        return patch_moments_gram(x) + (None,)
    B, C, H, W = x.shape
    res = _ShiftCorr.apply(x, positions)
    R13, total, ring = res[:3]
    P = res[3] if positions is not None else None
    E, half_of, flip = _assembly_tables(x.device)
    R25 = R13[half_of]
    R25 = torch.where(flip.view(25, 1, 1), R25.transpose(1, 2), R25)                           # R_-d = R_d^T
    G_ext = R25[E].permute(2, 0, 3, 1).reshape(9 * C, 9 * C)                                   # [t, t', c, c'] -> [(c,t), (c',t')]
    G = G_ext - _Gram.apply(ring)
    S1 = total.view(C, 1).expand(C, 9).reshape(9 * C) - ring.sum(2).double().sum(0)
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


_FUSED_STATS = os.environ.get("DCD_TRUNK_STATS_KERNELS", "1") != "0"       # 0: the finalisation as tensor operations (A/B timing)


class _TrunkScaleShift(torch.autograd.Function):
    """(Wall (T,O,K) fp32, S1 (K) fp64, G (K,K) fp64, gamma (T,O), beta (T,O)) -> scale, shift (T,O) fp32 of the trunks' training
    BatchNorm, and -- not differentiable -- mean, biased variance (T,O) fp64 for the running estimates.  Device tensors only:
    one fp64 library product W [G | S1] and two small kernels forward (row sums, finalisation: csrc/norm.hip), two kernels and two
    products backward, with the all-reduce of the per-channel sums (and of their gradients) in between under SyncBN.  The
    same mathematics as the tensor-operation form in `trunks_at` (which the host-logic tests run in fp64): ~30 small launches
    forward and ~50 in autograd's backward became 6 + 7."""

    @staticmethod
    def forward(ctx, Wall, S1, G, gamma, beta, n_local, eps, group):
        from dcd_amd import _lib
        L = _lib.lib()
        T, O, K = Wall.shape
        R = T * O
        dev = Wall.device
        st = _lib.stream_of(Wall)
        Wd = Wall.reshape(R, K).double()
        GS = torch.cat((G, S1.unsqueeze(1)), dim=1)                                             # (K, K + 1)
        WG = Wd @ GS                                                                            # (R, K + 1)
        sums = torch.empty((R, 2), dtype=torch.float64, device=dev)
        _lib.check(L.dcd_trunk_row_sums(st, WG.data_ptr(), Wd.data_ptr(), R, K, sums.data_ptr()), "dcd_trunk_row_sums")
        n = float(n_local)
        if group is not None:
            import torch.distributed as dist
            dist.all_reduce(sums, group=group)
            n *= dist.get_world_size(group)
        gamma, beta = gamma.reshape(R).float().contiguous(), beta.reshape(R).float().contiguous()
        scale = torch.empty(R, dtype=torch.float32, device=dev)
        shift = torch.empty(R, dtype=torch.float32, device=dev)
        stats = torch.empty((R, 3), dtype=torch.float64, device=dev)
        _lib.check(L.dcd_trunk_finalize_forward(st, sums.data_ptr(), gamma.data_ptr(), beta.data_ptr(), n, float(eps), R, scale.data_ptr(),
                                                shift.data_ptr(), stats.data_ptr()), "dcd_trunk_finalize_forward")
        ctx.save_for_backward(Wd, GS, WG, gamma, stats)
        ctx.geom, ctx.n, ctx.group = (T, O, K), n, group
        mean, var = stats[:, 0].reshape(T, O), stats[:, 1].reshape(T, O)
        ctx.mark_non_differentiable(mean, var)
        return scale.view(T, O), shift.view(T, O), mean, var

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dscale, dshift, _dmean, _dvar):
        from dcd_amd import _lib
        L = _lib.lib()
        Wd, GS, WG, gamma, stats = ctx.saved_tensors
        T, O, K = ctx.geom
        R = T * O
        dev = Wd.device
        st = _lib.stream_of(Wd)
        dscale, dshift = dscale.reshape(R).float().contiguous(), dshift.reshape(R).float().contiguous()
        dsums = torch.empty((R, 2), dtype=torch.float64, device=dev)
        dgamma = torch.empty(R, dtype=torch.float32, device=dev)
        dbeta = torch.empty(R, dtype=torch.float32, device=dev)
        _lib.check(L.dcd_trunk_finalize_backward(st, dscale.data_ptr(), dshift.data_ptr(), gamma.data_ptr(), stats.data_ptr(), ctx.n, R,
                                                 dsums.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr()), "dcd_trunk_finalize_backward")
        if ctx.group is not None:                     # the forward summed the ranks' sums: every rank's sums feel every rank's gradient
            import torch.distributed as dist
            dist.all_reduce(dsums, group=ctx.group)
        dWG = torch.empty((R, K + 1), dtype=torch.float64, device=dev)
        _lib.check(L.dcd_trunk_grad_wg(st, dsums.data_ptr(), Wd.data_ptr(), R, K, dWG.data_ptr()), "dcd_trunk_grad_wg")
        # sum y^2 = sum_k WG[r][k] Wd[r][k] depends on Wd twice: through the product and directly
        dWd = torch.addcmul(dWG @ GS.t(), dsums[:, 1:2], WG[:, :K])
        # (K, K + 1); the transposed view as the first operand makes hipBLASLt pick a 159 us fp64 kernel at DGDE's 2816 x 576 where
        # the same product from a contiguous copy takes 59 (+ 8 for the copy): tools/probes/f64_gemm_variants.py
        dGS = (Wd.t().contiguous() @ dWG) if Wd.is_cuda else Wd.t() @ dWG
        return dWd.float().view(T, O, K), dGS[:, K], dGS[:, :K], dgamma.view(T, O), dbeta.view(T, O), None, None, None


def _sum_rows(t):
    """(T, B, M, O) -> (T, O), the sum over B and M in steps of at most 64 elements per output: a reduction that short runs in
    one workgroup per output tile.  ATen's ONE-step form of the same sum (what autograd's broadcast backward issues) is a
    multi-block kernel whose scratch semaphore is cleared by hipMemsetAsync -- a memset node inside a captured graph, which on
    this stack is not ordered reliably against its kernel (profiles/r02_graph_memset_hazard.txt)."""
    T, B, M, O = t.shape
    while M > 1:
        # the largest divisor of M up to 64; an M whose smallest prime factor exceeds 64 (MAX_OBJECTS = 67, a border walk of
        # 16 * 67 cells) has none but 1: zero-pad to a multiple of 64 instead, so every pass shrinks M
        c = next(d for d in range(min(M, 64), 0, -1) if M % d == 0)
        if c == 1:
            c = 64
            t = torch.nn.functional.pad(t, (0, 0, 0, (-M) % c))
            M = t.shape[2]
        t = t.reshape(T, B, M // c, c, O).sum(3)
        M = M // c
    return t.reshape(T, B, O).sum(1)


class _AffineRelu(torch.autograd.Function):
    """relu(y * scale + shift) for y (T, B, M, O) and per-(trunk, channel) scale / shift (T, O), with the gradients of scale and
    shift summed by `_sum_rows`.  As plain tensor operations the broadcast's backward is a one-step reduction over B x M rows with
    a memset-cleared semaphore: in the whole-step HIP graph it returned wrong sums in some replays (the trunk's BatchNorm gradients
    up to 85x too large; a graphed training run left the eager trajectory after ~40 steps -- round 5, tools/probes/graph_twin2.py)."""

    @staticmethod
    def forward(ctx, y, scale, shift):
        T, O = scale.shape
        out = torch.relu(torch.addcmul(shift.view(T, 1, 1, O), y, scale.view(T, 1, 1, O)))
        ctx.save_for_backward(y, scale, out)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        y, scale, out = ctx.saved_tensors
        T, O = scale.shape
        gz = g * (out > 0).to(g.dtype)
        gy = gz * scale.view(T, 1, 1, O) if ctx.needs_input_grad[0] else None
        gscale = _sum_rows(gz * y) if ctx.needs_input_grad[1] else None
        gshift = _sum_rows(gz) if ctx.needs_input_grad[2] else None
        return gy, gscale, gshift


class _PatchLinear(torch.autograd.Function):
    """y[b, m, o] = sum_k X[b, k, m] W[o, k] (`einsum('bkm,ok->bmo')`) for the one trunk that is also evaluated at the ~832 border
    cells, as BATCHED products both ways.  The einsum's own weight gradient is ONE skinny product with a 6 656-long contraction
    (B x 832 positions), the shape for which BLAS libraries pick split-K kernels that accumulate into a cleared buffer; per-image
    products summed over the batch by an element-wise reduction have no such accumulator.  A precaution taken while the captured
    step's wrong gradients were hunted (round 5: they turned out to come from `_AffineRelu`'s predecessor, see there), kept
    because it also drops a transposed copy of X."""

    @staticmethod
    def forward(ctx, X, W):
        B, K, M = X.shape
        ctx.save_for_backward(X, W)
        return torch.bmm(X.transpose(1, 2), W.t().unsqueeze(0).expand(B, K, W.shape[0]))          # (B, M, O)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        X, W = ctx.saved_tensors
        B, K, M = X.shape
        gy = gy.contiguous()
        gX = torch.bmm(W.t().unsqueeze(0).expand(B, K, W.shape[0]), gy.transpose(1, 2)) if ctx.needs_input_grad[0] else None   # (B, K, M)
        gW = torch.bmm(gy.transpose(1, 2), X.transpose(1, 2)).sum(0) if ctx.needs_input_grad[1] else None                       # (O, K)
        return gX, gW


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
    n = B * H * W
    group = trunks[0][1].sync_group
    eps = trunks[0][1].eps
    if _FUSED_STATS and x.is_cuda and Wall.dtype == torch.float32:
        scale, shift, mean, var = _TrunkScaleShift.apply(Wall, S1, G, torch.stack([t[1].weight for t in trunks]),
                                                         torch.stack([t[1].bias for t in trunks]), n, eps, group)
        if group is not None:
            import torch.distributed as dist
            n = n * dist.get_world_size(group)
    else:
        Wd = Wall.double()
        # sum y = W S1 and sum y^2 = diag(W G W^T) from ONE product W [G | S1] (S1 as a 577th column: the separate fp64
        # matrix-vector product was a 100 us rocBLAS gemv, and another in the backward)
        WG = Wd.reshape(-1, K) @ torch.cat((G, S1.unsqueeze(1)), dim=1)                        # (T O, K + 1)
        sums = torch.stack((WG[:, K], (WG[:, :K] * Wd.reshape(-1, K)).sum(-1)), dim=-1).view(len(trunks), -1, 2)   # (T, O, 2): sum y, sum y^2
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
        scale = gamma * torch.rsqrt(var + eps)                                                 # (T, O) fp64
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
    at_centres = _AffineRelu.apply(y, scale, shift)
    at_extra = None
    if extra is not None:
        i, pos = extra
        Xe = patches(pos, False)
        ye = _PatchLinear.apply(Xe, Wall[i]) if x.is_cuda else torch.einsum('bkm,ok->bmo', Xe, Wall[i])
        at_extra = _AffineRelu.apply(ye.unsqueeze(0), scale[i:i + 1], shift[i:i + 1])[0]
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
