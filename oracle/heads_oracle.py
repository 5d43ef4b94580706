"""numpy restatements of the non-DCN hot-path routines.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Each function follows the reference line by line (paths relative to /root/reference) and is pinned
against outputs of the imported Python reference by tests/test_oracle_golden.py (fixtures in
tests/golden/, produced by tests/golden/make_golden.py).
"""
import numpy as np


def triu_pairs(K):
    """Pair order of Anno_Encoder.get_up (DGDE/model/anno_encoder.py:313-324): i<j, row major."""
    return np.triu_indices(K, k=1)


def pairs_kpts_depth(kps, kps3d, rot_y, P, kmask=None, training=False, num_k=1500, zmin=2.0, zmax=80.0,
                     normalized=False, sub_b3=True):
    """decode_pairs_kpts_depth (DGDE/model/anno_encoder.py:326-390); GMW compute_z (GMW/main.py:373-416)
    with zmin=0.1, normalized=True, sub_b3=False.  float32 arithmetic in the reference's operation order.
    Returns (depth, mask or None, idx or None); train-mode order = descending |dv|, ties by lower pair index."""
    f = np.float32
    kps, kps3d, P = kps.astype(f), kps3d.astype(f), P.astype(f)
    rot = np.asarray(rot_y, dtype=f).reshape(-1)
    N, K = kps.shape[:2]
    if normalized:
        v = kps[:, :, 1]
    else:
        v = (kps[:, :, 1] - P[:, None, 1, 2]) / P[:, None, 1, 1]           # :333-334
    X, Y, Z = kps3d[:, :, 0], kps3d[:, :, 1], kps3d[:, :, 2]
    cs, sn = np.cos(rot)[:, None].astype(f), np.sin(rot)[:, None].astype(f)  # :343-344
    C = X * sn - Z * cs                                                      # :349
    H1, H2 = Y, v * C                                                        # :347,351-353
    i, j = triu_pairs(K)
    hmat = (H1[:, i] - H1[:, j]) + (H2[:, i] - H2[:, j])                     # :367
    dv = np.abs(v[:, i] - v[:, j])                                           # :369
    z = np.abs(hmat) / np.maximum(dv, f(1e-10))                              # :371
    z = np.minimum(np.maximum(z, f(zmin)), f(zmax))                          # :375
    mask = None
    if kmask is not None:
        km = kmask.astype(f)
        mask = km[:, i] * km[:, j]                                           # :362-365
    idx = None
    if training:
        # torch.topk(|dv|, 1500) (:379): descending; tie order is unspecified in torch, fixed here
        idx = np.stack([np.lexsort((np.arange(dv.shape[1]), -dv[n].astype(np.float64)))[:num_k] for n in range(N)])
        z = np.take_along_axis(z, idx, axis=1)                               # :380
        if mask is not None:
            mask = np.take_along_axis(mask, idx, axis=1)                     # :382
    if sub_b3:
        z = z - P[:, 2, 3][:, None]                                          # :385
    return z.astype(f), mask, idx


def focal_loss(pred, target, alpha=2.0, beta=4.0):
    """FocalLoss.forward (DGDE/model/layers/focal_loss.py:57-86) -> (loss_sum, num_positive)."""
    p = np.clip(pred.astype(np.float32), np.float32(1e-10), np.float32(1 - 1e-10)).astype(np.float64)
    t = target.astype(np.float64)
    pos = (t == 1)
    neg = (t < 1) & (t >= 0)
    negw = np.power(1 - t, beta)
    pl = np.log(p) * np.power(1 - p, alpha) * pos
    nl = np.log(1 - p) * np.power(p, alpha) * negw * neg
    return float((-nl - pl).sum()), float(pos.sum())


def giou_loss(pred, target):
    """IOULoss('giou').forward (DGDE/model/layers/iou_loss.py:12-49) -> (losses, ious)."""
    f = np.float32
    p, t = pred.astype(f), target.astype(f)
    pl, pt, pr, pb = p[:, 0], p[:, 1], p[:, 2], p[:, 3]
    tl, tt, tr, tb = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
    ta = (tl + tr) * (tt + tb)
    pa = (pl + pr) * (pt + pb)
    wi = np.minimum(pl, tl) + np.minimum(pr, tr)
    gwi = np.maximum(pl, tl) + np.maximum(pr, tr)
    hi = np.minimum(pb, tb) + np.minimum(pt, tt)
    ghi = np.maximum(pb, tb) + np.maximum(pt, tt)
    ac = gwi * ghi + f(1e-7)
    ai = wi * hi
    au = ta + pa - ai
    ious = (ai + f(1.0)) / (au + f(1.0))
    gious = ious - (ac - au) / ac
    return (f(1) - gious).astype(f), ious.astype(f)


def nms_hm(heat):
    """nms_hm (DGDE/model/layers/utils.py:45-58): heat * (maxpool3x3(heat) == heat)."""
    B, C, H, W = heat.shape
    pad = np.full((B, C, H + 2, W + 2), -np.inf, dtype=heat.dtype)
    pad[:, :, 1:-1, 1:-1] = heat
    mx = heat.copy()
    for dy in range(3):
        for dx in range(3):
            mx = np.maximum(mx, pad[:, :, dy:dy + H, dx:dx + W])
    return heat * (mx == heat).astype(heat.dtype)


def select_topk(heat, K):
    """select_topk (DGDE/model/layers/utils.py:61-100).  The reference asserts CUDA tensors (:83-84,93), so it
    cannot be run here; this restates its arithmetic with a fixed tie rule (lower index first).
    Returns scores (B,K) f32, inds (B,K) i64, clses (B,K) f32, ys, xs (B,K) f32."""
    B, C, H, W = heat.shape
    flat = heat.reshape(B, C, H * W)
    scores_all = np.empty((B, C, K), np.float32)
    inds_all = np.empty((B, C, K), np.int64)
    for b in range(B):
        for c in range(C):
            order = np.lexsort((np.arange(H * W), -flat[b, c].astype(np.float64)))[:K]   # topk over H*W (:75)
            inds_all[b, c] = order
            scores_all[b, c] = flat[b, c, order]
    ys_all = (inds_all // W).astype(np.float32)                                           # :80 (int division then float)
    xs_all = (inds_all % W).astype(np.float32)                                            # :81
    sa = scores_all.reshape(B, C * K)
    scores = np.empty((B, K), np.float32)
    sel = np.empty((B, K), np.int64)
    for b in range(B):
        order = np.lexsort((np.arange(C * K), -sa[b].astype(np.float64)))[:K]            # :89
        sel[b] = order
        scores[b] = sa[b, order]
    clses = (sel.astype(np.float32) / np.float32(K)).astype(np.float32)                   # :91 true division
    inds = np.take_along_axis(inds_all.reshape(B, C * K), sel, axis=1)                    # :96
    ys = np.take_along_axis(ys_all.reshape(B, C * K), sel, axis=1)
    xs = np.take_along_axis(xs_all.reshape(B, C * K), sel, axis=1)
    return scores, inds, clses, ys, xs


def select_point_of_interest(index, feat):
    """select_point_of_interest (DGDE/model/layers/utils.py:120-145): index (B,M) linear or (B,M,2) (x,y)."""
    B, C, H, W = feat.shape
    if index.ndim == 3:
        index = index[:, :, 1] * W + index[:, :, 0]
    nhwc = feat.transpose(0, 2, 3, 1).reshape(B, H * W, C)
    return np.take_along_axis(nhwc, index.astype(np.int64)[:, :, None].repeat(C, axis=2), axis=1)
