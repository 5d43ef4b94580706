"""Autograd wrappers over the C-ABI kernels that are not DCNv2 (csrc/heads.hip).

Each function keeps the calling convention of the reference routine it replaces so the model code
reads like the reference's:
  pairs_kpts_depth        <- Anno_Encoder.decode_pairs_kpts_depth   (DGDE/model/anno_encoder.py:326-390)
  compute_z               <- GMW compute_z                          (GMW/main.py:373-416)
  focal_loss              <- FocalLoss.forward                      (DGDE/model/layers/focal_loss.py:57-86)
  giou_loss               <- IOULoss.forward (loss_type 'giou')     (DGDE/model/layers/iou_loss.py:12-49)
  nms_hm / select_topk / select_point_of_interest                   (DGDE/model/layers/utils.py:45-145)
All of them run on the GPU only and raise if the HIP library is missing.
"""
import os

import torch

from . import _lib


def _f32c(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


# ----------------------------------------------------------------------------------------------
# Edge-constraint depth solver
# ----------------------------------------------------------------------------------------------
class _PairsDepth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, kps, kps3d, rot_y, P, kmask, topk, zmin, zmax, normalized, sub_b3):
        _lib.require_cuda(kps, kps3d, rot_y, P)
        L = _lib.lib()
        kps, kps3d, P = _f32c(kps), _f32c(kps3d), _f32c(P)
        rot = _f32c(rot_y).reshape(-1)
        N, K = kps.shape[0], kps.shape[1]
        npairs = K * (K - 1) // 2
        M = topk if topk else npairs
        dev = kps.device
        depth = torch.empty((N, M), dtype=torch.float32, device=dev)
        pair_idx = torch.empty((N, M), dtype=torch.int32, device=dev) if topk else None
        km = None
        pmask = None
        if kmask is not None:
            km = kmask.to(torch.uint8).contiguous()
            pmask = torch.empty((N, M), dtype=torch.float32, device=dev) if topk else None
        st = L.dcd_edge_depth_forward(_lib.stream_of(kps), kps.data_ptr(), kps3d.data_ptr(), rot.data_ptr(),
                                      P.data_ptr(), _lib.ptr(km), N, K, int(topk), float(zmin), float(zmax),
                                      int(normalized), int(sub_b3), depth.data_ptr(), _lib.ptr(pair_idx),
                                      _lib.ptr(pmask))
        _lib.check(st, "dcd_edge_depth_forward")
        ctx.save_for_backward(kps, kps3d, rot, P, pair_idx if pair_idx is not None else torch.empty(0))
        ctx.cfg = (int(topk), float(zmin), float(zmax), int(normalized))
        ctx.mark_non_differentiable(*[t for t in (pair_idx, pmask) if t is not None])
        return depth, pair_idx, pmask

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gdepth, _gidx, _gmask):
        kps, kps3d, rot, P, pair_idx = ctx.saved_tensors
        topk, zmin, zmax, normalized = ctx.cfg
        L = _lib.lib()
        N, K = kps.shape[0], kps.shape[1]
        gk = torch.empty_like(kps)
        gk3 = torch.empty_like(kps3d)
        gdepth = _f32c(gdepth)
        st = L.dcd_edge_depth_backward(_lib.stream_of(kps), kps.data_ptr(), kps3d.data_ptr(), rot.data_ptr(),
                                       P.data_ptr(), gdepth.data_ptr(), pair_idx.data_ptr() if topk else None,
                                       N, K, topk, zmin, zmax, normalized, gk.data_ptr(), gk3.data_ptr())
        _lib.check(st, "dcd_edge_depth_backward")
        return gk, gk3, None, None, None, None, None, None, None, None


def pairs_kpts_depth(kps, kps_3d, rot_y, K, training=False, kpts_2d_mask=None, num_k=1500):
    """decode_pairs_kpts_depth: returns (depth_all, depth_mask).  Train: (N,1500) + float mask; eval: (N,2628), None."""
    topk = num_k if training else 0
    depth, _idx, pmask = _PairsDepth.apply(kps, kps_3d, rot_y, K, kpts_2d_mask, topk, 2.0, 80.0, 0, 1)
    if kpts_2d_mask is not None:
        if pmask is None:  # eval with a mask: the reference returns get_up(mask_i*mask_j) over all pairs
            m = kpts_2d_mask.float()
            iu = torch.triu_indices(m.shape[1], m.shape[1], offset=1, device=m.device)
            pmask = m[:, iu[0]] * m[:, iu[1]]
        return depth, pmask
    return depth, None


def compute_z(kpts_2d, kpts_3d, pred_rot, num_k=1500):
    """GMW compute_z: K-normalised keypoints, clamp [0.1, 80], no b3; returns (Z_v_raw (N,2628), good_idx (N,1500))."""
    N, K = kpts_2d.shape[0], kpts_2d.shape[1]
    P = torch.zeros((N, 3, 4), dtype=torch.float32, device=kpts_2d.device)
    P[:, 1, 1] = 1.0
    z_all, _, _ = _PairsDepth.apply(kpts_2d, kpts_3d, pred_rot, P, None, 0, 0.1, 80.0, 1, 0)
    with torch.no_grad():
        _, idx, _ = _PairsDepth.apply(kpts_2d.detach(), kpts_3d.detach(), pred_rot, P, None, num_k, 0.1, 80.0, 1, 0)
    return z_all, idx.long()


# ----------------------------------------------------------------------------------------------
# Losses
# ----------------------------------------------------------------------------------------------
class _Focal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, prediction, target, alpha, beta):
        _lib.require_cuda(prediction, target)
        L = _lib.lib()
        p, t = _f32c(prediction), _f32c(target)
        out = torch.empty(2, dtype=torch.float32, device=p.device)
        need_grad = prediction.requires_grad
        g = torch.empty_like(p) if need_grad else None
        st = L.dcd_focal_loss(_lib.stream_of(p), p.data_ptr(), t.data_ptr(), p.numel(), float(alpha), float(beta),
                              out.data_ptr(), _lib.ptr(g))
        _lib.check(st, "dcd_focal_loss")
        ctx.save_for_backward(g if g is not None else torch.empty(0))
        ctx.shape = prediction.shape
        npos = out[1]
        ctx.mark_non_differentiable(npos)
        return out[0], npos

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gloss, _gnpos):
        (g,) = ctx.saved_tensors
        return (g * gloss).view(ctx.shape), None, None, None


def focal_loss(prediction, target, alpha=2, beta=4):
    """Returns (loss_sum, num_positive) like FocalLoss.forward."""
    return _Focal.apply(prediction, target, alpha, beta)


class _GIoU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        _lib.require_cuda(pred, target)
        L = _lib.lib()
        p, t = _f32c(pred), _f32c(target)
        N = p.shape[0]
        losses = torch.empty(N, dtype=torch.float32, device=p.device)
        ious = torch.empty(N, dtype=torch.float32, device=p.device)
        g = torch.empty_like(p) if pred.requires_grad else None
        st = L.dcd_giou_loss(_lib.stream_of(p), p.data_ptr(), t.data_ptr(), N, losses.data_ptr(), ious.data_ptr(),
                             _lib.ptr(g))
        _lib.check(st, "dcd_giou_loss")
        ctx.save_for_backward(g if g is not None else torch.empty(0))
        ctx.mark_non_differentiable(ious)
        return losses, ious

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, glosses, _gious):
        (g,) = ctx.saved_tensors
        return g * glosses.unsqueeze(1), None


def giou_loss(pred, target):
    """Returns (losses, ious) like IOULoss('giou').forward."""
    return _GIoU.apply(pred, target)


# ----------------------------------------------------------------------------------------------
# Heat-map decode
# ----------------------------------------------------------------------------------------------
# ----------------------------------------------------------------------------------------------
# Output layers of the regression heads at listed rows (csrc/heads.hip, dcd_head_rows_*)
# ----------------------------------------------------------------------------------------------
class _HeadRows(torch.autograd.Function):
    @staticmethod
    def _args(feat, trunk_of, weights, biases):
        a = _lib.HeadRowsArgs()
        T, R, K = feat.shape
        a.n_heads, a.T, a.R, a.K = len(weights), T, R, K
        c = 0
        for j, (t, w) in enumerate(zip(trunk_of, weights)):
            a.trunk[j], a.ch0[j], a.out[j] = t, c, w.shape[0]
            a.weight[j] = w.data_ptr()
            a.bias[j] = None if biases[j] is None else biases[j].data_ptr()
            c += w.shape[0]
        a.C = c
        a.feat = feat.data_ptr()
        return a

    @staticmethod
    def forward(ctx, feat, trunk_of, n, *params):
        _lib.require_cuda(feat, *[p for p in params if p is not None])
        feat = _f32c(feat)
        weights = [_f32c(w).reshape(w.shape[0], -1) for w in params[:n]]
        biases = [None if b is None else _f32c(b) for b in params[n:]]
        a = _HeadRows._args(feat, trunk_of, weights, biases)
        y = torch.empty((feat.shape[1], a.C), dtype=torch.float32, device=feat.device)
        a.y = y.data_ptr()
        _lib.check(_lib.lib().dcd_head_rows_forward(_lib.stream_of(feat), a), "dcd_head_rows_forward")
        ctx.trunk_of, ctx.n = trunk_of, n
        ctx.shapes = [tuple(p.shape) for p in params[:n]]
        ctx.has_bias = [b is not None for b in biases]
        ctx.save_for_backward(feat, *weights)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        feat, weights = ctx.saved_tensors[0], list(ctx.saved_tensors[1:])
        n = ctx.n
        gy = _f32c(gy)
        a = _HeadRows._args(feat, ctx.trunk_of, weights, [None] * n)
        gfeat = torch.empty_like(feat)
        gws = [torch.empty_like(w) for w in weights]
        gbs = [torch.empty(w.shape[0], dtype=torch.float32, device=feat.device) if hb else None for w, hb in zip(weights, ctx.has_bias)]
        a.grad_y, a.grad_feat = gy.data_ptr(), gfeat.data_ptr()
        for j in range(n):
            a.grad_weight[j] = gws[j].data_ptr()
            a.grad_bias[j] = None if gbs[j] is None else gbs[j].data_ptr()
        _lib.check(_lib.lib().dcd_head_rows_backward(_lib.stream_of(feat), a), "dcd_head_rows_backward")
        return (gfeat, None, None) + tuple(g.view(s) for g, s in zip(gws, ctx.shapes)) + tuple(gbs)


def head_rows(feat, trunk_of, weights, biases):
    """y (R, sum out_j): the 1x1 output layers (weights[j] (out_j, K[, 1, 1]), biases[j]) of the regression heads applied to their
    trunks' outputs feat (T, R, K) at R listed rows; trunk_of[j] = trunk of head j, heads ordered by trunk
    (detector_predictor.py:84-101, :198-203).  One launch forward, two backward."""
    return _HeadRows.apply(feat, tuple(int(t) for t in trunk_of), len(weights), *weights, *biases)


# ----------------------------------------------------------------------------------------------
# Per-object rows of the training loss (csrc/loss_rows.hip)
# ----------------------------------------------------------------------------------------------
LOSS_ROWS_NCOL = 25
# (name in the targets dict, dtype the kernel reads)
_ROWS_TARGETS = (("reg_mask", torch.uint8), ("trunc_mask", torch.uint8), ("find_pcl", torch.uint8), ("ori_mask", torch.uint8),
                 ("cls_ids", torch.int32), ("target_centers", torch.int32), ("pad_size", torch.int64), ("bboxes", torch.float32),
                 ("locations", torch.float32), ("rotys", torch.float32), ("offset_3D", torch.float32),
                 ("dimensions", torch.float32), ("orientations", torch.float32), ("keypoints", torch.float32),
                 ("keypoints_depth_mask", torch.float32), ("extra_kpts_2d", torch.float32), ("extra_kpts_3d", torch.float32),
                 ("Calib_P", torch.float32), ("calib", torch.float32))
_ROWS_FIELD = {"target_centers": "centers", "keypoints_depth_mask": "kp_depth_mask", "extra_kpts_2d": "kpts2d",
               "extra_kpts_3d": "kpts3d", "Calib_P": "calib_P"}


def _typed(t, dtype):
    if t.dtype == torch.bool and dtype == torch.uint8:
        t = t.view(torch.uint8)
    elif t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


class _LossRows(torch.autograd.Function):
    """sums (25,) = column sums of the per-object loss rows; differentiable w.r.t. `pois` (B, M, C)."""

    @staticmethod
    def _args(spec, pois, targets, dim_mean):
        a = _lib.LossRowsArgs()
        B, M, C = pois.shape
        a.B, a.M, a.C, a.K, a.NP = B, M, C, spec["K"], spec["NP"]
        a.num_classes = dim_mean.shape[0]
        for k in ("ch_box2d", "ch_offset", "ch_corner", "ch_corner_unc", "ch_dims", "ch_ori_cls", "ch_ori_off", "ch_depth",
                  "ch_depth_unc", "ch_kpts2d", "ch_kpts3d", "trunc_log", "depth_lo", "depth_hi", "unc_lo", "unc_hi",
                  "depth_weight", "down_ratio", "kd_eps"):
            setattr(a, k, spec[k])
        a.dim_weight[0], a.dim_weight[1], a.dim_weight[2] = spec["dim_weight"]
        a.pois = pois.data_ptr()
        for (name, _), t in zip(_ROWS_TARGETS, targets):
            setattr(a, _ROWS_FIELD.get(name, name), t.data_ptr())
        a.dim_mean = dim_mean.data_ptr()
        return a

    @staticmethod
    def forward(ctx, pois, spec, dim_mean, *targets):
        _lib.require_cuda(pois, dim_mean, *targets)
        L = _lib.lib()
        pois = _f32c(pois)
        targets = tuple(_typed(t, dt) for t, (_, dt) in zip(targets, _ROWS_TARGETS))
        dim_mean = _f32c(dim_mean)
        B, M, C = pois.shape
        BM, K, NP = B * M, spec["K"], spec["NP"]
        dev, f32 = pois.device, torch.float32
        stream = _lib.stream_of(pois)
        a = _LossRows._args(spec, pois, targets, dim_mean)
        kps_pred, kps_tgt = torch.empty((2, BM, K, 2), dtype=f32, device=dev).unbind(0)
        kps3d_pred, kps3d_tgt = torch.empty((2, BM, K, 3), dtype=f32, device=dev).unbind(0)
        rot = torch.empty((2, BM), dtype=f32, device=dev)
        P_rows = torch.empty((2, BM, 3, 4), dtype=f32, device=dev)
        kmask = torch.empty((2, BM, K), dtype=torch.uint8, device=dev)
        a.kps_pred, a.kps_tgt, a.kps3d_pred, a.kps3d_tgt = (t.data_ptr() for t in (kps_pred, kps_tgt, kps3d_pred, kps3d_tgt))
        a.rot, a.P_rows, a.kmask = rot.data_ptr(), P_rows.data_ptr(), kmask.data_ptr()
        _lib.check(L.dcd_loss_rows_prepare(stream, a), "dcd_loss_rows_prepare")
        # one solver call over [predictions | targets]: the depths that are trained come from the predicted keypoints, the
        # pair mask from the TARGET keypoints' top-NP ordering (detector_loss.py:381 vs :378; kept as the reference has it)
        both_depth = torch.empty((2, BM, NP), dtype=f32, device=dev)
        both_mask = torch.empty((2, BM, NP), dtype=f32, device=dev)
        both_idx = torch.empty((2, BM, NP), dtype=torch.int32, device=dev)
        _lib.check(L.dcd_edge_depth_forward(stream, kps_pred.data_ptr(), kps3d_pred.data_ptr(), rot.data_ptr(), P_rows.data_ptr(),
                                            kmask.data_ptr(), 2 * BM, K, NP, 2.0, 80.0, 0, 1, both_depth.data_ptr(),
                                            both_idx.data_ptr(), both_mask.data_ptr()), "dcd_edge_depth_forward")
        depth, idx, pmask = both_depth[0], both_idx[0], both_mask[1]
        cols = torch.empty((LOSS_ROWS_NCOL, BM), dtype=f32, device=dev)
        corners = torch.empty((2, BM, 8, 3), dtype=f32, device=dev)
        iou3d = torch.empty(BM, dtype=f32, device=dev)
        sums = torch.empty(LOSS_ROWS_NCOL, dtype=f32, device=dev)
        a.pair_depth, a.pair_mask = depth.data_ptr(), pmask.data_ptr()
        a.cols, a.corners_pred, a.corners_tgt = cols.data_ptr(), corners[0].data_ptr(), corners[1].data_ptr()
        a.iou3d, a.sums = iou3d.data_ptr(), sums.data_ptr()
        _lib.check(L.dcd_loss_rows_forward(stream, a), "dcd_loss_rows_forward")
        ctx.spec = spec
        ctx.save_for_backward(pois, dim_mean, kps_pred, kps3d_pred, rot[0], P_rows[0], depth, pmask, idx, *targets)
        return sums

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gsums):
        pois, dim_mean, kps_pred, kps3d_pred, rot, P_rows, depth, pmask, idx = ctx.saved_tensors[:9]
        targets = ctx.saved_tensors[9:]
        spec = ctx.spec
        L = _lib.lib()
        B, M, C = pois.shape
        BM, K, NP = B * M, spec["K"], spec["NP"]
        stream = _lib.stream_of(pois)
        a = _LossRows._args(spec, pois, targets, dim_mean)
        gsums = _f32c(gsums)
        gpois = torch.empty_like(pois)
        gpair = torch.empty_like(depth)
        gk = torch.empty_like(kps_pred)
        gk3 = torch.empty_like(kps3d_pred)
        a.pair_depth, a.pair_mask, a.grad_sums = depth.data_ptr(), pmask.data_ptr(), gsums.data_ptr()
        a.grad_pois, a.grad_pair, a.grad_kps, a.grad_kps3d = gpois.data_ptr(), gpair.data_ptr(), gk.data_ptr(), gk3.data_ptr()
        _lib.check(L.dcd_loss_rows_backward(stream, a), "dcd_loss_rows_backward")
        _lib.check(L.dcd_edge_depth_backward(stream, kps_pred.data_ptr(), kps3d_pred.data_ptr(), rot.data_ptr(), P_rows.data_ptr(),
                                             gpair.data_ptr(), idx.data_ptr(), BM, K, NP, 2.0, 80.0, 0, gk.data_ptr(),
                                             gk3.data_ptr()), "dcd_edge_depth_backward")
        _lib.check(L.dcd_loss_rows_finish(stream, a), "dcd_loss_rows_finish")
        return (gpois, None, None) + (None,) * len(targets)


def loss_rows(pois, spec, dim_mean, targets):
    """Column sums (25,) of the per-object loss rows of Loss_Computation (detector_loss.py:405-583) for head outputs `pois`
    (B, M, C) at the object centres; `targets` = the stacked target fields (dict), `spec` = channel map and constants
    (keys of `dcd_loss_rows_args`).  One autograd node: 5 launches forward, 4 backward."""
    return _LossRows.apply(pois, spec, dim_mean, *[targets[name] for name, _ in _ROWS_TARGETS])


def nms_hm(heat_map, kernel=3, reso=1):
    kernel = int(kernel / reso)
    if kernel % 2 == 0:
        kernel += 1
    if kernel != 3:
        raise NotImplementedError("only the 3x3 NMS of the reference configuration is implemented in HIP")
    _lib.require_cuda(heat_map)
    hm = _f32c(heat_map)
    B, C, H, W = hm.shape
    out = torch.empty_like(hm)
    st = _lib.lib().dcd_nms_hm(_lib.stream_of(hm), hm.data_ptr(), B, C, H, W, out.data_ptr())
    _lib.check(st, "dcd_nms_hm")
    return out


def select_topk(heat_map, K=100, fuse_nms=False):
    """Returns (topk_scores, topk_inds, topk_clses, topk_ys, topk_xs), each (B, K), like the reference."""
    _lib.require_cuda(heat_map)
    L = _lib.lib()
    hm = _f32c(heat_map)
    B, C, H, W = hm.shape
    dev = hm.device
    scores = torch.empty((B, K), dtype=torch.float32, device=dev)
    inds = torch.empty((B, K), dtype=torch.int64, device=dev)
    clses = torch.empty((B, K), dtype=torch.float32, device=dev)
    ys = torch.empty((B, K), dtype=torch.float32, device=dev)
    xs = torch.empty((B, K), dtype=torch.float32, device=dev)
    nbytes = L.dcd_heatmap_topk_workspace_bytes(B, C, H, W, K)
    ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
    st = L.dcd_heatmap_topk(_lib.stream_of(hm), hm.data_ptr(), B, C, H, W, K, int(bool(fuse_nms)), scores.data_ptr(),
                            inds.data_ptr(), clses.data_ptr(), ys.data_ptr(), xs.data_ptr(), ws.data_ptr(), nbytes)
    _lib.check(st, "dcd_heatmap_topk")
    return scores, inds, clses, ys, xs


class _POI(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feature_maps, index):
        _lib.require_cuda(feature_maps, index)
        L = _lib.lib()
        f = _f32c(feature_maps)
        B, C, H, W = f.shape
        idx = index.to(torch.int64).contiguous()
        M = idx.shape[1]
        out = torch.empty((B, M, C), dtype=torch.float32, device=f.device)
        st = L.dcd_poi_gather(_lib.stream_of(f), f.data_ptr(), idx.data_ptr(), B, C, H, W, M, out.data_ptr())
        _lib.check(st, "dcd_poi_gather")
        ctx.save_for_backward(idx)
        ctx.shape = (B, C, H, W)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout):
        (idx,) = ctx.saved_tensors
        B, C, H, W = ctx.shape
        L = _lib.lib()
        gout = _f32c(gout)
        gfeat = torch.zeros((B, C, H, W), dtype=torch.float32, device=gout.device)
        st = L.dcd_poi_scatter_add(_lib.stream_of(gout), gout.data_ptr(), idx.data_ptr(), B, C, H, W, idx.shape[1],
                                   gfeat.data_ptr())
        _lib.check(st, "dcd_poi_scatter_add")
        return gfeat, None


def select_point_of_interest(batch, index, feature_maps):
    """(B, M, C) features at the given centres; `index` is (B,M,2) points (x,y) or (B,M) linear indices."""
    w = feature_maps.shape[3]
    if index.dim() == 3:
        index = index[:, :, 1] * w + index[:, :, 0]
    index = index.reshape(batch, -1)
    return _POI.apply(feature_maps, index)


class _ScatterAddAt(torch.autograd.Function):
    """fmap[b, :, index[b, m]] += vals[b, m, :] in place (one kernel, fp32 atomics); the gradient of `vals` is a gather of the
    map's gradient.  Replaces `index_put_(..., accumulate=True)` with four index tensors (eight bounds-check reductions, a
    sort and the scatter: ~40 launches) where the edge-fusion branch adds its outputs into the class map
    (DGDE/model/head/detector_predictor.py:186-196)."""

    @staticmethod
    def forward(ctx, fmap, vals, index):
        _lib.require_cuda(fmap, vals, index)
        if fmap.dtype != torch.float32 or not fmap.is_contiguous():
            raise RuntimeError("scatter_add_at: the map must be a contiguous float32 tensor (it is updated in place)")
        B, C, H, W = fmap.shape
        idx = index.to(torch.int64).contiguous()
        v = _f32c(vals)
        st = _lib.lib().dcd_poi_scatter_add(_lib.stream_of(fmap), v.data_ptr(), idx.data_ptr(), B, C, H, W, idx.shape[1],
                                            fmap.data_ptr())
        _lib.check(st, "dcd_poi_scatter_add")
        ctx.mark_dirty(fmap)
        ctx.save_for_backward(idx)
        return fmap

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        g = _f32c(g)
        B, C, H, W = g.shape
        gv = None
        if ctx.needs_input_grad[1]:
            gv = torch.empty((B, idx.shape[1], C), dtype=torch.float32, device=g.device)
            st = _lib.lib().dcd_poi_gather(_lib.stream_of(g), g.data_ptr(), idx.data_ptr(), B, C, H, W, idx.shape[1], gv.data_ptr())
            _lib.check(st, "dcd_poi_gather")
        return g, gv, None


class _HeadOutAndGather(torch.autograd.Function):
    """(W x + b, x at listed cells) for a 1x1 output convolution whose input is ALSO gathered at positions: the class head's last
    layer and the edge-fusion gather read the same 256-channel feature map (DGDE/model/head/detector_predictor.py:150-152,178-188).
    As one node the backward writes the feature gradient once -- W^T g as a batched GEMM, the gathered rows' gradient scatter-added
    into it in place -- instead of a zero-filled 251 MB map, a scatter and a full-size addition; the products are plain GEMMs on
    (B, C, HW) views, so the stock solver's NHWC transposes of the 251 MB operand (113 us per step) go too."""

    @staticmethod
    def forward(ctx, x, weight, bias, index):
        _lib.require_cuda(x, weight, index)
        x = _f32c(x)
        B, C, H, W = x.shape
        O = weight.shape[0]
        w2 = weight.reshape(O, C)
        out = torch.empty((B, O, H, W), dtype=torch.float32, device=x.device)      # returned as is (not a view: the edge-fusion
        torch.bmm(w2.unsqueeze(0).expand(B, O, C), x.view(B, C, H * W), out=out.view(B, O, H * W))   # term is added into it later)
        # (bmm with an expanded weight, not matmul(2-D, 3-D): that one folds the batch by transposing -- and copying -- the 251 MB operand)
        if bias is not None:
            out.add_(bias.view(1, O, 1, 1))
        idx = index.to(torch.int64).contiguous()
        M = idx.shape[1]
        g = torch.empty((B, M, C), dtype=torch.float32, device=x.device)
        st = _lib.lib().dcd_poi_gather(_lib.stream_of(x), x.data_ptr(), idx.data_ptr(), B, C, H, W, M, g.data_ptr())
        _lib.check(st, "dcd_poi_gather")
        ctx.save_for_backward(x, w2, idx)
        ctx.has_bias = bias is not None
        ctx.wshape = weight.shape
        return out, g

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gout, ggather):
        x, w2, idx = ctx.saved_tensors
        B, C, H, W = x.shape
        O = w2.shape[0]
        go = _f32c(gout).view(B, O, H * W)
        gx = torch.bmm(w2.t().unsqueeze(0).expand(B, C, O), go)                           # (B, C, HW), written once
        gg = _f32c(ggather)
        st = _lib.lib().dcd_poi_scatter_add(_lib.stream_of(x), gg.data_ptr(), idx.data_ptr(), B, C, H, W, idx.shape[1], gx.data_ptr())
        _lib.check(st, "dcd_poi_scatter_add")
        # the transposed operand is the small one (go^T, 0.7 MB): bmm may copy it, never the 251 MB feature map
        gw = torch.bmm(x.view(B, C, H * W), go.transpose(1, 2)).sum(0).t().reshape(ctx.wshape)
        # NOT go.sum((0, 2)): with one class that is a flat reduction of B*H*W values into one, which ATen runs as a multi-block
        # kernel with a memset-initialised accumulator -- inside a replayed HIP graph it returned a wrong sum from the second replay
        # on (the hazard of profiles/r02_graph_memset_hazard.txt; found in round 5 as a training run that left the eager
        # trajectory after ~40 graphed steps: tools/probes/graph_lr0.py).  The two-stage sums of csrc/norm.hip zero-fill with kernels.
        gb = channel_sums(_f32c(gout)) if ctx.has_bias else None
        return gx.view(B, C, H, W), gw, gb, None


def head_out_and_gather(x, weight, bias, index):
    """(conv1x1(x, weight, bias) (B,O,H,W), x gathered at the linear cell indices index (B,M) -> (B,M,C))."""
    return _HeadOutAndGather.apply(x, weight, bias, index)


class _Conv1x1OfCat(torch.autograd.Function):
    """conv1x1(cat(xs, dim=1), weight) without the concatenation: out = sum_i W[:, slice_i] x_i as batched GEMMs on (B, C_i, HW)
    views (the first writes, the others accumulate), input gradients W_i^T g written contiguously, weight gradient per slice.
    Replaces `torch.cat` + stock 1x1 convolution in DLA's Root (DGDE/model/backbone/dla_dcn.py:199-205): no concatenated copy in
    the forward, no strided gradient slices (each re-packed by its consumer) in the backward."""

    @staticmethod
    def forward(ctx, weight, *xs):
        _lib.require_cuda(weight, *xs)
        xs = [_f32c(x) for x in xs]
        B, _, H, W = xs[0].shape
        O = weight.shape[0]
        w2 = weight.reshape(O, -1)
        # MODEL.FP16 (the bf16 precision scope): one launch of csrc/conv1x1_bf16.inc over all inputs, the backward on the same file
        # (from 7 680 pixels per launch on: below, a workgroup's channel loop is a serial chain the library's split GEMMs beat --
        # 512+512+256 -> 512 @ 12x40 x 8: 96 us against 77)
        # Own pointwise kernels (one launch over all inputs, the backward on the same files): csrc/conv1x1_bf16.inc inside the bf16
        # precision scope, csrc/conv1x1_f32.inc (exact fp32, round 6) otherwise.  The gate covers the BACKWARD's launches too (the
        # input gradient runs the same kernel transposed with grad_output as its input: O % 16, O HW < 2^29; the weight gradient:
        # B max(O, C) HW < 2^29), so a layer that passes here never meets DCD_ERR_BAD_ARG inside backward.
        HW = H * W
        shapes_ok = (len(xs) <= 4 and HW % 4 == 0 and all(x.shape[1] % 16 == 0 for x in xs) and O % 16 == 0
                     and weight.dtype == torch.float32 and w2.is_contiguous() and O * HW < (1 << 29)
                     and all(B * max(O, x.shape[1]) * HW < (1 << 29) for x in xs))
        bf16 = _conv_prec(xs[0]) == PREC_BF16
        ctx.pw = None
        if shapes_ok and bf16 and _PW_BF16 and B * HW >= _PW_MIN_PIXELS:
            ctx.pw = "bf16"
        elif shapes_ok and not bf16 and _PW_F32 and B * HW >= _PW_F32_MIN_PIXELS and O * w2.shape[1] <= _PW_F32_MAX_WEIGHTS:
            ctx.pw = "f32"
        if ctx.pw:
            out = _pw_conv(ctx.pw, w2, 0, False, xs, O)
            ctx.save_for_backward(w2, *xs)
            ctx.wshape = weight.shape
            return out
        out = torch.empty((B, O, H, W), dtype=torch.float32, device=xs[0].device)
        o3 = out.view(B, O, H * W)
        c0 = 0
        for i, x in enumerate(xs):
            Ci = x.shape[1]
            wi = w2[:, c0:c0 + Ci].unsqueeze(0).expand(B, O, Ci)
            if i == 0:
                torch.bmm(wi, x.view(B, Ci, H * W), out=o3)
            else:
                o3.baddbmm_(wi, x.view(B, Ci, H * W))
            c0 += Ci
        ctx.save_for_backward(w2, *xs)
        ctx.wshape = weight.shape
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        w2, *xs = ctx.saved_tensors
        g = _f32c(g)
        B, O, H, W = g.shape
        if ctx.pw:
            L = _lib.lib()
            wrw_bytes, wrw = ((L.dcd_conv1x1_wrw_bf16_workspace_bytes, L.dcd_conv1x1_wrw_bf16) if ctx.pw == "bf16" else
                              (L.dcd_conv1x1_wrw_f32_workspace_bytes, L.dcd_conv1x1_wrw_f32))
            gw = torch.empty_like(w2)
            Ct, HW = w2.shape[1], H * W
            gxs, c0 = [], 0
            for i, x in enumerate(xs):
                Ci = x.shape[1]
                gxs.append(_pw_conv(ctx.pw, w2, c0, True, [g], Ci) if ctx.needs_input_grad[1 + i] else None)
                if ctx.needs_input_grad[0]:
                    n = wrw_bytes(B, O, Ci, HW)
                    ws = torch.empty(max(n, 16), dtype=torch.uint8, device=g.device)
                    _lib.check(wrw(_lib.stream_of(g), g.data_ptr(), x.data_ptr(), gw.data_ptr() + 4 * c0, Ct, B, O, Ci, HW,
                                   ws.data_ptr(), n), "dcd_conv1x1_wrw_" + ctx.pw)
                c0 += Ci
            return ((gw.reshape(ctx.wshape) if ctx.needs_input_grad[0] else None),) + tuple(gxs)
        g3 = g.view(B, O, H * W)
        gw = torch.empty_like(w2)
        gxs, c0 = [], 0
        for i, x in enumerate(xs):
            Ci = x.shape[1]
            x3 = x.view(B, Ci, H * W)
            if ctx.needs_input_grad[1 + i]:
                gxs.append(torch.bmm(w2[:, c0:c0 + Ci].t().unsqueeze(0).expand(B, Ci, O), g3).view(B, Ci, H, W))
            else:
                gxs.append(None)
            # transposed VIEW of x: no copy; the batch sum lands in the weight gradient's column slice directly
            torch.sum(torch.bmm(g3, x3.transpose(1, 2)), 0, out=gw[:, c0:c0 + Ci])
            c0 += Ci
        return (gw.reshape(ctx.wshape),) + tuple(gxs)


_PW_BF16 = os.environ.get("DCD_CONV1X1_BF16", "1") != "0"        # 0: the 1x1 convolutions stay fp32 library GEMMs under MODEL.FP16 (A/B)
_PW_MIN_PIXELS = int(os.environ.get("DCD_CONV1X1_BF16_MIN_PIXELS", "7680"))
_PW_F32 = os.environ.get("DCD_CONV1X1_F32", "1") != "0"          # 0: exact fp32 keeps the batched library GEMMs (A/B, round 5's path)
_PW_F32_MIN_PIXELS = int(os.environ.get("DCD_CONV1X1_F32_MIN_PIXELS", "7680"))
# In fp32 the pointwise kernels win where the layer is bound by bytes or by skinny GEMM shapes (<= 128 outputs, the projections);
# the 256-output Roots on the 24x80 maps are matrix-bound and the launch's 128 workgroups half-fill the chip: library GEMMs there
# (91 against 57 us forward, profiles/r06_conv1x1_f32.txt).  The bound is on outputs x concatenated inputs.
_PW_F32_MAX_WEIGHTS = int(os.environ.get("DCD_CONV1X1_F32_MAX_WEIGHTS", "65536"))


def _pw_conv(kind, w2, col0, transposed, xs, M):
    """out (B, M, H, W) = A . cat(xs) on csrc/conv1x1_bf16.inc (`kind` "bf16") or csrc/conv1x1_f32.inc ("f32"): A = w2 (M = rows), or
    w2[:, col0:col0 + M]^T (`transposed`: the input gradient of that column slice, xs = [grad_output])."""
    import ctypes
    L = _lib.lib()
    B, _, H, W = xs[0].shape
    n = len(xs)
    ptrs = (ctypes.c_void_p * n)(*[x.data_ptr() for x in xs])
    chs = (ctypes.c_int * n)(*[x.shape[1] for x in xs])
    out = torch.empty((B, M, H, W), dtype=torch.float32, device=xs[0].device)
    fn = L.dcd_conv1x1_bf16 if kind == "bf16" else L.dcd_conv1x1_f32
    st = fn(_lib.stream_of(xs[0]), w2.data_ptr() + 4 * col0, w2.shape[1], 1 if transposed else 0, n, ptrs, chs, out.data_ptr(), B, M, H * W)
    _lib.check(st, "dcd_conv1x1_" + kind)
    return out


def conv1x1_of_cat(xs, weight):
    """conv2d(torch.cat(xs, 1), weight (O, sum C_i, 1, 1)) for (B, C_i, H, W) tensors, without forming the concatenation."""
    return _Conv1x1OfCat.apply(weight, *xs)


class _Conv1dK3Replicate(torch.autograd.Function):
    """conv1d(x, weight (O, C, 3), bias, padding 1, padding_mode 'replicate') on (B, C, K) rows as ONE batched GEMM over the
    unfolded row -- the first layer of the two edge-fusion branches (DGDE/model/head/detector_predictor.py:124-131: Conv1d 256 ->
    256 along the <= 832 border cells).  MIOpen runs this 2.6 GFLOP layer as an NHWC implicit GEMM between four layout transposes
    (60 / 45 / 53 us forward / input gradient / weight gradient at bs 8 plus ~10 small launches each way); here: pad, one copy
    into (B, 3C, K) patch rows [index c * 3 + t, the order of weight.view(O, -1)], one GEMM with the bias as its addend; backward
    = two GEMMs, the fold of the patch gradient (col2im) and the padding's adjoint."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        B, C, K = x.shape
        O = weight.shape[0]
        xp = torch.nn.functional.pad(x, (1, 1), mode="replicate")                       # (B, C, K + 2)
        cols = xp.unfold(2, 3, 1).permute(0, 1, 3, 2).reshape(B, 3 * C, K)               # one copy
        w2 = weight.reshape(O, 3 * C)
        out = torch.baddbmm(bias.view(1, O, 1), w2.unsqueeze(0).expand(B, O, 3 * C), cols)
        ctx.save_for_backward(cols, w2)
        ctx.geom = (B, C, K, O, tuple(weight.shape))
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        cols, w2 = ctx.saved_tensors
        B, C, K, O, wshape = ctx.geom
        g = g.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gcols = torch.bmm(w2.t().unsqueeze(0).expand(B, 3 * C, O), g)                # (B, 3C, K)
            gxp = torch.nn.functional.fold(gcols, (1, K + 2), (1, 3)).view(B, C, K + 2)  # sum over the three taps
            gx = gxp[:, :, 1:K + 1].clone()                                              # adjoint of the replicate padding
            gx[:, :, 0] += gxp[:, :, 0]
            gx[:, :, K - 1] += gxp[:, :, K + 1]
        if ctx.needs_input_grad[1]:
            gw = torch.bmm(g, cols.transpose(1, 2)).sum(0).view(wshape)
        if ctx.needs_input_grad[2]:
            gb = channel_sums(g) if g.is_cuda and g.dtype == torch.float32 else g.sum((0, 2))     # (two-stage sums: safe inside a replayed graph)
        return gx, gw, gb


def conv1d_k3_replicate(x, weight, bias):
    return _Conv1dK3Replicate.apply(x, weight, bias)


def scatter_add_at(fmap, vals, index):
    """fmap (B,C,H,W) += vals (B,M,C) at the linear cell indices index (B,M); returns fmap (updated in place)."""
    return _ScatterAddAt.apply(fmap, vals, index)


def spd_buffer(b, n, device):
    """(b, n + 4, n) float32 buffer for `spd_solve_inplace`: rows 0..n-1 take the matrix, row n the right-hand side."""
    return torch.empty((b, n + 4, n), dtype=torch.float32, device=device)


def spd_solve_inplace(aug):
    """y[b] = S[b]^-1 r[b] with S = aug[b, :n], r = aug[b, n] (see `spd_buffer`); aug is overwritten (Cholesky factor / L^-1 r).
    Blocked Cholesky + substitutions on csrc/spd.hip; no host synchronisation: a matrix that is not positive definite yields a
    NaN y (the kernel poisons the factor at the first non-positive pivot) where torch.linalg.cholesky would raise, so the
    train step's non-finite guard skips the update."""
    _lib.require_cuda(aug)
    b, rows, n = aug.shape
    if aug.dtype != torch.float32 or not aug.is_contiguous() or rows < n + 1:
        raise RuntimeError("spd_solve_inplace: a contiguous float32 (b, >= n + 1, n) buffer is required")
    L = _lib.lib()
    nbytes = L.dcd_spd_solve_workspace_bytes(b, n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=aug.device)
    y = torch.empty((b, n), dtype=torch.float32, device=aug.device)
    st = L.dcd_spd_solve(_lib.stream_of(aug), aug.data_ptr(), y.data_ptr(), b, n, rows, None, ws.data_ptr(), nbytes)
    _lib.check(st, "dcd_spd_solve")
    return y


def spd_solve(S, rhs):
    """y[b] = S[b]^-1 rhs[b] for symmetric positive definite S (b, n, n) fp32 with n % 4 == 0 and rhs (b, n); S is left intact."""
    _lib.require_cuda(S, rhs)
    if S.dtype != torch.float32 or S.dim() != 3 or S.shape[1] != S.shape[2]:
        raise RuntimeError("spd_solve: S must be a float32 (b, n, n) tensor")
    b, n = S.shape[0], S.shape[1]
    aug = spd_buffer(b, n, S.device)
    aug[:, :n] = S
    aug[:, n] = rhs.reshape(b, n)
    return spd_solve_inplace(aug)


def schur_lower(G, inv_rows, cols, out):
    """out[b, :n, :n] (lower triangle and the tiles on the diagonal) = diag(cols[b]) - G[b]^T diag(inv_rows[b]) G[b] for
    G (b, m, n) fp32 (a view whose rows are n apart), on our own GEMM with the upper tiles skipped."""
    _lib.require_cuda(G, inv_rows, cols, out)
    b, m, n = G.shape
    if G.stride(2) != 1 or G.stride(1) != n or (G.stride(0) & 3) or (G.data_ptr() & 15):
        raise RuntimeError("schur_lower: G must be row-contiguous with row stride n")
    DG = (inv_rows.unsqueeze(-1) * G).contiguous()
    st = _lib.lib().dcd_sgemm(_lib.stream_of(G), G.data_ptr(), n, G.stride(0), 0, DG.data_ptr(), n, m * n, 0,
                              out.data_ptr(), n, out.stride(0), n, n, m, b, -1.0, 0, 1)
    _lib.check(st, "dcd_sgemm")
    out[:, :n].diagonal(dim1=-2, dim2=-1).add_(cols)
    return out


def iou_3d(pred_corners, target_corners):
    """(N,8,3) x (N,8,3) -> (N) 3-D IoU (BEV rectangle overlap x height overlap); no gradient."""
    _lib.require_cuda(pred_corners, target_corners)
    a, b = _f32c(pred_corners.detach()), _f32c(target_corners.detach())
    N = a.shape[0]
    out = torch.empty(N, dtype=torch.float32, device=a.device)
    st = _lib.lib().dcd_iou3d(_lib.stream_of(a), a.data_ptr(), b.data_ptr(), N, out.data_ptr())
    _lib.check(st, "dcd_iou3d")
    return out


# ----------------------------------------------------------------------------------------------
# Batch norm (+ residual) (+ ReLU), training mode, optionally synchronised over a process group
# ----------------------------------------------------------------------------------------------
def _bn_ws(C, dev):
    return torch.empty(_lib.lib().dcd_bn_workspace_bytes(C), dtype=torch.uint8, device=dev)


_BN_MASK_FROM_X = os.environ.get("DCD_BN_MASK_FROM_X", "1") != "0"     # 0: the backward reads the forward output for the ReLU mask (A/B timing)


class _BatchNormAct(torch.autograd.Function):
    """y = act(batch_norm(x) [+ residual]) on the HIP kernels of csrc/norm.hip (two launches forward, two backward;
    the stock chain is BN + add + ReLU = 3 kernels and 3 extra tensor round trips).  With `group` the per-channel fp64
    sums are all-reduced over RCCL between the two launches (SyncBatchNorm semantics; every rank must hold the same
    number of elements, which the reference's equal per-rank batches guarantee, DGDE/data/build.py:63-67)."""

    @staticmethod
    def forward(ctx, x, residual, weight, bias, running_mean, running_var, num_batches_tracked, momentum, eps, relu, group):
        _lib.require_cuda(x, residual, weight, bias)
        L = _lib.lib()
        x = _f32c(x)
        residual = None if residual is None else _f32c(residual)
        B, C = x.shape[0], x.shape[1]
        HW = x.numel() // (B * C)
        dev, st = x.device, _lib.stream_of(x)
        ws = _bn_ws(C, dev)
        count = float(B * HW)
        y = torch.empty_like(x)
        save_mean = torch.empty(C, dtype=torch.float32, device=dev)
        save_invstd = torch.empty(C, dtype=torch.float32, device=dev)
        if group is None:
            _lib.check(L.dcd_bn_train_forward(st, x.data_ptr(), _lib.ptr(residual), _lib.ptr(weight), _lib.ptr(bias),
                                              _lib.ptr(running_mean), _lib.ptr(running_var), _lib.ptr(num_batches_tracked),
                                              float(momentum), float(eps), int(bool(relu)), y.data_ptr(), save_mean.data_ptr(),
                                              save_invstd.data_ptr(), B, C, HW, ws.data_ptr(), ws.numel()), "dcd_bn_train_forward")
        else:
            import torch.distributed as dist
            stats = torch.empty((C, 2), dtype=torch.float64, device=dev)
            _lib.check(L.dcd_bn_stats(st, x.data_ptr(), B, C, HW, stats.data_ptr(), ws.data_ptr(), ws.numel()), "dcd_bn_stats")
            dist.all_reduce(stats, group=group)
            count *= dist.get_world_size(group)
            _lib.check(L.dcd_bn_train_apply(st, x.data_ptr(), _lib.ptr(residual), _lib.ptr(weight), _lib.ptr(bias), stats.data_ptr(),
                                            count, _lib.ptr(running_mean), _lib.ptr(running_var), _lib.ptr(num_batches_tracked),
                                            float(momentum), float(eps), int(bool(relu)), y.data_ptr(), save_mean.data_ptr(),
                                            save_invstd.data_ptr(), B, C, HW), "dcd_bn_train_apply")
        # ReLU fused, no residual, local statistics: the backward recomputes the mask y > 0 from x (dcd_bn_backward_relu_from_x) and
        # does not read y at all -- bias is kept instead of the output
        ctx.mask_from_x = bool(relu) and residual is None and _BN_MASK_FROM_X
        if ctx.mask_from_x:
            ctx.save_for_backward(x, bias, weight, save_mean, save_invstd)
        else:
            ctx.save_for_backward(x, y if relu else None, weight, save_mean, save_invstd)
        ctx.count, ctx.group, ctx.has_res = count, group, residual is not None
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        x, y, weight, save_mean, save_invstd = ctx.saved_tensors
        L = _lib.lib()
        gy = _f32c(gy)
        B, C = x.shape[0], x.shape[1]
        HW = x.numel() // (B * C)
        dev, st = x.device, _lib.stream_of(x)
        ws = _bn_ws(C, dev)
        gx = torch.empty_like(x)
        if ctx.mask_from_x:
            bias = y                                   # (the second saved tensor is the bias on this path)
            gw = torch.empty(C, dtype=torch.float32, device=dev)
            gb = torch.empty(C, dtype=torch.float32, device=dev)
            if ctx.group is not None:                  # statistics, all-reduce, apply
                import torch.distributed as dist
                sums = torch.empty((C, 2), dtype=torch.float64, device=dev)
                _lib.check(L.dcd_bn_backward_stats_params_relu_from_x(st, gy.data_ptr(), x.data_ptr(), _lib.ptr(weight), _lib.ptr(bias),
                                                                      save_mean.data_ptr(), save_invstd.data_ptr(), B, C, HW, sums.data_ptr(),
                                                                      gw.data_ptr(), gb.data_ptr(), ws.data_ptr(), ws.numel()),
                           "dcd_bn_backward_stats_params_relu_from_x")
                dist.all_reduce(sums, group=ctx.group)
                _lib.check(L.dcd_bn_backward_apply_relu_from_x(st, gy.data_ptr(), x.data_ptr(), _lib.ptr(weight), _lib.ptr(bias),
                                                               save_mean.data_ptr(), save_invstd.data_ptr(), sums.data_ptr(), ctx.count,
                                                               gx.data_ptr(), B, C, HW), "dcd_bn_backward_apply_relu_from_x")
                return (gx, None, gw if weight is not None else None, gb if weight is not None else None,
                        None, None, None, None, None, None, None)
            _lib.check(L.dcd_bn_backward_relu_from_x(st, gy.data_ptr(), x.data_ptr(), _lib.ptr(weight), _lib.ptr(bias), save_mean.data_ptr(),
                                                     save_invstd.data_ptr(), gx.data_ptr(), gw.data_ptr(), gb.data_ptr(), B, C, HW,
                                                     ws.data_ptr(), ws.numel()), "dcd_bn_backward_relu_from_x")
            return (gx, None, gw if weight is not None else None, gb if weight is not None else None,
                    None, None, None, None, None, None, None)
        want_res = ctx.has_res and ctx.needs_input_grad[1]
        gres = None
        if want_res:
            gres = torch.empty_like(x) if y is not None else gy      # no ReLU: d(residual) is grad_y itself
        gres_ptr = gres.data_ptr() if (want_res and y is not None) else None
        if ctx.group is None:
            gw = torch.empty(C, dtype=torch.float32, device=dev)
            gb = torch.empty(C, dtype=torch.float32, device=dev)
            _lib.check(L.dcd_bn_backward(st, gy.data_ptr(), _lib.ptr(y), x.data_ptr(), _lib.ptr(weight), save_mean.data_ptr(),
                                         save_invstd.data_ptr(), gx.data_ptr(), gres_ptr, gw.data_ptr(), gb.data_ptr(), B, C, HW,
                                         ws.data_ptr(), ws.numel()), "dcd_bn_backward")
        else:
            # weight / bias gradients stay local (DDP averages them); the input gradient needs the global sums
            import torch.distributed as dist
            sums = torch.empty((C, 2), dtype=torch.float64, device=dev)
            gw = torch.empty(C, dtype=torch.float32, device=dev)
            gb = torch.empty(C, dtype=torch.float32, device=dev)
            # this rank's parameter gradients come out of the same launch(es) as the sums (they were three ATen ops per layer)
            _lib.check(L.dcd_bn_backward_stats_params(st, gy.data_ptr(), _lib.ptr(y), x.data_ptr(), save_mean.data_ptr(),
                                                      save_invstd.data_ptr(), B, C, HW, sums.data_ptr(), gw.data_ptr(), gb.data_ptr(),
                                                      ws.data_ptr(), ws.numel()), "dcd_bn_backward_stats_params")
            dist.all_reduce(sums, group=ctx.group)
            _lib.check(L.dcd_bn_backward_apply(st, gy.data_ptr(), _lib.ptr(y), x.data_ptr(), _lib.ptr(weight), save_mean.data_ptr(),
                                               save_invstd.data_ptr(), sums.data_ptr(), ctx.count, gx.data_ptr(), gres_ptr, None, None,
                                               B, C, HW), "dcd_bn_backward_apply")
        return (gx, gres, gw if weight is not None else None, gb if weight is not None else None,
                None, None, None, None, None, None, None)


def batch_norm_act(x, residual, weight, bias, running_mean, running_var, num_batches_tracked, momentum, eps, relu,
                   group=None):
    """Training-mode BN (+residual) (+ReLU); updates the running buffers in place like F.batch_norm(training=True)."""
    return _BatchNormAct.apply(x, residual, weight, bias, running_mean, running_var, num_batches_tracked, momentum, eps,
                               relu, group)


def batch_norm_act_eval(x, residual, weight, bias, running_mean, running_var, eps, relu):
    """Inference-mode BN (+residual) (+ReLU) with the running statistics; no gradient."""
    _lib.require_cuda(x, residual)
    x = _f32c(x.detach())
    residual = None if residual is None else _f32c(residual.detach())
    B, C = x.shape[0], x.shape[1]
    HW = x.numel() // (B * C)
    y = torch.empty_like(x)
    st = _lib.lib().dcd_bn_eval_apply(_lib.stream_of(x), x.data_ptr(), _lib.ptr(residual), _lib.ptr(weight), _lib.ptr(bias),
                                      running_mean.data_ptr(), running_var.data_ptr(), float(eps), int(bool(relu)),
                                      y.data_ptr(), B, C, HW)
    _lib.check(st, "dcd_bn_eval_apply")
    return y


# ----------------------------------------------------------------------------------------------
# 3x3 stride-1 convolution (Winograd F(2x2,3x3) on the matrix pipe): forward + input gradient
# ----------------------------------------------------------------------------------------------
_CONV_MIN_MAP = int(os.environ.get("DCD_CONV_MIN_MAP", str(12 * 40)))      # A/B timing: 7680 restores the round-1 dispatch


def conv3x3_supported(x, weight):
    """Shapes the HIP kernel is used for: 3x3 weight, W % 4 == 0, H even, at least 64 input and output channels (narrower
    layers would idle most of a 64-wide output slice), maps of at least 12x40 (the 12 x 20 px regions and the split
    contraction of csrc/conv.hip keep every CU busy there; tools/time_conv.py)."""
    return (x.is_cuda and x.dtype == torch.float32 and weight.dim() == 4 and weight.shape[2] == 3 and weight.shape[3] == 3
            and x.shape[3] % 4 == 0 and x.shape[2] % 2 == 0 and weight.shape[0] >= 64 and weight.shape[1] >= 64
            and x.shape[2] * x.shape[3] >= _CONV_MIN_MAP)


_CONV_SPLIT_MIN_MAP = int(os.environ.get("DCD_CONV_SPLIT_MIN_MAP", "7680"))
_CONV_BF16_MIN_MAP = int(os.environ.get("DCD_CONV_BF16_MIN_MAP", "0"))
PREC_F32, PREC_BF16X3, PREC_BF16 = 0, 1, 2


def _conv_prec(like=None):
    """Precision of a 3x3 convolution's Winograd-domain products (a DCD_PREC_* value), from `_ext.get_precision()`:
      "bf16x3"  split-bf16 (what `bench.py --precision bf16x3` sets) -- the split kernel has the 8 x 32 px regions only and pays
                from 48 x 160 maps on (64->64 @ 96x320 123 -> 83 us, 128->128 @ 48x160 101 -> 69; 256->256 @ 24x80 95 -> 120:
                those stay on the fp32 kernel with its exact-cover 12 x 20 regions);
      "bf16"    one product of bf16-rounded operands (MODEL.FP16) on the same kernel, on every map of at least
                DCD_CONV_BF16_MIN_MAP pixels;
    DCD_CONV_SPLIT=0 keeps the convolutions exact fp32 in either mode (A/B timing).  like: the call's input."""
    from . import _ext
    p = _ext.get_precision()
    if p == "f32" or os.environ.get("DCD_CONV_SPLIT", "1") == "0":
        return PREC_F32
    hw = None if like is None else like.shape[2] * like.shape[3]
    if p == "bf16x3":
        return PREC_BF16X3 if hw is None or hw >= _CONV_SPLIT_MIN_MAP else PREC_F32
    return PREC_BF16 if hw is None or hw >= _CONV_BF16_MIN_MAP else PREC_F32


def _conv_split(like=None):
    return _conv_prec(like) != PREC_F32


def _wrw_prec(prec):
    """The weight-gradient kernel has the exact and the one-product form (no split one)."""
    return PREC_BF16 if prec == PREC_BF16 else PREC_F32


class SplitWeights:
    """Winograd-domain weights in split-bf16 layout (dcd_conv3x3_split_transform_weights); a marker type so that the call knows
    which kernel they belong to, and with which precision (DCD_PREC_BF16X3 / DCD_PREC_BF16: the same layout) it runs."""

    def __init__(self, tensor, prec):
        self.tensor = tensor
        self.prec = prec


class Bf16Weights(SplitWeights):
    """bf16-rounded weights in the operand order of the direct one-product kernel (dcd_conv3x3_bf16_transform_weights,
    csrc/conv_direct_bf16.inc): what DCD_PREC_BF16 runs on unless DCD_CONV_BF16_DIRECT=0 keeps the Winograd form (A/B timing)."""

    def __init__(self, tensor):
        super().__init__(tensor, PREC_BF16)


_BF16_DIRECT = os.environ.get("DCD_CONV_BF16_DIRECT", "1") != "0"


def _layout(prec):
    """Which prepared-weight layout a precision runs on: "f32" (Winograd, fp32), "split" (Winograd, hi | lo bf16), "bf16" (direct)."""
    if prec == PREC_F32:
        return "f32"
    return "bf16" if prec == PREC_BF16 and _BF16_DIRECT else "split"


def conv3x3_transform_weights(weight, forward=True, backward=True, like=None, prec=None):
    """Winograd-domain weights of a (Cout, Cin, 3, 3) filter for the forward and / or the backward-data call, ONE launch
    (split-bf16 form: one launch per direction).  like: the tensor the convolution will run on (decides the form, see _conv_prec);
    prec: a DCD_PREC_* value that overrides it."""
    L = _lib.lib()
    Co, Ci = weight.shape[0], weight.shape[1]
    prec = _conv_prec(like) if prec is None else prec
    if _layout(prec) == "bf16":
        tf = torch.empty(L.dcd_conv3x3_bf16_weights_bytes(Ci, Co, 0) // 4, dtype=torch.int32, device=weight.device) if forward else None
        tb = torch.empty(L.dcd_conv3x3_bf16_weights_bytes(Ci, Co, 1) // 4, dtype=torch.int32, device=weight.device) if backward else None
        _lib.check(L.dcd_conv3x3_bf16_transform_weights(_lib.stream_of(weight), weight.data_ptr(), Ci, Co, _lib.ptr(tf), _lib.ptr(tb)),
                   "dcd_conv3x3_bf16_transform_weights")
        return (Bf16Weights(tf) if forward else None), (Bf16Weights(tb) if backward else None)
    if prec != PREC_F32:
        tf = torch.empty(L.dcd_conv3x3_split_weights_bytes(Ci, Co, 0) // 4, dtype=torch.int32, device=weight.device) if forward else None
        tb = torch.empty(L.dcd_conv3x3_split_weights_bytes(Ci, Co, 1) // 4, dtype=torch.int32, device=weight.device) if backward else None
        _lib.check(L.dcd_conv3x3_split_transform_weights(_lib.stream_of(weight), weight.data_ptr(), Ci, Co, _lib.ptr(tf), _lib.ptr(tb)),
                   "dcd_conv3x3_split_transform_weights")
        return (SplitWeights(tf, prec) if forward else None), (SplitWeights(tb, prec) if backward else None)
    tf = torch.empty(L.dcd_conv3x3_weights_bytes(Ci, Co, 0) // 4, dtype=torch.float32, device=weight.device) if forward else None
    tb = torch.empty(L.dcd_conv3x3_weights_bytes(Ci, Co, 1) // 4, dtype=torch.float32, device=weight.device) if backward else None
    _lib.check(L.dcd_conv3x3_transform_weights(_lib.stream_of(weight), weight.data_ptr(), Ci, Co, _lib.ptr(tf), _lib.ptr(tb)),
               "dcd_conv3x3_transform_weights")
    return tf, tb


def _conv3x3_call(inp, weight, out_channels, backward_data, bias=None, residual=None, transformed=None, residual_inplace=True,
                  prec=None):
    """residual: a tensor of the output's shape that the result is added to -- IN PLACE (and returned) unless residual_inplace
    is False (then it is only read: a gradient the caller does not own).
    transformed: this direction's weights from conv3x3_transform_weights (else they are transformed inside the call).
    prec: DCD_PREC_* of a call that transforms its own weights (default: `_conv_prec(inp)`, i.e. the current precision scope --
    a backward call passes what its forward ran with)."""
    L = _lib.lib()
    B, _, H, W = inp.shape
    Co, Ci = weight.shape[0], weight.shape[1]
    if residual is not None:
        if tuple(residual.shape) != (B, out_channels, H, W) or residual.dtype != torch.float32 or not residual.is_contiguous():
            raise RuntimeError("conv3x3: residual must be a contiguous fp32 tensor of the output's shape")
        out = residual if residual_inplace else torch.empty_like(residual)
    else:
        out = torch.empty((B, out_channels, H, W), dtype=torch.float32, device=inp.device)
    if transformed is None:
        prec = _conv_prec(inp) if prec is None else prec
        if prec != PREC_F32:
            transformed = conv3x3_transform_weights(weight, not backward_data, backward_data, prec=prec)[1 if backward_data else 0]
    if isinstance(transformed, Bf16Weights):
        n = L.dcd_conv3x3_bf16_workspace_bytes(B, Ci, H, W, Co)
        ws = torch.empty(max(n, 16), dtype=torch.uint8, device=inp.device)
        st = L.dcd_conv3x3_bf16_prepared(_lib.stream_of(inp), inp.data_ptr(), transformed.tensor.data_ptr(), _lib.ptr(bias),
                                         _lib.ptr(residual), out.data_ptr(), B, Ci, H, W, Co, 1 if backward_data else 0, ws.data_ptr(), n)
        _lib.check(st, "dcd_conv3x3_bf16_prepared")
        return out
    if isinstance(transformed, SplitWeights):
        n = L.dcd_conv3x3_split_workspace_bytes(B, Ci, H, W, Co)
        ws = torch.empty(max(n, 16), dtype=torch.uint8, device=inp.device)
        st = L.dcd_conv3x3_split_prepared(_lib.stream_of(inp), inp.data_ptr(), transformed.tensor.data_ptr(), _lib.ptr(bias),
                                          _lib.ptr(residual), out.data_ptr(), B, Ci, H, W, Co, 1 if backward_data else 0,
                                          transformed.prec, ws.data_ptr(), n)
        _lib.check(st, "dcd_conv3x3_split_prepared")
        return out
    n = L.dcd_conv3x3_workspace_bytes(B, Ci, H, W, Co)
    if transformed is not None:
        n = max(n - min(L.dcd_conv3x3_weights_bytes(Ci, Co, 0), L.dcd_conv3x3_weights_bytes(Ci, Co, 1)), 0)   # partial images only (upper bound)
        ws = torch.empty(max(n, 16), dtype=torch.uint8, device=inp.device)
        st = L.dcd_conv3x3_prepared(_lib.stream_of(inp), inp.data_ptr(), transformed.data_ptr(), _lib.ptr(bias), _lib.ptr(residual),
                                    out.data_ptr(), B, Ci, H, W, Co, 1 if backward_data else 0, ws.data_ptr(), n)
        _lib.check(st, "dcd_conv3x3_prepared")
        return out
    ws = torch.empty(n, dtype=torch.uint8, device=inp.device)
    st = L.dcd_conv3x3(_lib.stream_of(inp), inp.data_ptr(), weight.data_ptr(), _lib.ptr(bias), _lib.ptr(residual), out.data_ptr(),
                       B, Ci, H, W, Co,
                       1 if backward_data else 0, ws.data_ptr(), n)
    _lib.check(st, "dcd_conv3x3")
    return out


_PREP_BOTH = os.environ.get("DCD_CONV_PREP_BOTH", "1") != "0"      # 0: every call transforms its own weights (A/B timing)
_PREP_TABLE = os.environ.get("DCD_CONV_PREP_TABLE", "1") != "0"    # 0: one transform launch per layer and step (A/B timing)


class _PreparedWeights:
    """Winograd-domain weights of every 3x3 convolution a train step has run, transformed by ONE launch per step
    (`refresh_conv_weights`, called by the trainer right before the forward) instead of one launch per layer: 37 launches
    of ~5 us at DGDE's size, which is 0.2 ms of a 14 ms one-image step.

    A layer enters the table the first time `_Conv3x3.forward` sees its weight (that call still transforms its own copy).
    An entry is only used while it is provably current: same storage, and `weight._version` equal to the version the last
    refresh transformed -- an optimizer step, a `load_state_dict` or any other in-place write in between sends the call back
    to its own transform (a write that BYPASSES the version counter -- `p.data.*`, a raw-pointer kernel, a replayed graph that
    is not GraphedTrainStep -- must be followed by `invalidate_conv_weights()`).  Entries hold weak references; a dead weight
    drops out at the next refresh.

    HIP graphs: a captured `refresh()` bakes the address of `self.table` into the graph, a captured convolution the
    addresses of its entry's buffers.  From the first lookup / refresh under capture on, nothing a capture may have seen is ever
    released: replaced tables and the buffers of removed entries move to `self._immortal` (a table rebuild happens when another
    model registers its convolutions or a parameter's storage moves -- a handful of times per process, a few MB each), so a live
    graph replays against memory that is still ours -- stale for a layer that no longer exists, never recycled."""

    def __init__(self, split=False):
        # entries hold: False the fp32 layout, True the split-bf16 one (DCD_PREC_BF16X3; DCD_PREC_BF16 on the Winograd form),
        # "bf16" the direct one-product kernel's
        self.split = split
        self.entries = {}          # id(weight) -> [weakref, data_ptr, forward buffer, backward buffer, version at refresh, idle]
        self.table = None          # device int64 (n, 5) the kernel reads
        self.order = []
        self.dirty = False
        self.captured = False      # a capture has read the table or an entry's buffers
        self._immortal = []

    def lookup(self, weight):
        e = self.entries.get(id(weight))
        if e is None or e[0]() is not weight or e[1] != weight.data_ptr() or e[4] != weight._version:
            return None
        if not self.captured and torch.cuda.is_current_stream_capturing():
            self.captured = True
        e[5] = 0                                       # in use
        return e

    def register(self, weight):
        import weakref
        if not _PREP_TABLE or torch.cuda.is_current_stream_capturing() or not isinstance(weight, torch.nn.Parameter):
            return                                     # parameters only: a cast / reshaped copy is a new object every step
        e = self.entries.get(id(weight))
        if e is not None and e[0]() is weight and e[1] == weight.data_ptr():
            return
        L = _lib.lib()
        Co, Ci = weight.shape[0], weight.shape[1]
        if self.split == "bf16":
            tf = torch.empty(L.dcd_conv3x3_bf16_weights_bytes(Ci, Co, 0) // 4, dtype=torch.int32, device=weight.device)
            tb = torch.empty(L.dcd_conv3x3_bf16_weights_bytes(Ci, Co, 1) // 4, dtype=torch.int32, device=weight.device)
        elif self.split:
            tf = torch.empty(L.dcd_conv3x3_split_weights_bytes(Ci, Co, 0) // 4, dtype=torch.int32, device=weight.device)
            tb = torch.empty(L.dcd_conv3x3_split_weights_bytes(Ci, Co, 1) // 4, dtype=torch.int32, device=weight.device)
        else:
            tf = torch.empty(L.dcd_conv3x3_weights_bytes(Ci, Co, 0) // 4, dtype=torch.float32, device=weight.device)
            tb = torch.empty(L.dcd_conv3x3_weights_bytes(Ci, Co, 1) // 4, dtype=torch.float32, device=weight.device)
        self.entries[id(weight)] = [weakref.ref(weight), weight.data_ptr(), tf, tb, -1, 0]
        self.dirty = True

    def refresh(self):
        if not self.entries or not _PREP_TABLE:
            return
        # an entry no convolution has looked up for three refreshes leaves the table (a model that changed its precision mode
        # would otherwise have BOTH layouts of every layer transformed every step); it re-enters at its next use
        dead = [k for k, e in self.entries.items() if e[0]() is None or e[0]().data_ptr() != e[1] or e[5] >= 3]
        if (dead or self.dirty) and torch.cuda.is_current_stream_capturing():
            return                                     # no table upload inside a capture: the calls transform their own weights
        if torch.cuda.is_current_stream_capturing():
            self.captured = True                       # this refresh (table address + every entry's buffers) is part of a graph
        for k in dead:
            if self.captured:
                self._immortal.append((self.entries[k][2], self.entries[k][3]))
            del self.entries[k]
            self.dirty = True
        if not self.entries:
            if self.captured and self.table is not None:
                self._immortal.append(self.table)
            self.table = None
            return
        if self.dirty or self.table is None:
            self.order = list(self.entries.values())
            dev = self.order[0][2].device
            rows = [[e[1], e[2].data_ptr(), e[3].data_ptr(), e[0]().shape[1], e[0]().shape[0]] for e in self.order]
            if self.captured and self.table is not None:
                self._immortal.append(self.table)      # a live graph may still launch the transform on the old table
            self.table = torch.tensor(rows, dtype=torch.int64).to(dev)
            self.dirty = False
        L = _lib.lib()
        fn = (L.dcd_conv3x3_bf16_transform_weights_table if self.split == "bf16" else
              L.dcd_conv3x3_split_transform_weights_table if self.split else L.dcd_conv3x3_transform_weights_table)
        _lib.check(fn(_lib.stream_of(self.table), self.table.data_ptr(), len(self.order)), "dcd_conv3x3_transform_weights_table")
        for e in self.order:
            e[4] = e[0]()._version
            e[5] += 1


_PREPARED = {}                     # (device index, layout: False fp32 | True split | "bf16" direct) -> _PreparedWeights
_LAYOUT_KEYS = (False, True, "bf16")


def invalidate_conv_weights(device=None):
    """Forget that the table's entries are current (the next `refresh_conv_weights` re-validates them).  For code that changes
    parameters without going through autograd's version counters -- a replayed HIP graph that contains the optimizer step."""
    if not torch.cuda.is_available():
        return
    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    for split in _LAYOUT_KEYS:
        p = _PREPARED.get((idx, split))
        if p is not None:
            for e in p.entries.values():
                e[4] = -1


def conv3x3_step_weights(weight, like):
    """(forward, backward-data) Winograd-domain weights for a call inside a train step: the step's table entry when it is
    current (no launch), else transformed now -- both directions, one launch -- and the layer is registered for the next
    `refresh_conv_weights`."""
    prec = _conv_prec(like)
    split = {"f32": False, "split": True, "bf16": "bf16"}[_layout(prec)]
    key = (like.device.index, split)
    prepared = _PREPARED.get(key)
    if prepared is None:
        prepared = _PREPARED[key] = _PreparedWeights(split)
    e = prepared.lookup(weight)
    if e is not None:
        if split == "bf16":
            return Bf16Weights(e[2]), Bf16Weights(e[3])
        return (SplitWeights(e[2], prec), SplitWeights(e[3], prec)) if split else (e[2], e[3])
    prepared.register(weight)
    return conv3x3_transform_weights(weight, prec=prec)


def refresh_conv_weights(device=None):
    """Transform the weights of every registered 3x3 convolution on `device` (default: the current one) in one launch, on the
    current stream.  Call it after the optimizer step / before the forward of a train step; calling it never is allowed (every
    convolution then transforms its own weights, as before)."""
    if not torch.cuda.is_available():
        return
    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    for split in _LAYOUT_KEYS:
        p = _PREPARED.get((idx, split))
        if p is not None:
            p.refresh()
_WRW_ENABLED = os.environ.get("DCD_CONV_WRW", "1") != "0"        # 0: weight gradient on the stock op (A/B timing)


def _conv3x3_wrw_call(x, gy, wshape, prec=PREC_F32):
    """prec: the DCD_PREC_* the layer's forward ran with (the one-product form exists, the split one runs exact)."""
    L = _lib.lib()
    B, Ci, H, W = x.shape
    Co = wshape[0]
    gw = torch.empty(tuple(wshape), dtype=torch.float32, device=x.device)
    n = L.dcd_conv3x3_wrw_workspace_bytes(B, Ci, H, W, Co)
    ws = torch.empty(n, dtype=torch.uint8, device=x.device)
    st = L.dcd_conv3x3_wrw(_lib.stream_of(x), x.data_ptr(), gy.data_ptr(), gw.data_ptr(), B, Ci, H, W, Co, _wrw_prec(prec),
                           ws.data_ptr(), n)
    _lib.check(st, "dcd_conv3x3_wrw")
    return gw


class _Conv3x3(torch.autograd.Function):
    """y = conv2d(x, w, stride 1, padding 1), dL/dx and dL/dw on csrc/conv.hip (Winograd F(2x2,3x3) on the fp32 matrix
    pipe; MIOpen's implicit-GEMM weight gradient runs near the fp32 matrix peak but does 2.25x the multiplies and needs
    NHWC transposes of both operands)."""

    @staticmethod
    def forward(ctx, x, weight):
        _lib.require_cuda(x, weight)
        x, weight = _f32c(x), _f32c(weight)
        ctx.save_for_backward(x, weight)
        ctx.tw_back = None
        ctx.prec = _conv_prec(x)                       # the backward runs outside the forward's precision scope
        if _PREP_BOTH and ctx.needs_input_grad[0]:
            # the weights of this call and of its backward-data call (they do not change in between)
            tw, ctx.tw_back = conv3x3_step_weights(weight, x)
            return _conv3x3_call(x, weight, weight.shape[0], False, transformed=tw)
        return _conv3x3_call(x, weight, weight.shape[0], False, prec=ctx.prec)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = _f32c(gy)
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = _conv3x3_call(gy, weight, weight.shape[1], True, transformed=ctx.tw_back, prec=ctx.prec)
        if ctx.needs_input_grad[1]:
            if _WRW_ENABLED:
                gw = _conv3x3_wrw_call(x, gy, weight.shape, ctx.prec)
            else:
                gw = torch.ops.aten.convolution_backward(gy, x, weight, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                         [False, True, False])[1]
        return gx, gw


def conv3x3(x, weight):
    return _Conv3x3.apply(x, weight)


class _Conv3x3Skip(torch.autograd.Function):
    """(conv3x3(x, w), x): the convolution and an alias of its input for a skip connection that starts at the same tensor
    (DLA's BasicBlock with an identity residual, dla_dcn.py:83-101: x feeds conv1 and the block's final addition).  As two
    consumers of x autograd adds their gradients in a pass of its own (read, read, write over the map); here the skip's
    gradient enters the input-gradient kernel as its `residual` and is added in the output transform."""

    @staticmethod
    def forward(ctx, x, weight):
        _lib.require_cuda(x, weight)
        x, weight = _f32c(x), _f32c(weight)
        ctx.save_for_backward(x, weight)
        ctx.prec = _conv_prec(x)
        tw, ctx.tw_back = conv3x3_step_weights(weight, x) if _PREP_BOTH and ctx.needs_input_grad[0] else (None, None)
        return _conv3x3_call(x, weight, weight.shape[0], False, transformed=tw, prec=ctx.prec), x.view_as(x)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy, gskip):
        x, weight = ctx.saved_tensors
        gx = gw = None
        if gy is None:                                      # only the skip was used
            return gskip, None
        gy = _f32c(gy)
        if ctx.needs_input_grad[0]:
            res = None if gskip is None else _f32c(gskip)   # read only: the gradient tensor belongs to autograd
            gx = _conv3x3_call(gy, weight, weight.shape[1], True, residual=res, transformed=ctx.tw_back, residual_inplace=False,
                               prec=ctx.prec)
        if ctx.needs_input_grad[1]:
            gw = (_conv3x3_wrw_call(x, gy, weight.shape, ctx.prec) if _WRW_ENABLED else
                  torch.ops.aten.convolution_backward(gy, x, weight, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1])
        return gx, gw


def conv3x3_with_skip(x, weight):
    """(conv3x3(x, weight), x) with the two gradients of x summed inside the input-gradient kernel."""
    return _Conv3x3Skip.apply(x, weight)


# ----------------------------------------------------------------------------------------------
# 3x3 / stride 2 / pad 1 convolution (the first convolution of every DLA level, DGDE/model/backbone/dla_dcn.py:76-78,313-326)
# on the stride-1 Winograd kernels, through space-to-depth
# ----------------------------------------------------------------------------------------------
_S2D_INDEX = {}
_S2D_MODE = os.environ.get("DCD_CONV_S2D", "auto")      # 1: always, 0: never (stock solver), auto: in the bf16 precision modes (and everywhere once a whole-step graph is built)


def stride2_on_own_kernels():
    """From now on the stride-2 3x3 convolutions of this process take the space-to-depth path on our kernels in exact fp32 too (what
    DCD_CONV_S2D=1 selects at start-up).  Called by `engine.trainer.GraphedTrainStep`: MIOpen's pick for these layers' input
    gradient at bs 8 can be a composable-kernel solver that zero-fills its output with hipMemsetAsync and accumulates -- a memset
    node in the captured step -- and no MIOPEN_DEBUG_* switch of the MIOpen build inside PyTorch
    turned it off reliably (tools/probes/miopen_env_probe.py); with the layers on our kernels the traced bs-8 graph has no
    memset and no MIOpen convolution left.  Costs the graphed fp32 step 0.4-0.6 ms (the regrouped filter's 4x multiplies)."""
    global _S2D_MODE
    prev = _S2D_MODE
    if _S2D_MODE == "auto":
        _S2D_MODE = "1"
    return prev


def restore_stride2_mode(prev):
    """Undo `stride2_on_own_kernels` (its return value): a `GraphedTrainStep` whose capture failed hands the process back to the
    eager step, which in exact fp32 is faster on the stock solver (2.72 against 3.75 ms per bs-8 step)."""
    global _S2D_MODE
    if prev in ("auto", "0", "1"):
        _S2D_MODE = prev


def stock_conv_in_capture(module, x):
    """Called by `layers.conv.Conv2d` right before it falls through to the stock convolution.  Inside a stream capture a stride-2
    3x3 layer must not get there: MIOpen's backward-data pick for these layers can be a solver that zero-fills its output with
    hipMemsetAsync and accumulates -- a memset node in the captured step, which on this stack returned wrong gradients in some
    replays.  `GraphedTrainStep` keeps them on our kernels (`stride2_on_own_kernels`), but only for the
    shapes `conv3x3_stride2_supported` takes (H % 4, W % 8, >= 16 input channels, a half-resolution map of >= 12 x 40): anything
    else raises here, the capture fails cleanly and the caller falls back to the eager step."""
    if (x.is_cuda and module.kernel_size == (3, 3) and module.stride == (2, 2) and torch.is_grad_enabled()
            and (x.requires_grad or module.weight.requires_grad) and torch.cuda.is_current_stream_capturing()):
        raise RuntimeError("stride-2 3x3 convolution %s on input %s would run on the stock solver inside a stream capture "
                           "(possible memset node: wrong gradients in replays); use the eager step for this input size"
                           % (tuple(module.weight.shape), tuple(x.shape)))


def _s2d_index(K, C, device):
    """Flat gather index of the (K, 4C, 3, 3) space-to-depth filter into [weight.flatten(), 0]: channel c * 4 + 2 rp + cp of the
    pixel-unshuffled input holds pixel (2 i + rp, 2 j + cp) of channel c, and tap a of the stride-2 filter reads row 2 i + a - 1:
    the odd row of s2d row i - 1 (a = 0), the even row of s2d row i (a = 1), the odd row of s2d row i (a = 2); likewise columns."""
    key = (K, C, str(device))
    pair = _S2D_INDEX.get(key)
    if pair is None:
        tap = {0: {1: 1}, 1: {0: 0, 1: 2}}                # parity -> {s2d tap A: original tap a}
        idx = torch.full((K, C, 2, 2, 3, 3), K * C * 9, dtype=torch.int64)
        inv = torch.zeros((K, C, 3, 3), dtype=torch.int64)    # where tap (a, b) of (k, c) sits in the regrouped filter
        base = torch.arange(K * C, dtype=torch.int64).view(K, C) * 9
        base4 = torch.arange(K * C, dtype=torch.int64).view(K, C) * 36
        for rp in (0, 1):
            for A, a in tap[rp].items():
                for cp in (0, 1):
                    for Bc, b in tap[cp].items():
                        idx[:, :, rp, cp, A, Bc] = base + a * 3 + b
                        inv[:, :, a, b] = base4 + ((rp * 2 + cp) * 3 + A) * 3 + Bc
        pair = _S2D_INDEX[key] = (idx.reshape(-1).to(device), inv.reshape(-1).to(device))
    return pair


class _S2DFilter(torch.autograd.Function):
    """(K, C, 3, 3) -> the (K, 4 C, 3, 3) space-to-depth filter: a gather forward, a gather backward (every original tap sits at
    exactly one regrouped position; autograd's own backward of an index gather is a sort-based scatter: 60 ms for 512 x 1024 x 9)."""

    @staticmethod
    def forward(ctx, weight):
        K, C = weight.shape[0], weight.shape[1]
        idx, inv = _s2d_index(K, C, weight.device)
        ctx.inv, ctx.shape = inv, weight.shape
        return torch.cat([weight.reshape(-1), weight.new_zeros(1)])[idx].view(K, 4 * C, 3, 3)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g4):
        return g4.reshape(-1)[ctx.inv].view(ctx.shape)


def conv3x3_stride2_supported(x, weight):
    """Stride-2 layers the space-to-depth form takes: even H, W % 8 == 0 (the half-resolution map must satisfy the stride-1 kernel's
    W % 4 == 0, H even -> H % 4 == 0), at least 64 channels after the regrouping (4 Cin) and a half-resolution map of 12 x 40."""
    if _S2D_MODE == "0" or not (x.is_cuda and x.dtype == torch.float32 and weight.dim() == 4 and tuple(weight.shape[2:]) == (3, 3)):
        return False
    if _S2D_MODE != "1" and _conv_prec() == PREC_F32:
        return False                                   # exact fp32 keeps the stock solver (2.72 against 3.75 ms per bs-8 step) unless
                                                       # a whole-step graph is in use (stride2_on_own_kernels)
    H, W = x.shape[2], x.shape[3]
    if _S2D_MODE == "1":
        # a process that captures whole-step graphs (or DCD_CONV_S2D=1): EVERY stride-2 layer with at least 16 input channels, at any
        # size -- maps off the alignment rules are zero-padded (conv3x3_stride2), small ones are slow but never on the stock solver
        # inside a capture (DLA levels 4 / 5 at 96x320, inputs whose deep maps are below 24x80)
        return 4 * weight.shape[1] >= 64
    return H % 4 == 0 and W % 8 == 0 and 4 * weight.shape[1] >= 64 and (H // 2) * (W // 2) >= _CONV_MIN_MAP


def conv3x3_stride2(x, weight):
    """conv2d(x, weight, stride 2, padding 1) = conv3x3(pixel_unshuffle(x, 2), W4) with the (Cout, 4 Cin, 3, 3) filter W4 that holds
    the nine taps at their space-to-depth positions and zeros elsewhere (27 of 36): forward, input gradient and weight gradient all
    on csrc/conv.hip (the stock path: MIOpen implicit GEMM in NHWC with a transpose of every operand, 3.5 ms per bs-8 step for the
    five layers).  The regrouped filter does 4x the multiplies of the direct form -- in the bf16 precision modes they run on matrix
    cores with 16x the fp32 rate and the kernels are bound by their transform arithmetic, i.e. by the 4 Cin channels at a quarter
    of the pixels = the cost of a stride-1 layer at full resolution.  Filter regrouping and its gradient are one gather / one
    scatter-add of 36 Cout Cin values (autograd)."""
    H, W = x.shape[2], x.shape[3]
    ph, pw = (-H) % 4, (-W) % 8
    if ph or pw:
        # zero rows / columns below and to the right: the outputs that exist for the original size read nothing but the zeros the
        # convolution's own padding would have supplied there; the extra outputs are cut off
        x = torch.nn.functional.pad(x, (0, pw, 0, ph))
    y = conv3x3(torch.nn.functional.pixel_unshuffle(x, 2), _S2DFilter.apply(weight))
    if ph or pw:
        y = y[:, :, :(H + 1) // 2, :(W + 1) // 2].contiguous()
    return y


# ----------------------------------------------------------------------------------------------
# ... and natively in exact fp32 (csrc/conv_s2_f32.inc, round 6): no space-to-depth copies, no 4x multiplies, no stock solver
# ----------------------------------------------------------------------------------------------
_S2_NATIVE = os.environ.get("DCD_CONV_S2_NATIVE", "1") != "0"          # 0: exact fp32 keeps the stock solver / the space-to-depth form (A/B)
_S2_NATIVE_MIN_PIXELS = int(os.environ.get("DCD_CONV_S2_NATIVE_MIN_PIXELS", "3840"))


def conv3x3_stride2_native_supported(x, weight):
    """Stride-2 layers the native fp32 kernels take: exact fp32, H % 4 == 0, W % 8 == 0 (and >= 16), channel counts that are multiples of
    16, at least 3 840 output pixels per launch -- all five DLA levels at 384x1280 x 8, levels 1-3 at one image.  Per level at bs 8
    (profiles/r06_stride2.txt): forward 124 / 110 / 104 / 107 / 125 us against the stock solver's 221 / 174 / 150 / 131 / 128, input
    gradient 227 / 121 / 124 / 132 / 148 against 243 / 186 / 180 / 157 / 164, weight gradient 231 / 178 / 126 / 121 / 127 against
    368 / 203 / 125 / 122 / 116: 2.1 ms for the five layers where the stock solver takes 2.7 and the space-to-depth form 3.75."""
    if not (_S2_NATIVE and x.is_cuda and x.dtype == torch.float32 and weight.dim() == 4 and tuple(weight.shape[2:]) == (3, 3)):
        return False
    # inside the split-bf16 scope as well (a scope permits reduced products, it does not oblige them): the exact kernels take 2.1 ms
    # for the five layers where the split space-to-depth form takes 3.75 -- bf16x3 step 34.4 -> 33.2 ms.  Not inside the bf16 scope:
    # the one-product space-to-depth form is the faster one there (everything but DCN 18.3 ms against 18.5-19.1 with these kernels)
    if _conv_prec(x) == PREC_BF16:
        return False
    B, C, H, W = x.shape
    K = weight.shape[0]
    return (H % 4 == 0 and W % 8 == 0 and W >= 16 and C % 16 == 0 and K % 16 == 0 and B * (H // 2) * (W // 2) >= _S2_NATIVE_MIN_PIXELS
            and B * max(C * H * W, K * (H // 2) * (W // 2)) < (1 << 29))


class _Conv3x3S2Native(torch.autograd.Function):
    """conv2d(x, weight, stride 2, padding 1) on csrc/conv_s2_f32.inc: forward, input gradient, weight gradient (224 / 177 / 125 / 119 us
    for DLA levels 1-4 at bs 8 against the stock solver's 368 / 203 / 125 / 122)."""

    @staticmethod
    def forward(ctx, x, weight):
        _lib.require_cuda(x, weight)
        x, weight = _f32c(x), _f32c(weight)
        B, C, H, W = x.shape
        K = weight.shape[0]
        y = torch.empty((B, K, H // 2, W // 2), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().dcd_conv3x3_s2_f32(_lib.stream_of(x), x.data_ptr(), weight.data_ptr(), y.data_ptr(), B, C, H, W, K),
                   "dcd_conv3x3_s2_f32")
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = _f32c(gy)
        L = _lib.lib()
        B, C, H, W = x.shape
        K = weight.shape[0]
        st = _lib.stream_of(x)
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            _lib.check(L.dcd_conv3x3_s2_f32_backward_data(st, gy.data_ptr(), weight.data_ptr(), gx.data_ptr(), B, C, H, W, K),
                       "dcd_conv3x3_s2_f32_backward_data")
        if ctx.needs_input_grad[1]:
            gw = torch.empty_like(weight)
            n = L.dcd_conv3x3_s2_f32_wrw_workspace_bytes(B, C, H, W, K)
            ws = torch.empty(max(n, 16), dtype=torch.uint8, device=x.device)
            _lib.check(L.dcd_conv3x3_s2_f32_wrw(st, x.data_ptr(), gy.data_ptr(), gw.data_ptr(), B, C, H, W, K, ws.data_ptr(), n),
                       "dcd_conv3x3_s2_f32_wrw")
        return gx, gw


def conv3x3_stride2_native(x, weight):
    return _Conv3x3S2Native.apply(x, weight)


def conv3x3_wrw_only_supported(x, weight):
    """Maps below the forward kernel's limit (only when DCD_CONV_MIN_MAP raises it) where the weight-gradient kernel still
    wins (256->256 @ 24x80: 0.14 vs 0.18 ms incl. the stock path's transposes, tools/time_conv.py)."""
    return (_WRW_ENABLED and x.is_cuda and x.dtype == torch.float32 and weight.dim() == 4 and weight.shape[2] == 3
            and weight.shape[3] == 3 and x.shape[3] % 4 == 0 and x.shape[2] % 2 == 0 and weight.shape[0] >= 64
            and weight.shape[1] >= 64 and 24 * 80 <= x.shape[2] * x.shape[3] < _CONV_MIN_MAP)


class _Conv3x3StockFwd(torch.autograd.Function):
    """Stock forward and input gradient, weight gradient on csrc/conv.hip."""

    @staticmethod
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        ctx.prec = _conv_prec(x)
        return torch.nn.functional.conv2d(x, weight, None, 1, 1)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = _f32c(gy)
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = torch.ops.aten.convolution_backward(gy, x, weight, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                     [True, False, False])[0]
        if ctx.needs_input_grad[1]:
            gw = _conv3x3_wrw_call(_f32c(x), gy, weight.shape, ctx.prec)
        return gx, gw


def conv3x3_stock_forward(x, weight):
    return _Conv3x3StockFwd.apply(x, weight)


# ----------------------------------------------------------------------------------------------
# Stem convolutions (3->16 7x7 and 16->16 3x3 at full resolution) on csrc/stem.hip
# ----------------------------------------------------------------------------------------------
def conv_stem_supported(x, weight, stride, padding, dilation, groups):
    k = weight.shape[2]
    return (x.is_cuda and x.dtype == torch.float32 and weight.dim() == 4 and weight.shape[0] == 16 and weight.shape[2] == weight.shape[3]
            and (weight.shape[1], k) in ((16, 3), (3, 7)) and tuple(stride) == (1, 1) and tuple(padding) == (k // 2, k // 2)
            and tuple(dilation) == (1, 1) and groups == 1 and x.shape[3] % 4 == 0 and x.shape[2] * x.shape[3] >= 1 << 16)


def _conv_stem_call(inp, weight, backward_data):
    L = _lib.lib()
    B, _, H, W = inp.shape
    Co, Ci, k = weight.shape[0], weight.shape[1], weight.shape[2]
    out = torch.empty((B, Ci if backward_data else Co, H, W), dtype=torch.float32, device=inp.device)
    n = L.dcd_conv_stem_workspace_bytes(Ci, Co, k)
    ws = torch.empty(n, dtype=torch.uint8, device=inp.device)
    st = L.dcd_conv_stem(_lib.stream_of(inp), inp.data_ptr(), weight.data_ptr(), out.data_ptr(), B, Ci, H, W, Co, k,
                         1 if backward_data else 0, ws.data_ptr(), n)
    _lib.check(st, "dcd_conv_stem")
    return out


class _ConvStem(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight):
        _lib.require_cuda(x, weight)
        x, weight = _f32c(x), _f32c(weight)
        ctx.save_for_backward(x, weight)
        return _conv_stem_call(x, weight, False)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = _f32c(gy)
        gx = gw = None
        if ctx.needs_input_grad[0]:
            gx = _conv_stem_call(gy, weight, True)
        if ctx.needs_input_grad[1]:
            L = _lib.lib()
            B, Ci, H, W = x.shape
            Co, k = weight.shape[0], weight.shape[2]
            gw = torch.empty_like(weight)
            n = L.dcd_conv_stem_wrw_workspace_bytes(Ci, Co, k)
            ws = torch.empty(n, dtype=torch.uint8, device=x.device)
            st = L.dcd_conv_stem_wrw(_lib.stream_of(x), x.data_ptr(), gy.data_ptr(), gw.data_ptr(), B, Ci, H, W, Co, k, ws.data_ptr(), n)
            _lib.check(st, "dcd_conv_stem_wrw")
        return gx, gw


def conv_stem(x, weight):
    return _ConvStem.apply(x, weight)


class _ContextNorm(torch.autograd.Function):
    """GMW's `gcn` (per-row zero mean, unit unbiased variance + 1e-3) as one launch forward and one backward."""

    @staticmethod
    def forward(ctx, x, eps):
        _lib.require_cuda(x)
        x = _f32c(x)
        K = x.shape[-1]
        rows = x.numel() // K
        y = torch.empty_like(x)
        inv = torch.empty(rows, dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().dcd_context_norm_forward(_lib.stream_of(x), x.data_ptr(), y.data_ptr(), inv.data_ptr(), rows, K, float(eps)),
                   "dcd_context_norm_forward")
        ctx.save_for_backward(y, inv)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        y, inv = ctx.saved_tensors
        gy = _f32c(gy)
        K = y.shape[-1]
        gx = torch.empty_like(y)
        _lib.check(_lib.lib().dcd_context_norm_backward(_lib.stream_of(y), gy.data_ptr(), y.data_ptr(), inv.data_ptr(), gx.data_ptr(),
                                                        y.numel() // K, K), "dcd_context_norm_backward")
        return gx, None


def context_norm(x, eps=1e-3):
    return _ContextNorm.apply(x, eps)


class _FanOut(torch.autograd.Function):
    """n aliases of one tensor whose gradients are summed by ONE kernel (autograd would run n-1 pairwise additions, each
    a read-read-write pass over the map)."""

    @staticmethod
    def forward(ctx, x, n):
        _lib.require_cuda(x)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *grads):
        import ctypes
        gs = [_f32c(g) for g in grads if g is not None]
        if not gs:
            return None, None
        if len(gs) == 1:
            return gs[0], None
        out = torch.empty_like(gs[0])
        L = _lib.lib()
        st = _lib.stream_of(out)
        for i in range(0, len(gs), 15):                       # 16 sources per launch; the running sum is one of them
            part = gs[i:i + 15] + ([out] if i else [])
            arr = (ctypes.c_void_p * len(part))(*[t.data_ptr() for t in part])
            _lib.check(L.dcd_sum_tensors(st, arr, len(part), out.data_ptr(), out.numel()), "dcd_sum_tensors")
        return out, None


def fan_out(x, n):
    """x -> n tensors aliasing x; use each for one consumer."""
    return _FanOut.apply(x, n)


_OFFSET_CONV_BWD = os.environ.get("DCD_OFFSET_CONV_BWD", "1") != "0"       # 0: stock input / weight gradients (A/B timing)
_OFFSET_CONV_FWD = os.environ.get("DCD_OFFSET_CONV_FWD", "1") != "0"       # 0: stock forward


_CHANNEL_SUM_WS = {}


def channel_sums(gy):
    """Per-channel sums of a (B, C, ...) fp32 tensor (a bias gradient) by the two-stage fp64 sums of csrc/norm.hip, one launch."""
    L = _lib.lib()
    gy = _f32c(gy)
    B, C = gy.shape[0], gy.shape[1]
    HW = gy.numel() // (B * C)
    sums = torch.empty(C, dtype=torch.float32, device=gy.device)
    st = _lib.stream_of(gy)
    # the kernel's arrival counters sit in the workspace and must enter zeroed (it leaves them zeroed): one zero-filled buffer
    # per (device, stream, size), never shared between streams
    key = (gy.device, int(st or 0), C)
    ws = _CHANNEL_SUM_WS.get(key)
    if ws is None:
        ws = _CHANNEL_SUM_WS[key] = torch.zeros(L.dcd_bn_workspace_bytes(C), dtype=torch.uint8, device=gy.device)
    status = L.dcd_channel_sums(st, gy.data_ptr(), B, C, HW, sums.data_ptr(), ws.data_ptr(), ws.numel())
    if status != 0:
        _CHANNEL_SUM_WS.pop(key, None)                 # a failed launch may have left arrival counters behind: start afresh
    _lib.check(status, "dcd_channel_sums")
    return sums


def conv3x3_bias_supported(x, weight, stride, padding, dilation):
    """Biased 3x3 / stride 1 / pad 1 layers that run on csrc/conv.hip end to end (see _ConvBias)."""
    return (_OFFSET_CONV_BWD and x.is_cuda and x.dtype == torch.float32 and weight.dim() == 4 and weight.shape[2] == 3
            and weight.shape[3] == 3 and (list(stride), list(padding), list(dilation)) == ([1, 1], [1, 1], [1, 1])
            and x.shape[3] % 4 == 0 and x.shape[2] % 2 == 0 and weight.shape[1] >= 64 and x.shape[2] * x.shape[3] >= _CONV_MIN_MAP)


class _ConvBias(torch.autograd.Function):
    """conv2d with a bias (DCN's `conv_offset_mask`, Cin -> 27).  The 3x3 / stride 1 / pad 1 layers with at least 64 inputs run
    on csrc/conv.hip: forward with the bias added in the output transform, input gradient and weight gradient, all on the
    kernels' one-output-block variants (a 64-wide output slice would be 58 % padding).  bs 8, ours vs stock (tools/time_conv.py
    with DCD_TIME_OFFSET_CONVS=1): 64 @ 96x320 forward 76 vs 101 us, input gradient 71-80 vs 100, weight gradient 80 vs 147 +
    the stock path's NHWC transposes of both operands; 128 @ 48x160: 37 / 40 / 48 vs 51 / 55 / 80.
    The bias gradient comes from the two-stage sums of csrc/norm.hip (ATen's generic reduction sums the (B,27,H,W) gradient at
    0.4 TB/s: 0.9 ms per step over the 16 layers).  Everything else: stock op."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, dilation):
        ctx.conf = (list(stride), list(padding), list(dilation))
        ctx.ours = conv3x3_bias_supported(x, weight, stride, padding, dilation)
        if ctx.ours:
            x, weight = _f32c(x), _f32c(weight)
        ctx.save_for_backward(x, weight)
        ctx.prec = _conv_prec(x) if ctx.ours else PREC_F32
        if ctx.ours and _OFFSET_CONV_FWD:
            return _conv3x3_call(x, weight, weight.shape[0], False, _f32c(bias), prec=ctx.prec)
        return torch.nn.functional.conv2d(x, weight, bias, stride, padding, dilation)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        stride, padding, dilation = ctx.conf
        gy = _f32c(gy)
        ours = ctx.ours
        ours_w = ours and _WRW_ENABLED
        gx = gw = None
        if ours and ctx.needs_input_grad[0]:
            gx = _conv3x3_call(gy, weight, weight.shape[1], True, prec=ctx.prec)
        if ours_w and ctx.needs_input_grad[1]:
            gw = _conv3x3_wrw_call(x, gy, weight.shape, ctx.prec)
        need_x, need_w = ctx.needs_input_grad[0] and gx is None, ctx.needs_input_grad[1] and gw is None
        if need_x or need_w:
            sx, sw, _ = torch.ops.aten.convolution_backward(gy, x, weight, None, stride, padding, dilation, False, [0, 0], 1,
                                                            [need_x, need_w, False])
            gx = sx if need_x else gx
            gw = sw if need_w else gw
        gb = channel_sums(gy) if ctx.needs_input_grad[2] else None
        return gx, gw, gb, None, None, None


def conv2d_bias(x, weight, bias, stride, padding, dilation):
    return _ConvBias.apply(x, weight, bias, stride, padding, dilation)


# ----------------------------------------------------------------------------------------------
# Depthwise transposed convolution of IDAUp (kernel 2f, stride f, padding f/2, groups = channels)
# ----------------------------------------------------------------------------------------------
class _UpsampleDW(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, f, skip=None):
        _lib.require_cuda(x, weight)
        x, weight = _f32c(x), _f32c(weight)
        B, C, H, W = x.shape
        y = torch.empty((B, C, H * f, W * f), dtype=torch.float32, device=x.device)
        L = _lib.lib()
        if skip is None:
            st = L.dcd_upsample_dw_forward(_lib.stream_of(x), x.data_ptr(), weight.data_ptr(), y.data_ptr(), B, C, H, W, f)
        else:
            skip = _f32c(skip)
            if tuple(skip.shape) != tuple(y.shape):
                raise RuntimeError("upsample_dw: skip must have the output's shape")
            st = L.dcd_upsample_dw_forward_add(_lib.stream_of(x), x.data_ptr(), weight.data_ptr(), skip.data_ptr(), y.data_ptr(),
                                               B, C, H, W, f)
        _lib.check(st, "dcd_upsample_dw_forward")
        ctx.save_for_backward(x, weight)
        ctx.f = f
        ctx.has_skip = skip is not None
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = _f32c(gy)
        B, C, H, W = x.shape
        gx, gw = torch.empty_like(x), torch.empty_like(weight)
        st = _lib.lib().dcd_upsample_dw_backward(_lib.stream_of(x), x.data_ptr(), weight.data_ptr(), gy.data_ptr(), gx.data_ptr(),
                                                 gw.data_ptr(), B, C, H, W, ctx.f)
        _lib.check(st, "dcd_upsample_dw_backward")
        return gx, gw, None, (gy if ctx.has_skip else None)       # the skip's gradient is the incoming one, untouched


class _MaxPool2x2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _lib.require_cuda(x)
        x = _f32c(x)
        B, C, H, W = x.shape
        y = torch.empty((B, C, H // 2, W // 2), dtype=torch.float32, device=x.device)
        st = _lib.lib().dcd_maxpool2x2_forward(_lib.stream_of(x), x.data_ptr(), y.data_ptr(), B * C, H, W)
        _lib.check(st, "dcd_maxpool2x2_forward")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        gy = _f32c(gy)
        gx = torch.empty_like(x)
        B, C, H, W = x.shape
        st = _lib.lib().dcd_maxpool2x2_backward(_lib.stream_of(x), x.data_ptr(), gy.data_ptr(), gx.data_ptr(), B * C, H, W)
        _lib.check(st, "dcd_maxpool2x2_backward")
        return gx


def maxpool2x2(x):
    """max_pool2d(x, 2, 2) for (B, C, H, W) with H even and W % 4 == 0 (csrc/upsample.hip); no index tensor is kept."""
    return _MaxPool2x2.apply(x)


def upsample_dw(x, weight, f, skip=None):
    """y = conv_transpose2d(x, weight, stride=f, padding=f//2, groups=C) for weight (C,1,2f,2f) (+ skip, in the same pass)."""
    return _UpsampleDW.apply(x, weight, f, skip)


# ----------------------------------------------------------------------------------------------
# Batch norm (+ReLU), training mode, evaluated at listed positions only
# ----------------------------------------------------------------------------------------------
class _BatchNormActAt(torch.autograd.Function):
    """y_at (B,N,C) = act(batch_norm(x))[b, :, pos[b,n]].  Statistics (and running buffers) exactly as the dense op; the dense
    normalised map is never written.  Backward: the gradient w.r.t. x is dense (every pixel feels the batch statistics) but
    needs one read of x and one write, instead of the seven tensor passes of the dense BN + ReLU backward."""

    @staticmethod
    def forward(ctx, x, pos, weight, bias, running_mean, running_var, num_batches_tracked, momentum, eps, relu, group):
        _lib.require_cuda(x, pos, weight, bias)
        L = _lib.lib()
        x = _f32c(x)
        pos = pos.contiguous().long()
        B, C = x.shape[0], x.shape[1]
        HW = x.numel() // (B * C)
        N = pos.shape[1]
        dev, st = x.device, _lib.stream_of(x)
        ws = _bn_ws(C, dev)
        count = float(B * HW)
        stats = None
        if group is not None:
            import torch.distributed as dist
            stats = torch.empty((C, 2), dtype=torch.float64, device=dev)
            _lib.check(L.dcd_bn_stats(st, x.data_ptr(), B, C, HW, stats.data_ptr(), ws.data_ptr(), ws.numel()), "dcd_bn_stats")
            dist.all_reduce(stats, group=group)
            count *= dist.get_world_size(group)
        x_at = torch.empty((B, N, C), dtype=torch.float32, device=dev)
        y_at = torch.empty((B, N, C), dtype=torch.float32, device=dev)
        save_mean = torch.empty(C, dtype=torch.float32, device=dev)
        save_invstd = torch.empty(C, dtype=torch.float32, device=dev)
        _lib.check(L.dcd_bn_at_forward(st, x.data_ptr(), pos.data_ptr(), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(stats), count,
                                       _lib.ptr(running_mean), _lib.ptr(running_var), _lib.ptr(num_batches_tracked),
                                       float(momentum), float(eps), int(bool(relu)), x_at.data_ptr(), y_at.data_ptr(),
                                       save_mean.data_ptr(), save_invstd.data_ptr(), B, C, HW, N, ws.data_ptr(), ws.numel()),
                   "dcd_bn_at_forward")
        ctx.save_for_backward(x, pos, x_at, y_at, weight, save_mean, save_invstd)
        ctx.count, ctx.group, ctx.relu = count, group, bool(relu)
        return y_at

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_at):
        x, pos, x_at, y_at, weight, save_mean, save_invstd = ctx.saved_tensors
        L = _lib.lib()
        g_at = _f32c(g_at)
        B, C, H, W = x.shape
        HW, N = H * W, pos.shape[1]
        dev, st = x.device, _lib.stream_of(x)
        sums = torch.empty((C, 2), dtype=torch.float64, device=dev)
        dzk = torch.empty((B, N, C), dtype=torch.float32, device=dev)
        _lib.check(L.dcd_bn_at_backward_sums(st, g_at.data_ptr(), x_at.data_ptr(), y_at.data_ptr(), _lib.ptr(weight),
                                             save_mean.data_ptr(), save_invstd.data_ptr(), int(ctx.relu), B * N, C, sums.data_ptr(),
                                             dzk.data_ptr()), "dcd_bn_at_backward_sums")
        gw = (sums[:, 1] * save_invstd.double()).float()       # local sums: DDP averages the parameter gradients
        gb = sums[:, 0].float()
        if ctx.group is not None:
            import torch.distributed as dist
            dist.all_reduce(sums, group=ctx.group)
        gx = torch.empty_like(x)
        _lib.check(L.dcd_bn_backward_apply(st, None, None, x.data_ptr(), _lib.ptr(weight), save_mean.data_ptr(),
                                           save_invstd.data_ptr(), sums.data_ptr(), ctx.count, gx.data_ptr(), None, None, None,
                                           B, C, HW), "dcd_bn_backward_apply")
        _lib.check(L.dcd_poi_scatter_add(st, dzk.data_ptr(), pos.data_ptr(), B, C, H, W, N, gx.data_ptr()), "dcd_poi_scatter_add")
        return (gx, None, gw if weight is not None else None, gb if weight is not None else None, None, None, None, None, None,
                None, None)


def batch_norm_act_at(x, pos, weight, bias, running_mean, running_var, num_batches_tracked, momentum, eps, relu, group=None):
    return _BatchNormActAt.apply(x, pos, weight, bias, running_mean, running_var, num_batches_tracked, momentum, eps, relu, group)
