"""PyTorch (CPU, autograd-capable) restatements of the routines dcd_amd.ops implements in HIP.

TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Same call signatures as dcd_amd.ops so that tests can run the host-side
model / loss logic on the CPU by patching these in (tests/conftest.py::cpu_backend); each body restates the
reference routine with stock torch ops, which also makes it a second checker for the HIP kernels' gradients.
"""
import torch
import torch.nn.functional as F


def pairs_kpts_depth(kps, kps_3d, rot_y, K, training=False, kpts_2d_mask=None, num_k=1500,
                     zmin=2.0, zmax=80.0, normalized=False, sub_b3=True):
    """Anno_Encoder.decode_pairs_kpts_depth (DGDE/model/anno_encoder.py:326-390) via triu indices."""
    n_kp = kps.shape[1]
    if normalized:
        v = kps[:, :, 1]
    else:
        v = (kps[:, :, 1] - K[:, None, 1, 2]) / K[:, None, 1, 1]
    rot = rot_y.reshape(-1, 1)
    C = kps_3d[:, :, 0] * torch.sin(rot) - kps_3d[:, :, 2] * torch.cos(rot)
    H1, H2 = kps_3d[:, :, 1], v * C
    iu = torch.triu_indices(n_kp, n_kp, offset=1)
    i, j = iu[0], iu[1]
    hmat = (H1[:, i] - H1[:, j]) + (H2[:, i] - H2[:, j])
    vmat = v[:, i] - v[:, j]
    z = (hmat.abs() / vmat.abs().clamp_min(1e-10)).clamp_min(zmin).clamp_max(zmax)
    mask = None
    if kpts_2d_mask is not None:
        m = kpts_2d_mask.to(z.dtype)
        mask = m[:, i] * m[:, j]
    if training:
        _, idx = torch.topk(vmat.abs(), num_k, dim=-1)
        z = z.gather(-1, idx)
        if mask is not None:
            mask = mask.gather(-1, idx)
    if sub_b3:
        z = z - K[:, 2, 3].unsqueeze(-1)
    return z, mask


def compute_z(kpts_2d, kpts_3d, pred_rot, num_k=1500):
    """GMW compute_z (GMW/main.py:373-416)."""
    dummy = torch.zeros(kpts_2d.shape[0], 3, 4, dtype=kpts_2d.dtype)
    z, _ = pairs_kpts_depth(kpts_2d, kpts_3d, pred_rot, dummy, training=False, zmin=0.1, normalized=True, sub_b3=False)
    v = kpts_2d[:, :, 1]
    iu = torch.triu_indices(v.shape[1], v.shape[1], offset=1)
    _, idx = torch.topk((v[:, iu[0]] - v[:, iu[1]]).abs(), num_k, dim=-1)
    return z, idx


def focal_loss(prediction, target, alpha=2, beta=4):
    """FocalLoss.forward (DGDE/model/layers/focal_loss.py:57-86)."""
    p = prediction.clamp(1e-10, 1 - 1e-10)
    pos = target.eq(1).float()
    neg = (target.lt(1) & target.ge(0)).float()
    negw = torch.pow(1 - target, beta)
    pl = torch.log(p) * torch.pow(1 - p, alpha) * pos
    nl = torch.log(1 - p) * torch.pow(p, alpha) * negw * neg
    return (-nl - pl).sum(), pos.sum()


def giou_loss(pred, target):
    """IOULoss('giou').forward (DGDE/model/layers/iou_loss.py:12-49)."""
    pl, pt, pr, pb = pred[:, 0], pred[:, 1], pred[:, 2], pred[:, 3]
    tl, tt, tr, tb = target[:, 0], target[:, 1], target[:, 2], target[:, 3]
    ta, pa = (tl + tr) * (tt + tb), (pl + pr) * (pt + pb)
    wi = torch.min(pl, tl) + torch.min(pr, tr)
    gwi = torch.max(pl, tl) + torch.max(pr, tr)
    hi = torch.min(pb, tb) + torch.min(pt, tt)
    ghi = torch.max(pb, tb) + torch.max(pt, tt)
    ac = gwi * ghi + 1e-7
    ai = wi * hi
    au = ta + pa - ai
    ious = (ai + 1.0) / (au + 1.0)
    return 1 - (ious - (ac - au) / ac), ious


def nms_hm(heat_map, kernel=3, reso=1):
    """nms_hm (DGDE/model/layers/utils.py:45-58)."""
    hmax = F.max_pool2d(heat_map, kernel_size=(3, 3), stride=1, padding=1)
    return heat_map * (hmax == heat_map).float()


def select_topk(heat_map, K=100, fuse_nms=False):
    """select_topk (DGDE/model/layers/utils.py:61-100) without its CUDA-only asserts."""
    if fuse_nms:
        heat_map = nms_hm(heat_map)
    batch, cls, height, width = heat_map.size()
    hm = heat_map.view(batch, cls, -1)
    scores_all, inds_all = torch.topk(hm, K)
    ys = (inds_all / width).int().float()
    xs = (inds_all % width).float()
    scores, inds = torch.topk(scores_all.view(batch, -1), K)
    clses = (inds / K).float()
    g = lambda t: t.view(batch, -1).gather(1, inds)
    return scores, g(inds_all), clses, g(ys), g(xs)


def select_point_of_interest(batch, index, feature_maps):
    """select_point_of_interest (DGDE/model/layers/utils.py:120-145)."""
    w = feature_maps.shape[3]
    if index.dim() == 3:
        index = index[:, :, 1] * w + index[:, :, 0]
    index = index.view(batch, -1)
    fm = feature_maps.permute(0, 2, 3, 1).contiguous()
    channel = fm.shape[-1]
    fm = fm.view(batch, -1, channel)
    return fm.gather(1, index.unsqueeze(-1).repeat(1, 1, channel).long())


def _poly_area(poly):
    x, y = poly[:, 0], poly[:, 1]
    return 0.5 * torch.abs(torch.sum(x * torch.roll(y, -1) - torch.roll(x, -1) * y))


def _clip_convex(subject, clip):
    """Sutherland-Hodgman: polygon `subject` (k,2) clipped by the convex polygon `clip` (4,2)."""
    out = subject
    orient = torch.sign(torch.sum(clip[:, 0] * torch.roll(clip[:, 1], -1) - torch.roll(clip[:, 0], -1) * clip[:, 1]))
    if orient == 0:
        return out.new_zeros((0, 2))
    for e in range(clip.shape[0]):
        if out.shape[0] == 0:
            break
        a, b = clip[e], clip[(e + 1) % clip.shape[0]]
        edge = b - a
        side = orient * (edge[0] * (out[:, 1] - a[1]) - edge[1] * (out[:, 0] - a[0]))
        pts = []
        n = out.shape[0]
        for k in range(n):
            j = (k + 1) % n
            if side[k] >= 0:
                pts.append(out[k])
            if (side[k] >= 0) != (side[j] >= 0):
                t = side[k] / (side[k] - side[j])
                pts.append(out[k] + t * (out[j] - out[k]))
        out = torch.stack(pts) if pts else out.new_zeros((0, 2))
    return out


def iou_3d(pred_corners, target_corners):
    """get_iou_3d (DGDE/model/layers/iou_loss.py:99-136) with the shapely rectangle intersection replaced by
    an explicit convex clip (the two are the same area for valid rectangles); float64 on the host."""
    A, B = pred_corners.detach().double().cpu(), target_corners.detach().double().cpu()
    N = A.shape[0]
    out = torch.zeros(N, dtype=torch.float64)
    min_h_a, max_h_a = -A[:, 0:4, 1].sum(1) / 4.0, -A[:, 4:8, 1].sum(1) / 4.0
    min_h_b, max_h_b = -B[:, 0:4, 1].sum(1) / 4.0, -B[:, 4:8, 1].sum(1) / 4.0
    h_overlap = torch.clamp(torch.min(max_h_a, max_h_b) - torch.max(min_h_a, min_h_b), min=0)
    for i in range(N):
        pa, pb = A[i, 0:4][:, [0, 2]], B[i, 0:4][:, [0, 2]]
        inter = _clip_convex(pa, pb)
        overlap = _poly_area(inter) if inter.shape[0] >= 3 else torch.zeros((), dtype=torch.float64)
        o3 = overlap * h_overlap[i]
        union = _poly_area(pa) * (max_h_a[i] - min_h_a[i]) + _poly_area(pb) * (max_h_b[i] - min_h_b[i]) - o3
        out[i] = o3 / union
    return out.float()
