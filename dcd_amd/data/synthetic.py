"""Seeded synthetic KITTI-shaped batches for the DGDE hot path (no dataset exists on either box).

Produces exactly the `ParamsList` field schema the reference's `KITTIDataset.__getitem__` emits
(DGDE/data/datasets/kitti.py:572-606) from random but geometrically consistent objects:
every 2-D quantity is the projection of the sampled 3-D boxes / keypoints with KITTI's P2, so the
edge-constraint solver recovers the true depth and the losses behave as on real labels.
"""
import numpy as np
import torch

from dcd_amd.data.calibration import Calibration, KITTI_P2
from dcd_amd.structures.params_3d import ParamsList

PI = np.pi
ALPHA_CENTERS = np.array([0, PI / 2, PI, -PI / 2])


def edge_indices_for(image_size, pad_size, down_ratio=4):
    """Clockwise walk over the border cells of the un-padded image area on the stride-`down_ratio` map:
    left column, bottom row, right column, top row (what kitti.py:170-223 builds); returns (n,2) int64 (x, y)."""
    img_w, img_h = image_size
    x_min, y_min = int(np.ceil(pad_size[0] / down_ratio)), int(np.ceil(pad_size[1] / down_ratio))
    x_max, y_max = (pad_size[0] + img_w - 1) // down_ratio, (pad_size[1] + img_h - 1) // down_ratio
    left = [(x_min, y) for y in range(y_min, y_max)]
    bottom = [(x, y_max) for x in range(x_min, x_max)]
    right = [(x_max, y) for y in range(y_max, y_min, -1)]
    top = [(x, y_min) for x in range(x_max, x_min - 1, -1)]
    return np.array(left + bottom + right + top, dtype=np.int64)


def encode_alpha_multibin(alpha, num_bin=4, margin=1 / 6):
    """[bin flags | bin offsets] (kitti.py:225-244)."""
    out = np.zeros(num_bin * 2)
    bin_size = 2 * PI / num_bin
    range_size = bin_size / 2 + bin_size * margin
    offsets = alpha - ALPHA_CENTERS[:num_bin]
    offsets[offsets > PI] -= 2 * PI
    offsets[offsets < -PI] += 2 * PI
    for i in range(num_bin):
        if abs(offsets[i]) < range_size:
            out[i] = 1
            out[i + num_bin] = offsets[i]
    return out


def gaussian_peak(hm, cx, cy, radius):
    """CenterNet-style Gaussian splat with peak exactly 1 at the (integer) centre."""
    H, W = hm.shape
    sigma = (2 * radius + 1) / 6.0
    x0, x1 = max(0, cx - radius), min(W, cx + radius + 1)
    y0, y1 = max(0, cy - radius), min(H, cy + radius + 1)
    ys, xs = np.mgrid[y0:y1, x0:x1]
    g = np.exp(-((xs - cx) ** 2 + (ys - cy) ** 2) / (2 * sigma * sigma))
    hm[y0:y1, x0:x1] = np.maximum(hm[y0:y1, x0:x1], g)
    hm[cy, cx] = 1.0


def scaled_P2(scale):
    """KITTI P2 for an image down-scaled by `scale` (reduced-resolution test inputs)."""
    P = KITTI_P2.copy()
    P[:2] *= scale
    return P


def make_target(seed, n_objects=6, input_size=(1280, 384), image_size=(1242, 375), down_ratio=4, max_objs=40,
                n_extra=63, P=None, with_ori_img=False, img_idx=None):
    """One image worth of labels as a ParamsList (train mode).  `P` defaults to KITTI's P2 scaled to the input width."""
    rng = np.random.RandomState(seed)
    in_w, in_h = input_size
    if P is None:
        P = KITTI_P2 if in_w == 1280 else scaled_P2(in_w / 1280.0)
    if in_w != 1280:
        image_size = (int(image_size[0] * in_w / 1280.0), int(image_size[1] * in_w / 1280.0))
    fw, fh = in_w // down_ratio, in_h // down_ratio
    # centred padding when the image is smaller than the input; clamp otherwise (reduced-resolution tests)
    image_size = (min(image_size[0], in_w), min(image_size[1], in_h))
    pad = np.array([(in_w - image_size[0]) // 2, (in_h - image_size[1]) // 2], dtype=np.int64)
    calib = Calibration(P)
    K = n_extra + 10

    f = np.float32
    t = dict(
        hm=np.zeros((1, fh, fw), f), cls_ids=np.zeros(max_objs, np.int32), target_centers=np.zeros((max_objs, 2), np.int32),
        offset_3D=np.zeros((max_objs, 2), f), bboxes=np.zeros((max_objs, 4), f), keypoints=np.zeros((max_objs, 10, 3), f),
        keypoints_depth_mask=np.zeros((max_objs, 3), f), extra_kpts_2d=np.zeros((max_objs, K, 3), f),
        extra_kpts_3d=np.zeros((max_objs, K, 3), f), extra_kpts_depth_mask=np.zeros((max_objs, K), f),
        Calib_P=np.zeros((max_objs, 3, 4), f), find_pcl=np.zeros(max_objs, bool), dimensions=np.zeros((max_objs, 3), f),
        locations=np.zeros((max_objs, 3), f), rotys=np.zeros(max_objs, f), alphas=np.zeros(max_objs, f),
        orientations=np.zeros((max_objs, 8), f), reg_mask=np.zeros(max_objs, np.uint8), trunc_mask=np.zeros(max_objs, np.uint8),
        reg_weight=np.zeros(max_objs, f), ori_mask=np.zeros(max_objs, bool))

    Pm = np.asarray(P, dtype=np.float64)

    def project(pts):  # (n,3) camera frame -> (n,2) feature-map coordinates of the padded input
        hom = np.concatenate([pts, np.ones((pts.shape[0], 1))], 1) @ Pm.T
        uv = hom[:, :2] / hom[:, 2:3]
        return (uv + pad[None]) / down_ratio

    n = 0
    attempts = 0
    while n < n_objects and attempts < 200:
        attempts += 1
        z = rng.uniform(8, 50)
        x = rng.uniform(-0.25, 0.25) * z
        l, h, w = rng.normal(3.9, 0.3), rng.normal(1.5, 0.1), rng.normal(1.6, 0.1)
        y = 1.65 - h / 2                                   # centre at mid height, bottom on the ground plane
        roty = rng.uniform(-PI, PI)
        loc = np.array([x, y, z])
        R = np.array([[np.cos(roty), 0, np.sin(roty)], [0, 1, 0], [-np.sin(roty), 0, np.cos(roty)]])
        # 8 corners in the anno-encoder order (dcd_amd/model/anno_encoder.py encode_box3d) + bottom / top centres
        sx = np.array([-1, -1, 1, 1, -1, -1, 1, 1]) * l / 2
        sy = np.array([1, 1, 1, 1, -1, -1, -1, -1]) * h / 2
        sz = np.array([-1, 1, 1, -1, -1, 1, 1, -1]) * w / 2
        box_obj = np.stack([sx, sy, sz], 1)
        ten_obj = np.concatenate([box_obj, [[0, h / 2, 0], [0, -h / 2, 0]]], 0)
        extra_obj = rng.uniform(-0.5, 0.5, (n_extra, 3)) * np.array([l, h, w])
        kp_obj = np.concatenate([extra_obj, ten_obj], 0)                       # last 10 = box points (kitti_utils.py:147)
        centre_f = project(loc[None])[0]
        cx, cy = int(centre_f[0]), int(centre_f[1])
        if not (0 <= cx < fw and 0 <= cy < fh):
            continue
        kp_f = project(kp_obj @ R.T + loc)
        ten_f = kp_f[n_extra:]
        x1, y1 = np.clip(ten_f[:8, 0].min(), 0, fw - 1), np.clip(ten_f[:8, 1].min(), 0, fh - 1)
        x2, y2 = np.clip(ten_f[:8, 0].max(), 0, fw - 1), np.clip(ten_f[:8, 1].max(), 0, fh - 1)
        alpha = roty - np.arctan2(x, z)
        alpha = (alpha + PI) % (2 * PI) - PI

        t['cls_ids'][n] = 0
        t['target_centers'][n] = (cx, cy)
        t['offset_3D'][n] = centre_f - np.array([cx, cy])
        t['bboxes'][n] = (x1, y1, x2, y2)
        t['keypoints'][n, :, :2] = ten_f - centre_f
        t['keypoints'][n, :, 2] = 1
        t['keypoints_depth_mask'][n] = 1
        t['extra_kpts_2d'][n, :, :2] = kp_f - centre_f
        inside = (kp_f[:, 0] >= 0) & (kp_f[:, 0] < fw) & (kp_f[:, 1] >= 0) & (kp_f[:, 1] < fh)
        t['extra_kpts_2d'][n, :, 2] = inside
        t['extra_kpts_3d'][n] = kp_obj
        t['extra_kpts_depth_mask'][n] = 1
        t['Calib_P'][n] = Pm
        t['find_pcl'][n] = True
        t['dimensions'][n] = (l, h, w)
        t['locations'][n] = loc
        t['rotys'][n] = roty
        t['alphas'][n] = alpha
        t['orientations'][n] = encode_alpha_multibin(alpha, 4)
        t['reg_mask'][n] = 1
        t['trunc_mask'][n] = 0
        t['reg_weight'][n] = 1
        t['ori_mask'][n] = True
        radius = max(1, int(0.1 * min(x2 - x1, y2 - y1)))
        gaussian_peak(t['hm'][0], cx, cy, radius)
        n += 1

    if n < n_objects:
        raise RuntimeError("could only place %d of %d synthetic objects inside the image" % (n, n_objects))
    edges = edge_indices_for(image_size, pad, down_ratio)
    max_edge = (fw + fh) * 2
    edge_pad = np.zeros((max_edge, 2), np.int64)
    edge_pad[:len(edges)] = edges

    target = ParamsList(image_size=image_size, is_train=True)
    for key, name in (('hm', 'hm'), ('cls_ids', 'cls_ids'), ('target_centers', 'target_centers'), ('offset_3D', 'offset_3D'),
                      ('bboxes', '2d_bboxes'), ('keypoints', 'keypoints'), ('keypoints_depth_mask', 'keypoints_depth_mask'),
                      ('extra_kpts_2d', 'extra_kpts_2d'), ('extra_kpts_3d', 'extra_kpts_3d'),
                      ('extra_kpts_depth_mask', 'extra_kpts_depth_mask'), ('Calib_P', 'Calib_P'), ('find_pcl', 'find_pcl'),
                      ('dimensions', 'dimensions'), ('locations', 'locations'), ('rotys', 'rotys'), ('alphas', 'alphas'),
                      ('orientations', 'orientations'), ('reg_mask', 'reg_mask'), ('trunc_mask', 'trunc_mask'),
                      ('reg_weight', 'reg_weight'), ('ori_mask', 'ori_mask')):
        target.add_field(name, t[key])
    target.add_field("pad_size", pad)
    target.add_field("calib", calib)
    target.add_field("edge_indices", edge_pad)
    target.add_field("edge_len", len(edges) - 1)
    target.add_field("final_output_w", fw)
    target.add_field("final_output_h", fh)
    target.add_field("img_idx", img_idx if img_idx is not None else "%06d" % seed)
    if with_ori_img:
        target.add_field("ori_img", np.zeros((in_h, in_w, 3), np.uint8))
    return target


def make_batch(batch_size, seed=0, n_objects=6, input_size=(1280, 384), device=None, **kw):
    """(images (B,3,H,W) ~ N(0,1), [ParamsList] * B); image i uses seed 1000+seed*batch+i for its labels."""
    g = torch.Generator().manual_seed(seed)
    images = torch.randn(batch_size, 3, input_size[1], input_size[0], generator=g)
    targets = [make_target(1000 + seed * batch_size + i, n_objects=n_objects, input_size=input_size, **kw)
               for i in range(batch_size)]
    if device is not None:
        images = images.to(device)
        targets = [t.to(device) for t in targets]
    # collate: one batched tensor per field, the per-image fields are its slices (the loader's job, once per batch; the device
    # target encoder hands its outputs over the same way) -- stacking them again in the loss is then free
    from dcd_amd.structures.params_3d import collate_fields
    return images, collate_fields(targets)
