"""CPU restatement (numpy) of the reference's training-target encoding -- TEST INFRASTRUCTURE, not product code.

Follows `KITTIDataset.__getitem__` (DGDE/data/datasets/kitti.py:283-606) for one image without augmentation, with the helpers it
calls: `pad_image` (:262-272), `get_edge_utils` (:165-223), `encode_alpha_multibin` (:225-244), `Object3d.generate_corners3d` /
`generate_extra_kpts_3d_loc` (kitti_utils.py:131-159), `Calibration.project_rect_to_image` (:361-369), `approx_proj_center`
(:1040-1078), `gaussian_radius`, `gaussian2D`, `draw_umich_gaussian`, `ellip_gaussian2D`, `draw_umich_gaussian_2D`
(DGDE/model/heatmap_coder.py:37-124).  Pinned by tests/golden/target_encoding.npz, which the reference's own code produced
(tests/golden/make_golden_targets.py).  Arithmetic types follow the reference: float64 throughout, except where it works on
float32 arrays (`obj.t`, `obj.box2d`)."""
import numpy as np

PI = np.pi
ALPHA_CENTERS = np.array([0, PI / 2, PI, -PI / 2])


def gaussian_radius(height, width, min_overlap=0.7):          # heatmap_coder.py:37-57
    a1, b1, c1 = 1, (height + width), width * height * (1 - min_overlap) / (1 + min_overlap)
    r1 = (b1 + np.sqrt(b1 ** 2 - 4 * a1 * c1)) / 2
    a2, b2, c2 = 4, 2 * (height + width), (1 - min_overlap) * width * height
    r2 = (b2 + np.sqrt(b2 ** 2 - 4 * a2 * c2)) / 2
    a3, b3, c3 = 4 * min_overlap, -2 * min_overlap * (height + width), (min_overlap - 1) * width * height
    r3 = (b3 + np.sqrt(b3 ** 2 - 4 * a3 * c3)) / 2
    return min(r1, r2, r3)


def _gauss(shape, sigma_x, sigma_y):                          # gaussian2D :59-68 / ellip_gaussian2D :126-134
    m, n = [(ss - 1.) / 2. for ss in shape]
    y, x = np.ogrid[-m:m + 1, -n:n + 1]
    h = np.exp(-(x * x) / (2 * sigma_x * sigma_x) - (y * y) / (2 * sigma_y * sigma_y))
    h[h < np.finfo(h.dtype).eps * h.max()] = 0
    return h


def draw_gaussian(heatmap, center, radius_x, radius_y):       # draw_umich_gaussian :84-107 (rx == ry) / _2D :109-124
    g = _gauss((2 * radius_y + 1, 2 * radius_x + 1), (2 * radius_x + 1) / 6, (2 * radius_y + 1) / 6)
    x, y = int(center[0]), int(center[1])
    height, width = heatmap.shape[0:2]
    left, right = min(x, radius_x), min(width - x, radius_x + 1)
    top, bottom = min(y, radius_y), min(height - y, radius_y + 1)
    mh = heatmap[y - top:y + bottom, x - left:x + right]
    mg = g[radius_y - top:radius_y + bottom, radius_x - left:radius_x + right]
    if min(mg.shape) > 0 and min(mh.shape) > 0:
        np.maximum(mh, mg, out=mh)
    return heatmap


def encode_alpha_multibin(alpha, num_bin=4, margin=1 / 6):    # kitti.py:225-244
    out = np.zeros(num_bin * 2)
    bin_size = 2 * PI / num_bin
    range_size = bin_size / 2 + bin_size * margin
    offsets = alpha - ALPHA_CENTERS[:num_bin]
    offsets[offsets > PI] = offsets[offsets > PI] - 2 * PI
    offsets[offsets < -PI] = offsets[offsets < -PI] + 2 * PI
    for i in range(num_bin):
        if abs(offsets[i]) < range_size:
            out[i] = 1
            out[i + num_bin] = offsets[i]
    return out


def edge_indices(image_size, pad_size, down_ratio=4):         # kitti.py:165-223 (torch.unique of a straight run = the run)
    img_w, img_h = image_size
    x_min, y_min = int(np.ceil(pad_size[0] / down_ratio)), int(np.ceil(pad_size[1] / down_ratio))
    x_max, y_max = (pad_size[0] + img_w - 1) // down_ratio, (pad_size[1] + img_h - 1) // down_ratio
    left = [(x_min, y) for y in range(y_min, y_max)]
    bottom = [(x, y_max) for x in range(x_min, x_max)]
    right = [(x_max, y) for y in range(y_max, y_min, -1)]
    top = [(x, y_min) for x in range(x_max, x_min - 1, -1)]
    return np.array(left + bottom + right + top, dtype=np.int64)


def approx_proj_center(proj_center, surface_centers, img_size):   # kitti_utils.py:1040-1078
    img_w, img_h = img_size
    inside = (surface_centers[:, 0] >= 0) & (surface_centers[:, 1] >= 0) & (surface_centers[:, 0] <= img_w - 1) & \
             (surface_centers[:, 1] <= img_h - 1)
    if inside.sum() == 0:
        return None
    tsc = surface_centers[inside.argmax()]
    a, b = np.polyfit([proj_center[0], tsc[0]], [proj_center[1], tsc[1]], 1)
    pts = []
    left_y = b
    if 0 <= left_y <= img_h - 1:
        pts.append(np.array([0, left_y]))
    right_y = (img_w - 1) * a + b
    if 0 <= right_y <= img_h - 1:
        pts.append(np.array([img_w - 1, right_y]))
    top_x = -b / a
    if 0 <= top_x <= img_w - 1:
        pts.append(np.array([top_x, 0]))
    bottom_x = (img_h - 1 - b) / a
    if 0 <= bottom_x <= img_w - 1:
        pts.append(np.array([bottom_x, img_h - 1]))
    pts = np.stack(pts)
    return pts[np.argmin(np.linalg.norm(pts - proj_center.reshape(1, 2), axis=1))]


def project(P, pts):                                           # kitti_utils.py:361-369
    hom = np.hstack((pts, np.ones((pts.shape[0], 1)))) @ P.T
    return hom[:, :2] / hom[:, 2:3], hom[:, 2]


def encode_image(image_size, P, trunc_occ, box2d, hwl, t, ry, alpha, find_pcl, kpts3d, input_size=(1280, 384), down_ratio=4,
                 max_objs=40, n_extra=63, filter_params=(0.9, 20), edge_heatmap_ratio=0.5, num_bin=4):
    """One image: raw label values (as `Object3d` holds them) -> dict of the ParamsList fields (kitti.py:354-606)."""
    img_w, img_h = int(image_size[0]), int(image_size[1])
    pad = np.array([(input_size[0] - img_w) // 2, (input_size[1] - img_h) // 2], dtype=np.int64)          # pad_image :262-272
    fw, fh = input_size[0] // down_ratio, input_size[1] // down_ratio
    x_min, y_min = int(np.ceil(pad[0] / down_ratio)), int(np.ceil(pad[1] / down_ratio))
    x_max, y_max = (pad[0] + img_w - 1) // down_ratio, (pad[1] + img_h - 1) // down_ratio
    K = n_extra + 10
    f = np.float32
    o = dict(hm=np.zeros((1, fh, fw), f), cls_ids=np.zeros(max_objs, np.int32), target_centers=np.zeros((max_objs, 2), np.int32),
             gt_bboxes=np.zeros((max_objs, 4), f), bboxes=np.zeros((max_objs, 4), f), extra_kpts_3d=np.zeros((max_objs, K, 3), f),
             extra_kpts_2d=np.zeros((max_objs, K, 3), f), Calib_P=np.zeros((max_objs, 3, 4), f), find_pcl=np.zeros(max_objs, bool),
             keypoints=np.zeros((max_objs, 10, 3), f), keypoints_depth_mask=np.zeros((max_objs, 3), f),
             extra_kpts_depth_mask=np.zeros((max_objs, K), f), dimensions=np.zeros((max_objs, 3), f),
             locations=np.zeros((max_objs, 3), f), rotys=np.zeros(max_objs, f), alphas=np.zeros(max_objs, f),
             offset_3D=np.zeros((max_objs, 2), f), occlusions=np.zeros(max_objs), truncations=np.zeros(max_objs),
             ori_mask=np.ones(max_objs, bool), orientations=np.zeros((max_objs, num_bin * 2), f), reg_mask=np.zeros(max_objs, np.uint8),
             trunc_mask=np.zeros(max_objs, np.uint8), reg_weight=np.zeros(max_objs, f))
    for i in range(len(ry)):
        h, w, l = (float(v) for v in hwl[i])
        locs = t[i].astype(np.float32).copy()
        locs[1] = locs[1] - h / 2
        if locs[-1] <= 0:
            continue
        R = np.array([[np.cos(ry[i]), 0, np.sin(ry[i])], [0, 1, 0], [-np.sin(ry[i]), 0, np.cos(ry[i])]])
        c_obj = np.vstack([[l / 2, l / 2, -l / 2, -l / 2, l / 2, l / 2, -l / 2, -l / 2], [0, 0, 0, 0, -h, -h, -h, -h],
                           [w / 2, -w / 2, -w / 2, w / 2, w / 2, -w / 2, -w / 2, w / 2]])
        raw_kpts_3d = np.vstack((c_obj.T.copy(), np.array([[0., 0., 0.], [0., -h, 0.]])))
        corners_3d = np.dot(R, c_obj).T + t[i]
        corners_2d, _ = project(P, corners_3d)
        pb = np.array([corners_2d[:, 0].min(), corners_2d[:, 1].min(), corners_2d[:, 0].max(), corners_2d[:, 1].max()])
        if pb[0] >= 0 and pb[1] >= 0 and pb[2] <= img_w - 1 and pb[3] <= img_h - 1:
            b2 = pb.copy()
        else:
            b2 = box2d[i].astype(np.float32).copy()
        if trunc_occ[i, 0] >= filter_params[0] and (b2[2:] - b2[:2]).min() <= filter_params[1]:
            continue
        proj_center, _ = project(P, locs.reshape(-1, 3))
        proj_center = proj_center[0]
        inside = (0 <= proj_center[0] <= img_w - 1) & (0 <= proj_center[1] <= img_h - 1)
        approx = False
        if not inside:
            approx = True
            center_2d = (b2[:2] + b2[2:]) / 2
            target_proj_center = approx_proj_center(proj_center, center_2d.reshape(1, 2), (img_w, img_h))
        else:
            target_proj_center = proj_center.copy()
        kp3 = np.concatenate((corners_3d, np.stack((corners_3d[:4].mean(axis=0), corners_3d[4:].mean(axis=0)), axis=0)), axis=0)
        kp2, _ = project(P, kp3)
        e3 = kpts3d[i]
        e3_cam = np.dot(R, e3.T).T + t[i]
        e2, _ = project(P, e3_cam)
        kv = (kp2[:, 0] >= 0) & (kp2[:, 0] <= img_w - 1) & (kp2[:, 1] >= 0) & (kp2[:, 1] <= img_h - 1) & (kp3[:, -1] > 0)
        ev = (e2[:, 0] >= 0) & (e2[:, 0] <= img_w - 1) & (e2[:, 1] >= 0) & (e2[:, 1] <= img_h - 1) & (e3_cam[:, -1] > 0)
        kv = np.append(np.tile(kv[:4] | kv[4:8], 2), np.tile(kv[8] | kv[9], 2))                       # KEYPOINT_VISIBLE_MODIFY :481-486
        kdv = np.stack((kv[[8, 9]].all(), kv[[0, 2, 4, 6]].all(), kv[[1, 3, 5, 7]].all())).astype(np.float32)
        kv = kv.astype(np.float32)
        kp2 = (kp2 + pad.reshape(1, 2)) / down_ratio
        e2 = (e2[:, :2] + pad.reshape(1, 2).repeat(e2.shape[0], axis=0)) / down_ratio
        target_proj_center = (target_proj_center + pad) / down_ratio
        proj_center = (proj_center + pad) / down_ratio
        b2[0::2] += pad[0]
        b2[1::2] += pad[1]
        b2 /= down_ratio
        bbox_dim = b2[2:] - b2[:2]
        tc = target_proj_center.round().astype(int)
        tc[0] = np.clip(tc[0], x_min, x_max)
        tc[1] = np.clip(tc[1], y_min, y_max)
        pred_2d = tc[0] >= b2[0] and tc[1] >= b2[1] and tc[0] <= b2[2] and tc[1] <= b2[3]
        if (bbox_dim > 0).all() and (0 <= tc[0] <= fw - 1) and (0 <= tc[1] <= fh - 1):
            if approx:
                bw = min(tc[0] - b2[0], b2[2] - tc[0])
                bh = min(tc[1] - b2[1], b2[3] - tc[1])
                rx, ry_ = max(0, int(bw * edge_heatmap_ratio)), max(0, int(bh * edge_heatmap_ratio))
                assert min(rx, ry_) == 0
                draw_gaussian(o['hm'][0], tc, rx, ry_)
            else:
                r = max(0, int(gaussian_radius(bbox_dim[1], bbox_dim[0])))
                draw_gaussian(o['hm'][0], tc, r, r)
            o['cls_ids'][i] = 0
            o['target_centers'][i] = tc
            o['offset_3D'][i] = proj_center - tc
            o['gt_bboxes'][i] = box2d[i]
            if pred_2d:
                o['bboxes'][i] = b2
            o['keypoints'][i] = np.concatenate((kp2 - tc.reshape(1, -1), kv[:, np.newaxis]), axis=1)
            o['extra_kpts_2d'][i] = np.vstack((np.concatenate((e2 - tc.reshape(1, -1), ev[:, np.newaxis]), axis=1), o['keypoints'][i]))
            o['extra_kpts_3d'][i] = np.vstack((e3, raw_kpts_3d))
            o['Calib_P'][i] = P
            o['find_pcl'][i] = find_pcl[i]
            o['keypoints_depth_mask'][i] = kdv
            o['extra_kpts_depth_mask'][i] = np.concatenate((ev, kv))
            o['dimensions'][i] = np.array([l, h, w])
            o['locations'][i] = locs
            o['rotys'][i] = ry[i]
            o['alphas'][i] = alpha[i]
            o['orientations'][i] = encode_alpha_multibin(alpha[i], num_bin=num_bin)
            o['reg_mask'][i] = 1
            o['reg_weight'][i] = 1
            o['trunc_mask'][i] = int(approx)
            o['occlusions'][i] = trunc_occ[i, 1]
            o['truncations'][i] = trunc_occ[i, 0]
    edges = edge_indices((img_w, img_h), pad, down_ratio)
    edge_pad = np.zeros(((fw + fh) * 2, 2), np.int64)
    edge_pad[:len(edges)] = edges
    o.update(pad_size=pad, edge_indices=edge_pad, edge_len=np.array(len(edges) - 1), final_output_w=np.array(fw), final_output_h=np.array(fh))
    o['2d_bboxes'] = o.pop('bboxes')
    return o
