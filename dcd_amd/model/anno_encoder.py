"""Geometry decode / encode helpers (API of DGDE/model/anno_encoder.py:13-393).

Same method names, arguments and results as the reference's `Anno_Encoder`, restructured for the GPU:
  * per-image Python loops over `torch.unique(batch_idxs).tolist()` (anno_encoder.py:150-159, :206-218: a
    device->host sync each) become one vectorised expression indexed by `batch_idxs` into a small per-image
    calibration table;
  * `get_up` + the (N,73,73) pair matrices of `decode_pairs_kpts_depth` (:313-390: ~8k micro-launches per call)
    become one HIP kernel launch (dcd_amd.ops.pairs_kpts_depth).
"""
import numpy as np
import torch
import torch.nn.functional as F

from dcd_amd import ops

PI = np.pi


# attribute <- cfg path (the names anno_encoder.py:14-50 reads); tensors are listed separately below
_CFG_ATTRS = {
    "min_radius": "DATASETS.MIN_RADIUS", "max_radius": "DATASETS.MAX_RADIUS", "center_ratio": "DATASETS.CENTER_RADIUS_RATIO",
    "target_center_mode": "INPUT.HEATMAP_CENTER", "orien_bin_size": "INPUT.ORIENTATION_BIN_SIZE",
    "center_mode": "MODEL.HEAD.CENTER_MODE", "depth_mode": "MODEL.HEAD.DEPTH_MODE", "depth_range": "MODEL.HEAD.DEPTH_RANGE",
    "scale_depth_by_focal_lengths_factor": "MODEL.HEAD.SCALE_DEPTH_BY_FOCAL_LENGTHS_FACTOR",
    "dim_modes": "MODEL.HEAD.DIMENSION_REG", "down_ratio": "MODEL.BACKBONE.DOWN_RATIO", "fp16": "MODEL.FP16",
}
_CFG_TENSORS = {"depth_ref": "MODEL.HEAD.DEPTH_REFERENCE", "dim_mean": "MODEL.HEAD.DIMENSION_MEAN",
                "dim_std": "MODEL.HEAD.DIMENSION_STD"}


def _cfg_get(cfg, path):
    for part in path.split("."):
        cfg = getattr(cfg, part)
    return cfg


class Anno_Encoder():
    INF = 100000000
    EPS = 1e-3

    def __init__(self, cfg):
        self.device = device = cfg.MODEL.DEVICE
        for attr, path in _CFG_ATTRS.items():
            setattr(self, attr, _cfg_get(cfg, path))
        for attr, path in _CFG_TENSORS.items():
            setattr(self, attr, torch.as_tensor(_cfg_get(cfg, path)).to(device=device))
        self.num_cls = len(cfg.DATASETS.DETECT_CLASSES)
        self.multibin = cfg.INPUT.ORIENTATION == 'multi-bin'
        self.offset_mean, self.offset_std = cfg.MODEL.HEAD.REGRESSION_OFFSET_STAT[:2]
        self.alpha_centers = torch.tensor([0, PI / 2, PI, - PI / 2]).to(device=device)
        # box corner signs: x uses l/2, y uses h/2, z uses w/2 (the gather table of anno_encoder.py:119-123)
        self._corner_sign = torch.tensor([[-1, -1, 1, 1, -1, -1, 1, 1],
                                          [1, 1, 1, 1, -1, -1, -1, -1],
                                          [-1, 1, 1, -1, -1, 1, 1, -1]], dtype=torch.float32)
        self._calib_cache = (None, None)
        self._kp_idx = (None, None, None)

    # ------------------------------------------------------------------------------------------
    def _calib_table(self, calibs, device):
        """(len(calibs), 6) float32 rows [c_u, c_v, f_u, f_v, b_x, b_y]; cached for the last VALUES seen.  The key is the
        intrinsics themselves, not object identity: a data loader builds new Calibration objects every batch and CPython
        reuses freed addresses, so id()-keys would hand a stale table to the next batch."""
        if torch.is_tensor(calibs):                              # already a table (graph-captured loss)
            return calibs
        if len(calibs) and torch.is_tensor(calibs[0]):           # per-image (6,) rows (GraphedTrainStep's static targets)
            return torch.stack(list(calibs)).to(device)
        rows = tuple((float(c.c_u), float(c.c_v), float(c.f_u), float(c.f_v), float(c.b_x), float(c.b_y)) for c in calibs)
        key = (rows, str(device))
        if self._calib_cache[0] != key:
            self._calib_cache = (key, torch.tensor(rows, dtype=torch.float32, device=device))
        return self._calib_cache[1]

    @staticmethod
    def rad_to_matrix(rotys, N):
        """(N,3,3) rotations about the camera y axis (anno_encoder.py:53-71)."""
        cos, sin = rotys.cos(), rotys.sin()
        zero, one = torch.zeros_like(cos), torch.ones_like(cos)
        return torch.stack([cos, zero, sin, zero, one, zero, -sin, zero, cos], dim=1).view(N, 3, 3)

    def decode_box2d_fcos(self, centers, pred_offset, pad_size=None, out_size=None):
        box2d_center = centers.view(-1, 2)
        box2d = torch.cat((box2d_center - pred_offset[:, :2], box2d_center + pred_offset[:, 2:]), dim=1)
        if pad_size is not None:  # inference: back to the un-padded image, clamped to its bounds (:82-89)
            out_size = out_size[0]
            box2d = box2d * self.down_ratio - pad_size.repeat(1, 2)
            lim = torch.stack((out_size[0], out_size[1], out_size[0], out_size[1])).to(box2d) - 1
            box2d = torch.min(box2d.clamp(min=0), lim.view(1, 4))
        return box2d

    def encode_box3d(self, rotys, dims, locs):
        """8 corners (N,8,3) from yaw, (l,h,w) and centre (anno_encoder.py:93-128)."""
        if rotys.dim() == 2:
            rotys = rotys.flatten()
        dims = dims.view(-1, 3)
        locs = locs.view(-1, 3)
        N = rotys.shape[0]
        if self._corner_sign.device != dims.device or self._corner_sign.dtype != dims.dtype:   # one host->device copy, ever
            self._corner_sign = self._corner_sign.to(device=dims.device, dtype=dims.dtype)
        sign = self._corner_sign
        obj = (dims * 0.5).unsqueeze(-1) * sign.unsqueeze(0)              # (N,3,8): x<-l, y<-h, z<-w
        ry = self.rad_to_matrix(rotys, N).to(obj.dtype)
        box_3d = torch.matmul(ry, obj) + locs.unsqueeze(-1)
        return box_3d.permute(0, 2, 1)

    def decode_depth(self, depths_offset, calib_P=None):
        if self.depth_mode == 'exp':
            depth = depths_offset.exp()
        elif self.depth_mode == 'linear':
            depth = depths_offset * self.depth_ref[1] + self.depth_ref[0]
        elif self.depth_mode == 'inv_sigmoid':
            depth = 1.0 / torch.sigmoid(depths_offset) - 1.0
        else:
            raise ValueError
        if self.depth_range is not None:
            depth = torch.clamp(depth, min=self.depth_range[0], max=self.depth_range[1])
        return depth

    def decode_location_flatten(self, points, offsets, depths, calibs, pad_size, batch_idxs):
        """Back-project (centre + offset) at the given depth with each object's own image calibration
        (anno_encoder.py:147-161; project_image_to_rect, data/datasets/kitti_utils.py:399-418)."""
        batch_idxs = batch_idxs.long()
        tab = self._calib_table(calibs, points.device)[batch_idxs]          # (n, 6): c_u, c_v, f_u, f_v, b_x, b_y
        pts = (points + offsets) * self.down_ratio - pad_size[batch_idxs]
        depths = depths.float()
        # x and y together: ((u - c) * z) / f + b
        xy = ((pts - tab[:, 0:2]) * depths.unsqueeze(1)) / tab[:, 2:4] + tab[:, 4:6]
        return torch.cat((xy, depths.unsqueeze(1)), dim=1)

    def decode_depth_from_keypoints_batch(self, pred_keypoints, pred_dimensions, calibs, batch_idxs=None, f_u=None):
        """Depth from the projected height of the centre line and the two corner groups (anno_encoder.py:193-224).
        Quirk kept: the reference indexes `calibs` by the RANK of the image among the images that own objects
        (`calibs[idx]`, :206-207), which differs from the image index only if some image has no object."""
        pred_height_3D = pred_dimensions[:, 1]
        n_img = len(calibs)
        tab = self._calib_table(calibs, pred_keypoints.device)
        if f_u is not None:                                      # the caller's per-row focal lengths (PostProcessor.forward_batch)
            pass
        elif n_img == 1 or batch_idxs is None:
            f_u = tab[0, 2]
        else:
            bi = batch_idxs.long()
            present = torch.zeros(n_img, dtype=torch.long, device=bi.device).index_fill_(0, bi, 1)
            rank = torch.cumsum(present, 0) - 1
            f_u = tab[rank[bi], 2]
        # heights of the centre line (keypoints 8-9) and of the corner pairs 0-4, 2-6 | 1-5, 3-7 in one (n, 5) tensor; the
        # index tensors live on the device (a Python list index would be a host->device copy on every call)
        nk = pred_keypoints.shape[1]
        key = (nk, str(pred_keypoints.device))
        if self._kp_idx[0] != key:
            self._kp_idx = (key, torch.tensor([nk - 2, 0, 2, 1, 3], device=pred_keypoints.device),
                            torch.tensor([nk - 1, 4, 6, 5, 7], device=pred_keypoints.device))
        ky = pred_keypoints[:, :, 1]
        heights = ky.index_select(1, self._kp_idx[1]) - ky.index_select(1, self._kp_idx[2])
        fh = f_u * pred_height_3D
        d = fh.unsqueeze(-1) / (F.relu(heights) * self.down_ratio + self.EPS)
        depths = torch.cat((d[:, :1], d[:, 1:].reshape(-1, 2, 2).mean(dim=2)), dim=1)      # centre, corners 0/2, corners 1/3
        return torch.clamp(depths, min=self.depth_range[0], max=self.depth_range[1])

    def decode_dimension(self, cls_id, dims_offset):
        if self.dim_modes[0] == 'None':
            return dims_offset
        cls_id = cls_id.flatten().long()
        mean = self.dim_mean.to(dims_offset.device)[cls_id, :]
        if self.dim_modes[0] == 'exp':
            dims_offset = dims_offset.exp()
        if self.dim_modes[2]:
            return dims_offset * self.dim_std.to(dims_offset.device)[cls_id, :] + mean
        return dims_offset * mean

    def decode_axes_orientation(self, vector_ori, locations):
        """(rotys, alphas) from the multi-bin / head-axis encoding and the viewing ray (anno_encoder.py:254-304)."""
        centers = self.alpha_centers.to(vector_ori.device)
        if self.multibin:
            nb = self.orien_bin_size
            bin_cls = torch.softmax(vector_ori[:, : nb * 2].view(-1, nb, 2), dim=2)[..., 1]
            best = bin_cls.argmax(dim=1)
            off = vector_ori[:, nb * 2:].view(-1, nb, 2).gather(1, best.view(-1, 1, 1).expand(-1, 1, 2)).squeeze(1)
            orientations = torch.atan2(off[:, 0], off[:, 1]) + centers[best]
        else:
            axis_cls = torch.softmax(vector_ori[:, :2], dim=1)
            axis_cls = axis_cls[:, 0] < axis_cls[:, 1]
            head_cls = torch.softmax(vector_ori[:, 2:4], dim=1)
            head_cls = head_cls[:, 0] < head_cls[:, 1]
            orientations = centers[axis_cls.long() + head_cls.long() * 2]
            sin_cos_offset = F.normalize(vector_ori[:, 4:])
            orientations = orientations + torch.atan(sin_cos_offset[:, 0] / sin_cos_offset[:, 1])
        locations = locations.view(-1, 3)
        rays = torch.atan2(locations[:, 0], locations[:, 2])
        alphas = orientations
        rotys = alphas + rays

        def wrap(a):
            a = torch.where(a > PI, a - 2 * PI, a)
            return torch.where(a < -PI, a + 2 * PI, a)
        return wrap(rotys), wrap(alphas)

    def decode_kpts_2d(self, kpts_2d, bbox_2d):
        half_w = (bbox_2d[:, 2] - bbox_2d[:, 0]) / 2.
        half_h = (bbox_2d[:, 3] - bbox_2d[:, 1]) / 2.
        return kpts_2d * torch.stack((half_w, half_h), dim=1).unsqueeze(1)

    def decode_pairs_kpts_depth(self, kps, kps_3d, rot_y, K, training=False, kpts_2d_mask=None, gt_depth=None, weight=None):
        """Edge-constraint depth for every keypoint pair (anno_encoder.py:326-390): (N,1500)+mask in training,
        (N,2628) otherwise.  One kernel launch; differentiable w.r.t. kps and kps_3d."""
        return ops.pairs_kpts_depth(kps, kps_3d, rot_y, K, training=training, kpts_2d_mask=kpts_2d_mask)

    def decode_kpts_2d_img(self, kpts_2d, bbox_points, offset_3D, pad_size):
        return (kpts_2d + (bbox_points + offset_3D).unsqueeze(1).expand_as(kpts_2d)) * 4 - pad_size
