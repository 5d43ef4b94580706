"""Training targets of a whole batch encoded ON THE DEVICE.

The reference builds every sample's targets with numpy inside the data-loader workers (`KITTIDataset.__getitem__`,
DGDE/data/datasets/kitti.py:354-606) and ships ~30 arrays per image through the collate / `.to(device)` path.  Here the loader
only has to parse the label files: the raw values of a batch (what `Object3d` holds, kitti_utils.py:61-113) go to the GPU as three
small arrays and ONE launch (csrc/targets.hip) produces every ParamsList field as a batch tensor -- box / key-point projection,
visibility, truncated-object centres, Gaussian heat map, multi-bin orientation, border walk.  `encode_targets` returns the
`ParamsList` objects the model consumes (same field names, dtypes and shapes as the reference's).

There is no CPU path: the numpy restatement lives in oracle/target_oracle.py as test infrastructure."""
import ctypes

import numpy as np
import torch

from dcd_amd import _lib
from dcd_amd.data.calibration import Calibration
from dcd_amd.structures.params_3d import ParamsList

# (name, trailing shape, dtype) in the order of the C ABI's `outputs` (include/dcd_hip.h)
_OUTPUTS = (
    ("hm", None, torch.float32), ("cls_ids", (), torch.int32), ("target_centers", (2,), torch.int32), ("gt_bboxes", (4,), torch.float32),
    ("2d_bboxes", (4,), torch.float32), ("keypoints", (10, 3), torch.float32), ("keypoints_depth_mask", (3,), torch.float32),
    ("extra_kpts_2d", ("K", 3), torch.float32), ("extra_kpts_3d", ("K", 3), torch.float32), ("extra_kpts_depth_mask", ("K",), torch.float32),
    ("Calib_P", (3, 4), torch.float32), ("find_pcl", (), torch.uint8), ("dimensions", (3,), torch.float32), ("locations", (3,), torch.float32),
    ("rotys", (), torch.float32), ("alphas", (), torch.float32), ("offset_3D", (2,), torch.float32), ("occlusions", (), torch.float64),
    ("truncations", (), torch.float64), ("orientations", (8,), torch.float32), ("reg_mask", (), torch.uint8), ("trunc_mask", (), torch.uint8),
    ("reg_weight", (), torch.float32),
)


def pack_raw(samples, max_objs, n_extra):
    """Host side: list of per-image dicts (image_size, P, trunc_occ (n,2), box2d (n,4), hwl (n,3), t (n,3), ry, alpha, find_pcl,
    kpts3d (n,n_extra,3), optional cls (n)) -> the three packed arrays of the C ABI (+ sizes and counts)."""
    B = len(samples)
    objs = np.zeros((B, max_objs, 16), np.float64)
    kpts = np.zeros((B, max_objs, n_extra, 3), np.float64)
    P = np.zeros((B, 3, 4), np.float64)
    size = np.zeros((B, 2), np.int32)
    count = np.zeros(B, np.int32)
    for b, s in enumerate(samples):
        n = len(s["ry"])
        if n > max_objs:
            raise ValueError("image %d has %d objects, DATASETS.MAX_OBJECTS is %d" % (b, n, max_objs))
        count[b] = n
        P[b] = s["P"]
        size[b] = s["image_size"]
        o = objs[b, :n]
        o[:, 0:2] = s["trunc_occ"]
        o[:, 2:6] = np.asarray(s["box2d"], np.float32)
        o[:, 6:9] = s["hwl"]
        o[:, 9:12] = np.asarray(s["t"], np.float32)
        o[:, 12], o[:, 13], o[:, 14] = s["ry"], s["alpha"], s["find_pcl"]
        o[:, 15] = s.get("cls", 0)
        kpts[b, :n] = s["kpts3d"]
    return objs, kpts, P, size, count


def encode_targets(samples, cfg, device, img_ids=None):
    """Raw label values of a batch -> [ParamsList] with every training field (kitti.py:572-606), computed on `device`."""
    device = torch.device(device)
    if device.type != "cuda":
        raise _lib.DcdHipError("dcd_amd.data.target_encoder runs on the GPU only; there is no CPU path")
    ok = (cfg.INPUT.HEATMAP_CENTER == '3D' and cfg.INPUT.ORIENTATION == 'multi-bin' and cfg.INPUT.ORIENTATION_BIN_SIZE == 4
          and cfg.INPUT.KEYPOINT_VISIBLE_MODIFY and cfg.INPUT.ADJUST_BOUNDARY_HEATMAP and cfg.DATASETS.CONSIDER_OUTSIDE_OBJS
          and cfg.INPUT.APPROX_3D_CENTER == 'intersect' and cfg.DATASETS.FILTER_ANNO_ENABLE)
    if not ok:
        raise NotImplementedError("the device encoder implements the DGDE.yaml configuration of the target encoding")
    M, n_extra = cfg.DATASETS.MAX_OBJECTS, cfg.MODEL.HEAD.EXTRA_KPTS_NUM
    in_w, in_h, down = cfg.INPUT.WIDTH_TRAIN, cfg.INPUT.HEIGHT_TRAIN, cfg.MODEL.BACKBONE.DOWN_RATIO
    n_cls = cfg.DATASETS.MAX_CLASSES_NUM
    B, K = len(samples), n_extra + 10
    fw, fh = in_w // down, in_h // down
    packed = pack_raw(samples, M, n_extra)
    dev_in = [torch.from_numpy(a).pin_memory().to(device, non_blocking=True) for a in packed]
    out = {}
    for name, tail, dtype in _OUTPUTS:
        shape = (B, n_cls, fh, fw) if tail is None else (B, M) + tuple(K if d == "K" else d for d in tail)
        out[name] = torch.zeros(shape, dtype=dtype, device=device)
    out["pad_size"] = torch.zeros((B, 2), dtype=torch.int64, device=device)
    out["edge_indices"] = torch.zeros((B, (fw + fh) * 2, 2), dtype=torch.int64, device=device)
    out["edge_len"] = torch.zeros((B,), dtype=torch.int64, device=device)
    order = [n for n, _, _ in _OUTPUTS] + ["pad_size", "edge_indices", "edge_len"]
    ptrs = (ctypes.c_void_p * len(order))(*[out[n].data_ptr() for n in order])
    L = _lib.lib()
    ft, fs = cfg.DATASETS.FILTER_ANNOS
    _lib.check(L.dcd_encode_targets(_lib.stream_of(out["hm"]), *[t.data_ptr() for t in dev_in], B, M, n_extra, in_w, in_h, down,
                                    float(ft), float(fs), float(cfg.INPUT.HEATMAP_RATIO), 4, n_cls, ptrs, len(order)),
               "dcd_encode_targets")
    out["find_pcl"] = out["find_pcl"].bool()
    ori_mask = torch.ones((B, M), dtype=torch.bool, device=device)
    targets = []
    for b, s in enumerate(samples):
        t = ParamsList(image_size=(in_w, in_h), is_train=True)        # the padded size, like the reference (kitti.py:572)
        for name in order[:-3]:
            t.add_field(name, out[name][b])
        t.add_field("ori_mask", ori_mask[b])
        t.add_field("pad_size", out["pad_size"][b])
        t.add_field("calib", Calibration(np.asarray(s["P"], np.float64)))
        t.add_field("edge_indices", out["edge_indices"][b])
        t.add_field("edge_len", out["edge_len"][b])
        t.add_field("final_output_w", torch.tensor(fw))
        t.add_field("final_output_h", torch.tensor(fh))
        t.add_field("img_idx", img_ids[b] if img_ids is not None else "%06d" % b)
        targets.append(t)
    return targets
