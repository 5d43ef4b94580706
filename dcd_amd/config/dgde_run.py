"""The DGDE experiment (the one run configuration the reference ships, DGDE/runs/DGDE.yaml) as overrides of the defaults.

Written as tables rather than as a YAML tree: the ten regression heads with their channel counts, and the thirteen loss
terms with their initial weights, are each ONE list here and are unzipped into the two parallel config keys the reference's
code reads (`REGRESSION_HEADS` / `REGRESSION_CHANNELS`, `LOSS_NAMES` / `INIT_LOSS_WEIGHT`).  `get_cfg()` merges these by
default; `get_cfg(yaml_file=path)` merges a YAML file (e.g. the reference's own) instead.
"""

# (head name, output channels); heads that share a 3x3 trunk are grouped  (DGDE.yaml:27-28; channel order of the `reg` map)
REGRESSION_TRUNKS = [
    [("2d_dim", 4)],
    [("3d_offset", 2)],
    [("corner_offset", 20)],
    [("corner_uncertainty", 3)],
    [("3d_dim", 3)],
    [("ori_cls", 8), ("ori_offset", 8)],
    [("depth", 1)],
    [("depth_uncertainty", 1)],
    [("extra_kpts_2d", 146)],          # 73 keypoints x (u, v)
    [("extra_kpts_3d", 219)],          # 73 keypoints x (x, y, z)
]

# (config name of the loss term, initial weight)  (DGDE.yaml:43-44)
LOSS_TERMS = [
    ("hm_loss", 1), ("bbox_loss", 1), ("depth_loss", 0.2), ("offset_loss", 0.6), ("orien_loss", 1), ("dims_loss", 0.33),
    ("corner_loss", 0.025), ("keypoint_loss", 0.02), ("keypoint_depth_loss", 0.066), ("trunc_offset_loss", 0.6),
    ("extra_kpts_2d_loss", 1.0), ("extra_kpts_3d_loss", 1.0), ("pairs_kpts_depth_loss", 0.3),
]


def dgde_overrides():
    """Flat [key, value, key, value, ...] list for `CfgNode.merge_from_list`."""
    data = {
        "DATASETS.DETECT_CLASSES": ("Car",), "DATASETS.MAX_CLASSES_NUM": 1,
        "DATASETS.TRAIN": ("kitti_train",), "DATASETS.TEST": ("kitti_train",),
        "DATASETS.TRAIN_SPLIT": "train", "DATASETS.TEST_SPLIT": "val",
        "DATASETS.CONSIDER_OUTSIDE_OBJS": True, "DATASETS.FILTER_ANNO_ENABLE": True,
    }
    target_encoding = {
        "INPUT.HEATMAP_CENTER": "3D", "INPUT.APPROX_3D_CENTER": "intersect", "INPUT.ADJUST_BOUNDARY_HEATMAP": True,
        "INPUT.KEYPOINT_VISIBLE_MODIFY": True, "INPUT.AUG_PARAMS": [[0.5]],
        "INPUT.ORIENTATION": "multi-bin", "INPUT.ORIENTATION_BIN_SIZE": 4, "INPUT.MODIFY_ALPHA": False,
    }
    head = {
        "EXTRA_KPTS_NUM": 63,                                             # + 10 box points = 73 keypoints per object
        "REGRESSION_HEADS": [[name for name, _ in trunk] for trunk in REGRESSION_TRUNKS],
        "REGRESSION_CHANNELS": [[ch for _, ch in trunk] for trunk in REGRESSION_TRUNKS],
        "USE_NORMALIZATION": "BN", "BN_MOMENTUM": 0.1, "UNCERTAINTY_INIT": True,
        "ENABLE_EDGE_FUSION": True, "EDGE_FUSION_NORM": "BN", "TRUNCATION_OUTPUT_FUSION": "add",
        "HEATMAP_TYPE": "centernet", "CENTER_MODE": "max",
        "DIMENSION_REG": ["exp", True, False], "DIMENSION_WEIGHT": [1, 1, 1],
        "OUTPUT_DEPTH": "edges", "CORNER_LOSS_DEPTH": "edges", "MODIFY_INVALID_KEYPOINT_DEPTH": True, "USE_UNCERTAINTY": False,
        "LOSS_TYPE": ["Penalty_Reduced_FocalLoss", "L1", "giou", "L1"], "TRUNCATION_OFFSET_LOSS": "log",
        "LOSS_NAMES": [name for name, _ in LOSS_TERMS],
        "INIT_LOSS_WEIGHT": [w for _, w in LOSS_TERMS],
    }
    model = {"MODEL.REDUCE_LOSS_NORM": True, "MODEL.USE_SYNC_BN": True}
    model.update({"MODEL.HEAD." + k: v for k, v in head.items()})
    schedule = {
        "SOLVER.OPTIMIZER": "adamw", "SOLVER.BASE_LR": 3e-4, "SOLVER.WEIGHT_DECAY": 1e-5, "SOLVER.IMS_PER_BATCH": 8,
        "SOLVER.LR_WARMUP": True, "SOLVER.WARMUP_STEPS": 2000,
        "SOLVER.MAX_EPOCHS": 100.0, "SOLVER.DECAY_EPOCH_STEPS": [80.0, 90.0], "SOLVER.LR_DECAY": 0.1,
        "SOLVER.SAVE_CHECKPOINT_EPOCH_INTERVAL": 20.0, "SOLVER.EVAL_INTERVAL": 1000,
    }
    evaluation = {"TEST.DETECTIONS_THRESHOLD": 0.2, "TEST.UNCERTAINTY_AS_CONFIDENCE": True, "TEST.METRIC": ["R40"]}
    flat = []
    for group in (data, target_encoding, model, schedule, evaluation):
        for key, value in group.items():
            flat += [key, value]
    return flat
