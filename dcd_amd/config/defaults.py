"""Default configuration tree `_C` with the reference's key names and default values
(DGDE/config/defaults.py:9-380).  Written as one nested literal; `cfg` is the global instance the
reference modules import (`from config import cfg`, DGDE/config/__init__.py:1).
"""
import os

from .cfgnode import CfgNode

_DEFAULTS = {
    "MODEL": {
        "DEVICE": "cuda", "WEIGHT": "", "PRETRAIN": True, "PRETRAIN_PATH": None, "USE_SYNC_BN": False,
        "REDUCE_LOSS_NORM": True, "NORM": "BN", "INPLACE_ABN": False, "FP16": False, "FINETUNE": False,
        "FREEZE_BACKBONE_EPOCH": -1, "FREEZE_BACKBONE_STEPS": -1, "FREEZE_NAME": [],
        "BATCH_WEIGHT_FACTOR": 18, "ATTR_AND_VELO": False,
        "BACKBONE": {"CONV_BODY": "dla34", "FREEZE_CONV_BODY_AT": 0, "DOWN_RATIO": 4, "TYPE": "DGDE",
                     "USE_FPN": False, "FPN_STRIDE": [8, 16, 32, 64, 128]},
        "GROUP_NORM": {"DIM_PER_GP": -1, "NUM_GROUPS": 32, "EPSILON": 1e-5},
        "HEAD": {
            "PREDICTOR": "Base_Predictor", "CENTER_AGGREGATION": False, "EXTRA_KPTS_NUM": 63,
            "USE_2D_3D_KPTS_LOC_LOSS": False, "KPTS_3D_EGO": False, "DEEPER_HEAD": False, "STACKED_CONVS": 2,
            "DCN_ON_LAST_CONV": True,
            "LOSS_TYPE": ["Penalty_Reduced_FocalLoss", "L1", "giou", "berhu"], "HEATMAP_TYPE": "centernet",
            "LOSS_ALPHA": 0.25, "LOSS_GAMMA": 2, "LOSS_PENALTY_ALPHA": 2, "LOSS_BETA": 4, "NUM_CHANNEL": 256,
            "USE_NORMALIZATION": "BN", "ACTIVE_FUNC": "relu",
            "REGRESSION_HEADS": [["2d_dim"], ["3d_offset"], ["3d_dim"], ["ori_cls", "ori_offset"], ["depth"]],
            "REGRESSION_CHANNELS": [[4], [2], [3], [4, 2], [1]],
            "MODIFY_INVALID_KEYPOINT_DEPTH": False, "BIAS_BEFORE_BN": False, "BN_MOMENTUM": 0.1,
            "UNCERTAINTY_INIT": True, "UNCERTAINTY_RANGE": [-10, 10], "UNCERTAINTY_WEIGHT": 1.0,
            "KEYPOINT_LOSS": "L1", "KEYPOINT_NORM_FACTOR": 1.0, "CORNER_LOSS_DEPTH": "direct",
            "KEYPOINT_XY_WEIGHT": [1, 1], "DEPTH_FROM_KEYPOINT": False, "KEYPOINT_TO_DEPTH_RELU": True,
            "DEPTH_MODE": "inv_sigmoid", "DEPTH_RANGE": [0.1, 100], "DEPTH_REFERENCE": (26.494627, 16.05988),
            "SUPERVISE_CORNER_DEPTH": False,
            "REGRESSION_OFFSET_STAT": [-0.5844396972302358, 9.075032501413093],
            "REGRESSION_OFFSET_STAT_NORMAL": [-0.01571878324572745, 0.05915441457040611],
            "USE_UNCERTAINTY": False,
            "LOSS_NAMES": ["hm_loss", "center_loss", "bbox_loss", "depth_loss", "offset_loss", "orien_loss",
                           "dims_loss", "corner_loss"],
            "LOSS_UNCERTAINTY": [True, True, True, False, False, True, True, True], "INIT_LOSS_WEIGHT": [],
            "REGRESSION_AREA": False, "ENABLE_EDGE_FUSION": False, "EDGE_FUSION_KERNEL_SIZE": 3,
            "EDGE_FUSION_NORM": "BN", "EDGE_FUSION_RELU": False, "TRUNCATION_OFFSET_LOSS": "L1",
            "TRUNCATION_OUTPUT_FUSION": "replace", "TRUNCATION_CLS": False, "OUTPUT_DEPTH": "direct",
            "SCALE_DEPTH_BY_FOCAL_LENGTHS_FACTOR": 800,
            "DIMENSION_MEAN": ((3.8840, 1.5261, 1.6286), (0.8423, 1.7607, 0.6602), (1.7635, 1.7372, 0.5968)),
            "DIMENSION_STD": ((0.4259, 0.1367, 0.1022), (0.2349, 0.1133, 0.1427), (0.1766, 0.0948, 0.1242)),
            "DIMENSION_REG": ["linear", True, False], "DIMENSION_WEIGHT": [1, 1, 1],
            "INIT_P": 0.01, "CENTER_SAMPLE": "center", "CENTER_MODE": "max",
        },
        "DEPTH_REFINE": {"ENABLE": False, "DETACH_DEPTH": True, "USE_EARLY_FEAT": True, "REFINE_THRESH_TYPE": "2D",
                         "REFINE_THRESH": 0.2, "NUM_CHANNEL": [64, 128], "OUTPUT_SIZE": [14, 14], "JITTER": [2, 1],
                         "BIN_NUM": 5, "BIN_SIZE": 1},
    },
    "INPUT": {
        "HEIGHT_TRAIN": 384, "WIDTH_TRAIN": 1280, "HEIGHT_TEST": 384, "WIDTH_TEST": 1280,
        "PIXEL_MEAN": [0.485, 0.456, 0.406], "PIXEL_STD": [0.229, 0.224, 0.225], "TO_BGR": False,
        "MODIFY_ALPHA": False, "USE_APPROX_CENTER": False, "HEATMAP_CENTER": "3D", "ADJUST_DIM_HEATMAP": False,
        "ADJUST_BOUNDARY_HEATMAP": False, "HEATMAP_RATIO": 0.5, "ELLIP_GAUSSIAN": False, "IGNORE_DONT_CARE": False,
        "KEYPOINT_VISIBLE_MODIFY": False, "ALLOW_OUTSIDE_CENTER": False, "APPROX_3D_CENTER": "intersect",
        "ORIENTATION": "head-axis", "ORIENTATION_BIN_SIZE": 4, "AUG_PARAMS": [[0.5]],
        "MULTI_TRAIN_SIZE": ((1120, 640), (1376, 768), (1600, 896), (1824, 1024), (2048, 1152)),
    },
    "DATASETS": {
        "TRAIN": (), "TEST": (), "TRAIN_SPLIT": "", "TEST_SPLIT": "", "DETECT_CLASSES": ("Car", "Pedestrian", "Cyclist"),
        "FILTER_ANNO_ENABLE": False, "FILTER_ANNOS": [0.9, 20], "USE_RIGHT_IMAGE": False,
        "CONSIDER_OUTSIDE_OBJS": False, "MAX_OBJECTS": 40, "MIN_RADIUS": 0.0, "MAX_RADIUS": 0.0,
        "CENTER_RADIUS_RATIO": 0.1, "USE_TTA": False, "TTA_AUG_PARAMS": [[0.0]], "MAX_CLASSES_NUM": 3,
        "INFER_ON_RIGHT_IMG": False,
    },
    "DATALOADER": {"NUM_WORKERS": 8, "SIZE_DIVISIBILITY": 0, "ASPECT_RATIO_GROUPING": False},
    "SOLVER": {
        "OPTIMIZER": "adamw", "BASE_LR": 3e-3, "WEIGHT_DECAY": 1e-5, "MAX_ITERATION": 30000, "MAX_EPOCHS": 70.0,
        "MOMS": [0.95, 0.85], "PCT_START": 0.4, "DIV_FACTOR": 10, "STEPS": (20000, 25000),
        "DECAY_EPOCH_STEPS": [35.0, 45.0], "LR_DECAY": 0.1, "LR_CLIP": 0.0000001, "LR_WARMUP": False,
        "WARMUP_EPOCH": 1, "WARMUP_STEPS": -1, "GRAD_NORM_CLIP": 15, "SAVE_CHECKPOINT_INTERVAL": 1000,
        "EVAL_INTERVAL": 2000, "SAVE_CHECKPOINT_EPOCH_INTERVAL": 5.0, "EVAL_EPOCH_INTERVAL": 2.0,
        "EVAL_AND_SAVE_EPOCH": True, "GRAD_CLIP_FACTOR": 99, "GRAD_ALPHA": 0.9, "BIAS_LR_FACTOR": 2.0,
        "BACKBONE_LR_FACTOR": 1.0, "LOAD_OPTIMIZER_SCHEDULER": True, "IMS_PER_BATCH": 32, "MASTER_BATCH": -1,
    },
    "TEST": {
        "SINGLE_GPU_TEST": True, "IMS_PER_BATCH": 1, "PRED_2D": True, "GENERATE_GMW": False,
        "GEN_DATA_AT_FINISH": False, "UNCERTAINTY_AS_CONFIDENCE": False, "METRIC": ["R40"], "EVAL_DIS_IOUS": False,
        "EVAL_DEPTH": False, "EVAL_DEPTH_METHODS": [], "USE_NMS": "none", "NMS_THRESH": -1.0,
        "NMS_CLASS_AGNOSTIC": False, "DETECTIONS_PER_IMG": 50, "DETECTIONS_THRESHOLD": 0.1,
        "VISUALIZE_THRESHOLD": 1e-2, "USE_ONLY_EXTRA_KPTS": False,
    },
    "OUTPUT_DIR": "./tools/logs", "SEED": -1, "CUDNN_BENCHMARK": True, "START_TIME": 0,
    "PATHS_CATALOG": os.path.join(os.path.dirname(__file__), "paths_catalog.py"),
}

_C = CfgNode(_DEFAULTS)
