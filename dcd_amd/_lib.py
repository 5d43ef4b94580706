"""ctypes loader for libdcd_hip.so (the C ABI in include/dcd_hip.h).

There is no CPU fallback: importing this module never fails, but the first call of `lib()`
raises if the shared library has not been built (`make -C dcd_amd/csrc` or `__graft_entry__.build()`).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdcd_hip.so")
_LIB = None

c_void_p, c_int, c_float, c_size_t, c_int64, c_double = (ctypes.c_void_p, ctypes.c_int, ctypes.c_float,
                                                         ctypes.c_size_t, ctypes.c_int64, ctypes.c_double)


class LossRowsArgs(ctypes.Structure):
    """`dcd_loss_rows_args` of include/dcd_hip.h, field for field."""
    _INTS = ("B", "M", "C", "K", "NP", "num_classes", "ch_box2d", "ch_offset", "ch_corner", "ch_corner_unc", "ch_dims",
             "ch_ori_cls", "ch_ori_off", "ch_depth", "ch_depth_unc", "ch_kpts2d", "ch_kpts3d", "trunc_log")
    _FLOATS = ("depth_lo", "depth_hi", "unc_lo", "unc_hi", "depth_weight")
    _POINTERS = ("pois", "reg_mask", "trunc_mask", "find_pcl", "ori_mask", "cls_ids", "centers", "pad_size", "bboxes",
                 "locations", "rotys", "offset_3D", "dimensions", "orientations", "keypoints", "kp_depth_mask", "kpts2d",
                 "kpts3d", "calib_P", "calib", "dim_mean", "kps_pred", "kps_tgt", "kps3d_pred", "kps3d_tgt", "rot", "P_rows",
                 "kmask", "pair_depth", "pair_mask", "cols", "corners_pred", "corners_tgt", "iou3d", "sums", "grad_sums",
                 "grad_pois", "grad_pair", "grad_kps", "grad_kps3d")
    _fields_ = ([(n, c_int) for n in _INTS] + [(n, c_float) for n in _FLOATS] + [("dim_weight", c_float * 3)]
                + [("down_ratio", c_float), ("kd_eps", c_float)] + [(n, c_void_p) for n in _POINTERS])

HEADS_MAX = 16


class HeadRowsArgs(ctypes.Structure):
    """`dcd_head_rows_args` of include/dcd_hip.h, field for field."""
    _fields_ = ([(n, c_int) for n in ("n_heads", "T", "R", "K", "C")]
                + [(n, c_int * HEADS_MAX) for n in ("trunk", "ch0", "out")]
                + [("weight", c_void_p * HEADS_MAX), ("bias", c_void_p * HEADS_MAX), ("feat", c_void_p), ("y", c_void_p),
                   ("grad_y", c_void_p), ("grad_feat", c_void_p), ("grad_weight", c_void_p * HEADS_MAX),
                   ("grad_bias", c_void_p * HEADS_MAX)])


# name -> (restype, argtypes); mirrors include/dcd_hip.h one to one
SIGNATURES = {
    "dcd_version": (ctypes.c_char_p, []),
    "dcd_dcn_v2_workspace_bytes": (c_size_t, [c_int] * 14),
    "dcd_dcn_v2_forward": (c_int, [c_void_p] * 7 + [c_int] * 15 + [c_void_p, c_size_t]),
    "dcd_dcn_v2_backward": (c_int, [c_void_p] * 12 + [c_int] * 15 + [c_void_p, c_size_t]),
    "dcd_dcn_v2_forget": (c_int, [c_void_p]),
    "dcd_dcn_v2_policy_state": (c_int, [c_void_p, c_void_p]),
    "dcd_dcn_v2_policy_free": (c_int, []),
    "dcd_dcn_v2_set_handover": (c_int, [c_int]),
    "dcd_edge_depth_forward": (c_int, [c_void_p] * 6 + [c_int, c_int, c_int, c_float, c_float, c_int, c_int]
                               + [c_void_p] * 3),
    "dcd_edge_depth_backward": (c_int, [c_void_p] * 7 + [c_int, c_int, c_int, c_float, c_float, c_int]
                                + [c_void_p] * 2),
    "dcd_focal_loss": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_void_p, c_void_p]),
    "dcd_giou_loss": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "dcd_nms_hm": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dcd_heatmap_topk": (c_int, [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p] * 5 + [c_void_p, c_size_t]),
    "dcd_heatmap_topk_workspace_bytes": (c_size_t, [c_int] * 5),
    "dcd_poi_gather": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 5 + [c_void_p]),
    "dcd_poi_scatter_add": (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 5 + [c_void_p]),
    "dcd_patch_scatter_add": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int, c_int, c_void_p]),
    "dcd_iou3d": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "dcd_conv3x3_workspace_bytes": (c_size_t, [c_int] * 5),
    "dcd_conv3x3": (c_int, [c_void_p] * 6 + [c_int] * 6 + [c_void_p, c_size_t]),
    "dcd_conv3x3_weights_bytes": (c_size_t, [c_int] * 3),
    "dcd_conv3x3_transform_weights": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "dcd_conv3x3_transform_weights_table": (c_int, [c_void_p, c_void_p, c_int]),
    "dcd_conv3x3_prepared": (c_int, [c_void_p] * 6 + [c_int] * 6 + [c_void_p, c_size_t]),
    "dcd_conv3x3_split_weights_bytes": (c_size_t, [c_int] * 3),
    "dcd_conv3x3_split_transform_weights": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "dcd_conv3x3_split_transform_weights_table": (c_int, [c_void_p, c_void_p, c_int]),
    "dcd_conv3x3_split_workspace_bytes": (c_size_t, [c_int] * 5),
    "dcd_conv3x3_split_prepared": (c_int, [c_void_p] * 6 + [c_int] * 7 + [c_void_p, c_size_t]),
    "dcd_conv1x1_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64]),
    "dcd_conv1x1_wrw_bf16_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int64]),
    "dcd_conv1x1_wrw_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int64, c_void_p, c_size_t]),
    "dcd_conv3x3_s2_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int]),
    "dcd_conv3x3_s2_f32_backward_data": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int]),
    "dcd_conv3x3_s2_f32_wrw_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "dcd_conv3x3_s2_f32_wrw": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t]),
    "dcd_clip_adamw_workspace_bytes": (c_size_t, [c_int, c_void_p]),
    "dcd_clip_grad_norm_scalars": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_float, c_void_p, c_size_t, c_void_p]),
    "dcd_adamw_apply": (c_int, [c_void_p, c_int] + [c_void_p] * 7 + [c_double] * 4 + [c_void_p]),
    "dcd_conv1x1_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64]),
    "dcd_conv1x1_wrw_f32_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int64]),
    "dcd_conv1x1_wrw_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int64, c_void_p, c_size_t]),
    "dcd_conv3x3_bf16_weights_bytes": (c_size_t, [c_int] * 3),
    "dcd_conv3x3_bf16_transform_weights": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "dcd_conv3x3_bf16_transform_weights_table": (c_int, [c_void_p, c_void_p, c_int]),
    "dcd_conv3x3_bf16_workspace_bytes": (c_size_t, [c_int] * 5),
    "dcd_conv3x3_bf16_prepared": (c_int, [c_void_p] * 6 + [c_int] * 6 + [c_void_p, c_size_t]),
    "dcd_conv_stem_workspace_bytes": (c_size_t, [c_int] * 3),
    "dcd_conv_stem": (c_int, [c_void_p] * 4 + [c_int] * 7 + [c_void_p, c_size_t]),
    "dcd_conv_stem_wrw_workspace_bytes": (c_size_t, [c_int] * 3),
    "dcd_conv_stem_wrw": (c_int, [c_void_p] * 4 + [c_int] * 6 + [c_void_p, c_size_t]),
    "dcd_context_norm_forward": (c_int, [c_void_p] * 4 + [c_int, c_int, c_float]),
    "dcd_context_norm_backward": (c_int, [c_void_p] * 5 + [c_int, c_int]),
    "dcd_sum_tensors": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int64]),
    "dcd_conv3x3_wrw_workspace_bytes": (c_size_t, [c_int] * 5),
    "dcd_conv3x3_wrw": (c_int, [c_void_p] * 4 + [c_int] * 6 + [c_void_p, c_size_t]),
    "dcd_upsample_dw_forward": (c_int, [c_void_p] * 4 + [c_int] * 5),
    "dcd_upsample_dw_forward_add": (c_int, [c_void_p] * 5 + [c_int] * 5),
    "dcd_maxpool2x2_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int]),
    "dcd_maxpool2x2_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int]),
    "dcd_upsample_dw_backward": (c_int, [c_void_p] * 6 + [c_int] * 5),
    "dcd_bn_workspace_bytes": (c_size_t, [c_int]),
    "dcd_bn_stats": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p, c_void_p, c_size_t]),
    "dcd_channel_sums": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p, c_void_p, c_size_t]),
    "dcd_bn_train_apply": (c_int, [c_void_p] * 6 + [c_double] + [c_void_p] * 3 + [c_float, c_float, c_int] + [c_void_p] * 3
                           + [c_int, c_int, c_int64]),
    "dcd_bn_eval_apply": (c_int, [c_void_p] * 7 + [c_float, c_int, c_void_p, c_int, c_int, c_int64]),
    "dcd_bn_backward_stats": (c_int, [c_void_p] * 5 + [c_int, c_int, c_int64, c_void_p, c_void_p, c_size_t]),
    "dcd_trunk_row_sums": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "dcd_trunk_finalize_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_double, c_int, c_void_p, c_void_p, c_void_p]),
    "dcd_trunk_finalize_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_int, c_void_p, c_void_p, c_void_p]),
    "dcd_trunk_grad_wg": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "dcd_bn_backward_relu_from_x": (c_int, [c_void_p] * 10 + [c_int, c_int, c_int64, c_void_p, c_size_t]),
    "dcd_bn_backward_stats_params_relu_from_x": (c_int, [c_void_p] * 7 + [c_int, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "dcd_bn_backward_apply_relu_from_x": (c_int, [c_void_p] * 8 + [c_double, c_void_p, c_int, c_int, c_int64]),
    "dcd_bn_backward_stats_params": (c_int, [c_void_p] * 6 + [c_int, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t]),
    "dcd_bn_backward_apply": (c_int, [c_void_p] * 8 + [c_double] + [c_void_p] * 4 + [c_int, c_int, c_int64]),
    "dcd_bn_at_forward": (c_int, [c_void_p] * 6 + [c_double] + [c_void_p] * 3 + [c_float, c_float, c_int] + [c_void_p] * 4
                          + [c_int, c_int, c_int64, c_int] + [c_void_p, c_size_t]),
    "dcd_bn_at_backward_sums": (c_int, [c_void_p] * 7 + [c_int, c_int, c_int] + [c_void_p, c_void_p]),
    "dcd_bn_train_forward": (c_int, [c_void_p] * 8 + [c_float, c_float, c_int] + [c_void_p] * 3 + [c_int, c_int, c_int64]
                             + [c_void_p, c_size_t]),
    "dcd_bn_backward": (c_int, [c_void_p] * 11 + [c_int, c_int, c_int64] + [c_void_p, c_size_t]),
    "dcd_dcn_offset_mask_split": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64]),
    "dcd_dcn_offset_mask_merge": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64]),
    "dcd_sgemm_shifted": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int,
                                  c_int64, c_int64, c_int, c_int, c_int, c_int, c_int]),
    "dcd_sgemm": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, c_int, c_int64, c_int, c_void_p, c_int, c_int64,
                          c_int, c_int, c_int, c_int, c_float, c_int, c_int]),
    "dcd_spd_solve_workspace_bytes": (c_size_t, [c_int, c_int]),
    "dcd_spd_solve": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t]),
    "dcd_head_rows_forward": (c_int, [c_void_p, ctypes.POINTER(HeadRowsArgs)]),
    "dcd_head_rows_backward": (c_int, [c_void_p, ctypes.POINTER(HeadRowsArgs)]),
    "dcd_loss_rows_prepare": (c_int, [c_void_p, ctypes.POINTER(LossRowsArgs)]),
    "dcd_loss_rows_forward": (c_int, [c_void_p, ctypes.POINTER(LossRowsArgs)]),
    "dcd_loss_rows_backward": (c_int, [c_void_p, ctypes.POINTER(LossRowsArgs)]),
    "dcd_loss_rows_finish": (c_int, [c_void_p, ctypes.POINTER(LossRowsArgs)]),
    "dcd_encode_targets": (c_int, [c_void_p] * 6 + [c_int] * 6 + [c_double] * 3 + [c_int, c_int, c_void_p, c_int]),
}

STATUS = {1: "bad argument", 2: "workspace too small", 3: "kernel launch failed"}


class DcdHipError(RuntimeError):
    pass


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise DcdHipError(
                "libdcd_hip.so is not built (%s missing). The DGDE hot path has no CPU fallback; "
                "build it with `make -C dcd_amd/csrc` or `__graft_entry__.build()`." % LIB_PATH)
        # PyTorch-ROCm bundles its own libamdhip64; it must be the HIP runtime already in the process when our
        # library is loaded, otherwise the kernels register with a second runtime and every launch on a torch stream fails.
        import torch  # noqa: F401
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = handle
    return _LIB


def check(status, what):
    if status != 0:
        raise DcdHipError("%s failed: %s (status %d)" % (what, STATUS.get(status, "unknown"), status))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream_of(t):
    """Raw hipStream_t of torch's current stream on the tensor's device (the direct binding: `torch.cuda.current_stream()`
    builds a Stream object per call, 6 us x 1100 calls per step)."""
    import torch
    return torch._C._cuda_getCurrentRawStream(t.device.index)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise DcdHipError("dcd_amd ops run on the GPU only (got a %s tensor); there is no CPU path" % t.device)
