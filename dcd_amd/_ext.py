"""Drop-in for the reference's compiled `_ext` module (DGDE/model/backbone/DCNv2/DCN/src/vision.cpp:4-9).

`dcn_v2_forward` / `dcn_v2_backward` keep the reference's exact positional signatures
(DCN/src/dcn_v2.h:9-24 and :48-59: weight and bias come BEFORE offset and mask), allocate and return
new tensors like the reference (cuda/dcn_v2_cuda.cu:89-91, :247-255, return order :338-340), raise
RuntimeError on shape mismatch (:60-84), and enqueue on the current stream without syncing.
The PS-ROI pooling entry points of the reference module are not on the DGDE path and are not provided.
"""
import torch

from . import _lib

import threading

PRECISION = {"f32": 0, "bf16x3": 1, "bf16": 2}
_default_precision = "f32"             # process default (set_precision)
_scoped = threading.local()            # the innermost precision_scope of THIS thread, if any: a forward running on another thread
                                       # inside someone else's scope keeps its own precision


def set_precision(name):
    """Select the MFMA path used for the weight contractions of the DCNv2 op and of the 3x3 convolutions:
    "f32" exact, "bf16x3" split-bf16 (~2^-16 per product), "bf16" mixed precision -- operands rounded to bf16, one
    product on the bf16 matrix cores, fp32 accumulate (MODEL.FP16, DGDE/model/detector.py:34-36).  The process default; a
    `precision_scope` of the calling thread takes precedence while it is open."""
    global _default_precision
    if name not in PRECISION:
        raise ValueError("precision must be one of %s" % sorted(PRECISION))
    _default_precision = name


def get_precision():
    return getattr(_scoped, "name", None) or _default_precision


class precision_scope:
    """`with precision_scope("bf16"):` -- the contraction precision of every op whose FORWARD runs inside the block on this thread
    (autograd nodes remember it for their backward).  What MODEL.FP16 wraps around the backbone and the predictor where the
    reference has `torch.cuda.amp.autocast()` (DGDE/model/detector.py:34-36, head/detector_head.py:20-22)."""

    def __init__(self, name):
        if name is not None and name not in PRECISION:
            raise ValueError("precision must be one of %s" % sorted(PRECISION))
        self.name = name

    def __enter__(self):
        self.saved = getattr(_scoped, "name", None)
        if self.name is not None:
            _scoped.name = self.name
        return self

    def __exit__(self, *exc):
        _scoped.name = self.saved
        return False


HANDOVER = {"never": 0, "always": 1, "auto": 2, None: -1}


def set_handover(mode):
    """Pin the DCNv2 backward's far-sample launch policy for this process ("never" / "always" / "auto"; None: back to the
    DCD_DCN_HANDOVER environment default) -- include/dcd_hip.h "Per-layer launch policy".  Tests that compare two launch sequences
    of the same step (a captured graph against an eager twin) pin it so that both take the same kernels."""
    if mode not in HANDOVER:
        raise ValueError("handover mode must be one of 'never', 'always', 'auto', None")
    _lib.check(_lib.lib().dcd_dcn_v2_set_handover(HANDOVER[mode]), "dcd_dcn_v2_set_handover")


def _check(input, weight, bias, offset, mask, kernel_h, kernel_w):
    _lib.require_cuda(input, weight, bias, offset, mask)
    for name, t in (("input", input), ("weight", weight), ("bias", bias), ("offset", offset), ("mask", mask)):
        if t.dtype != torch.float32:
            raise RuntimeError("%s must be float32 (the op is fp32-only, like the reference)" % name)
    if input.dim() != 4 or weight.dim() != 4:
        raise RuntimeError("input and weight must be 4-D")
    if weight.shape[2] != kernel_h or weight.shape[3] != kernel_w:
        raise RuntimeError("Input shape and kernel shape wont match: (%d x %d vs %d x %d)." % (
            kernel_h, kernel_w, weight.shape[2], weight.shape[3]))
    if input.shape[1] != weight.shape[1]:
        raise RuntimeError("Input shape and kernel channels wont match: (%d vs %d)." % (input.shape[1], weight.shape[1]))
    if bias.numel() != weight.shape[0]:
        raise RuntimeError("bias must have Cout elements")


def _dims(input, weight, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w):
    B, C, H, W = input.shape
    Ho = (H + 2 * pad_h - (dilation_h * (kernel_h - 1) + 1)) // stride_h + 1
    Wo = (W + 2 * pad_w - (dilation_w * (kernel_w - 1) + 1)) // stride_w + 1
    return B, C, H, W, weight.shape[0], Ho, Wo


def _workspace(L, input, geom):
    nbytes = L.dcd_dcn_v2_workspace_bytes(*geom)
    if nbytes == 0:
        raise RuntimeError("dcn_v2: unsupported geometry %s" % (geom,))
    return torch.empty(nbytes, dtype=torch.uint8, device=input.device), nbytes


def dcn_v2_forward(input, weight, bias, offset, mask, kernel_h, kernel_w, stride_h, stride_w,
                   pad_h, pad_w, dilation_h, dilation_w, deformable_group, precision=None):
    _check(input, weight, bias, offset, mask, kernel_h, kernel_w)
    L = _lib.lib()
    B, C, H, W, Co, Ho, Wo = _dims(input, weight, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w,
                                   dilation_h, dilation_w)
    if tuple(offset.shape) != (B, 2 * deformable_group * kernel_h * kernel_w, Ho, Wo):
        raise RuntimeError("offset shape %s does not match (B, 2*dg*kh*kw, Ho, Wo)" % (tuple(offset.shape),))
    if tuple(mask.shape) != (B, deformable_group * kernel_h * kernel_w, Ho, Wo):
        raise RuntimeError("mask shape %s does not match (B, dg*kh*kw, Ho, Wo)" % (tuple(mask.shape),))
    input, weight, bias, offset, mask = (t.contiguous() for t in (input, weight, bias, offset, mask))
    geom = (B, C, H, W, Co, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w,
            deformable_group)
    ws, nbytes = _workspace(L, input, geom)
    output = torch.empty((B, Co, Ho, Wo), dtype=torch.float32, device=input.device)
    prec = PRECISION[precision or get_precision()]
    st = L.dcd_dcn_v2_forward(_lib.stream_of(input), input.data_ptr(), weight.data_ptr(), bias.data_ptr(),
                              offset.data_ptr(), mask.data_ptr(), output.data_ptr(), *geom, prec,
                              ws.data_ptr(), nbytes)
    _lib.check(st, "dcd_dcn_v2_forward")
    return output


def dcn_v2_backward(input, weight, bias, offset, mask, grad_output, kernel_h, kernel_w, stride_h, stride_w,
                    pad_h, pad_w, dilation_h, dilation_w, deformable_group, precision=None):
    _check(input, weight, bias, offset, mask, kernel_h, kernel_w)
    _lib.require_cuda(grad_output)
    # the reference insists on contiguous input and weight (cuda/dcn_v2_cuda.cu:219-220)
    if not input.is_contiguous():
        raise RuntimeError("input tensor has to be contiguous")
    if not weight.is_contiguous():
        raise RuntimeError("weight tensor has to be contiguous")
    L = _lib.lib()
    B, C, H, W, Co, Ho, Wo = _dims(input, weight, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w,
                                   dilation_h, dilation_w)
    if tuple(grad_output.shape) != (B, Co, Ho, Wo):
        raise RuntimeError("grad_output shape %s does not match the output" % (tuple(grad_output.shape),))
    if grad_output.dtype != torch.float32:
        raise RuntimeError("grad_output must be float32")
    bias, offset, mask, grad_output = (t.contiguous() for t in (bias, offset, mask, grad_output))
    geom = (B, C, H, W, Co, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w,
            deformable_group)
    ws, nbytes = _workspace(L, input, geom)
    grad_input = torch.empty_like(input)
    grad_offset = torch.empty_like(offset)
    grad_mask = torch.empty_like(mask)
    grad_weight = torch.empty_like(weight)
    grad_bias = torch.empty_like(bias)
    prec = PRECISION[precision or get_precision()]
    st = L.dcd_dcn_v2_backward(_lib.stream_of(input), input.data_ptr(), weight.data_ptr(), bias.data_ptr(),
                               offset.data_ptr(), mask.data_ptr(), grad_output.data_ptr(),
                               grad_input.data_ptr(), grad_offset.data_ptr(), grad_mask.data_ptr(),
                               grad_weight.data_ptr(), grad_bias.data_ptr(), *geom, prec, ws.data_ptr(), nbytes)
    _lib.check(st, "dcd_dcn_v2_backward")
    return [grad_input, grad_offset, grad_mask, grad_weight, grad_bias]
