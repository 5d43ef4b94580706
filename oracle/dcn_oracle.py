"""ctypes binding of oracle/liboracle.so with the reference's `_ext` signatures.

TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Mirrors the positional signatures of
`dcn_v2_forward` / `dcn_v2_backward` (reference DGDE/model/backbone/DCNv2/DCN/src/dcn_v2.h:9-24,
:48-59; return order DCN/src/cpu/dcn_v2_cpu.cpp:226-228) on CPU torch tensors, so a test can
plug this module in where the reference does `import _ext as _backend`
(DGDE/model/backbone/DCNv2/dcn_v2.py:13).  float32 follows the reference's arithmetic type;
float64 exists for finite-difference gradient checks only.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _suffix(t):
    if t.dtype == torch.float32:
        return "f32"
    if t.dtype == torch.float64:
        return "f64"
    raise TypeError("oracle DCN supports float32/float64, got %s" % t.dtype)


def _out_hw(H, W, kh, kw, sh, sw, ph, pw, dh, dw):
    return ((H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1, (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1)


def dcn_v2_forward(input, weight, bias, offset, mask, kernel_h, kernel_w, stride_h, stride_w,
                   pad_h, pad_w, dilation_h, dilation_w, deformable_group):
    suf = _suffix(input)
    input, weight, bias, offset, mask = [t.detach().contiguous() for t in (input, weight, bias, offset, mask)]
    assert input.device.type == "cpu"
    B, C, H, W = input.shape
    Cout = weight.shape[0]
    # reference asserts (dcn_v2_cpu.cpp:59-63)
    if tuple(weight.shape[2:]) != (kernel_h, kernel_w):
        raise RuntimeError("Input shape and kernel shape wont match")
    if weight.shape[1] != C:
        raise RuntimeError("Input shape and kernel channels wont match")
    Ho, Wo = _out_hw(H, W, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w)
    out = torch.empty(B, Cout, Ho, Wo, dtype=input.dtype)
    rc = getattr(lib(), "dcn_oracle_forward_" + suf)(
        _p(input), _p(weight), _p(bias), _p(offset), _p(mask), B, C, H, W, Cout, kernel_h, kernel_w,
        stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, deformable_group, _p(out))
    if rc:
        raise RuntimeError("dcn_oracle_forward failed rc=%d" % rc)
    return out


def dcn_v2_backward(input, weight, bias, offset, mask, grad_output, kernel_h, kernel_w, stride_h, stride_w,
                    pad_h, pad_w, dilation_h, dilation_w, deformable_group):
    suf = _suffix(input)
    input, weight, bias, offset, mask, grad_output = [
        t.detach().contiguous() for t in (input, weight, bias, offset, mask, grad_output)]
    B, C, H, W = input.shape
    Cout = weight.shape[0]
    gi, go, gm = torch.empty_like(input), torch.empty_like(offset), torch.empty_like(mask)
    gw, gb = torch.empty_like(weight), torch.empty_like(bias)
    rc = getattr(lib(), "dcn_oracle_backward_" + suf)(
        _p(input), _p(weight), _p(bias), _p(offset), _p(mask), _p(grad_output), B, C, H, W, Cout,
        kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, deformable_group,
        _p(gi), _p(go), _p(gm), _p(gw), _p(gb))
    if rc:
        raise RuntimeError("dcn_oracle_backward failed rc=%d" % rc)
    return [gi, go, gm, gw, gb]
