"""DCNv2 modules on top of the HIP backend (mirrors DGDE/model/backbone/DCNv2/dcn_v2.py:16-128).

`_DCNv2` calls `_backend.dcn_v2_forward/backward` with the reference's positional arguments; `_backend` is
`dcd_amd._ext` (the C-ABI binding).  There is no CPU implementation: a CPU tensor raises.
State-dict keys are the reference's: `weight`, `bias`, `conv_offset_mask.{weight,bias}`.
"""
import math
import os

import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from dcd_amd import _ext as _backend
from dcd_amd.model.layers.conv import Conv2d


def _precision_scope():
    from dcd_amd import _ext
    return _ext.get_precision()


class _DCNv2(Function):
    """Python-side autograd node of the op: argument order (input, offset, mask, weight, bias, ...) as in the reference's
    `_DCNv2` (dcn_v2.py:16-54); the extension takes (input, weight, bias, offset, mask, kh, kw, sh, sw, ph, pw, dh, dw, dg)."""

    @staticmethod
    def forward(ctx, input, offset, mask, weight, bias, stride, padding, dilation, deformable_groups, precision=None):
        kernel = tuple(weight.shape[2:4])
        ctx.geometry = (*kernel, *_pair(stride), *_pair(padding), *_pair(dilation), int(deformable_groups))
        # None: the current precision scope -- remembered for the backward, which runs outside the scope.  Exact fp32 passes no
        # keyword at all: `_backend` then only needs the reference's own signature (the host tests bind the oracle here)
        p = _precision_scope() if precision is None else precision
        ctx.precision = {} if p == "f32" else {"precision": p}
        ctx.save_for_backward(input, offset, mask, weight, bias)
        return _backend.dcn_v2_forward(input, weight, bias, offset, mask, *ctx.geometry, **ctx.precision)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        input, offset, mask, weight, bias = ctx.saved_tensors
        grads = _backend.dcn_v2_backward(input.contiguous(), weight.contiguous(), bias, offset, mask, grad_output.contiguous(),
                                         *ctx.geometry, **ctx.precision)
        grad_input, grad_offset, grad_mask, grad_weight, grad_bias = grads
        return (grad_input, grad_offset, grad_mask, grad_weight, grad_bias) + (None,) * 5


_FUSED_GLUE = os.environ.get("DCD_OFFSET_MASK_FUSED", "1") != "0"      # 0: slice + sigmoid through stock ops (A/B timing)


def _offset_mask_split(out):
    from dcd_amd import _lib
    B, C3, H, W = out.shape
    T = C3 // 3
    offset = torch.empty((B, 2 * T, H, W), dtype=out.dtype, device=out.device)
    mask = torch.empty((B, T, H, W), dtype=out.dtype, device=out.device)
    st = _lib.lib().dcd_dcn_offset_mask_split(_lib.stream_of(out), out.data_ptr(), offset.data_ptr(), mask.data_ptr(), B, T, H * W)
    _lib.check(st, "dcd_dcn_offset_mask_split")
    return offset, mask


def _offset_mask_merge(goff, gmask, mask):
    from dcd_amd import _lib
    B, T, H, W = mask.shape
    goff, gmask = goff.contiguous(), gmask.contiguous()
    gout = torch.empty((B, 3 * T, H, W), dtype=mask.dtype, device=mask.device)
    st = _lib.lib().dcd_dcn_offset_mask_merge(_lib.stream_of(mask), goff.data_ptr(), gmask.data_ptr(), mask.data_ptr(),
                                              gout.data_ptr(), B, T, H * W)
    _lib.check(st, "dcd_dcn_offset_mask_merge")
    return gout


class _OffsetMask(torch.autograd.Function):
    """out (B, 3T, H, W) of `conv_offset_mask` -> (offset (B, 2T, H, W) contiguous, mask = sigmoid(last T channels)); the backward
    assembles grad_out in one pass (csrc/dcn_v2.hip: dcn_offset_mask_split / _merge)."""

    @staticmethod
    def forward(ctx, out):
        offset, mask = _offset_mask_split(out)
        ctx.save_for_backward(mask)
        return offset, mask

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, goff, gmask):
        (mask,) = ctx.saved_tensors
        return _offset_mask_merge(goff, gmask, mask)


_ONE_NODE = os.environ.get("DCD_DCN_NODE", "1") != "0"                 # 0: offset conv, split and DCN as three nodes (A/B timing)


class _DCNWithOffsets(Function):
    """`DCN.forward` (dcn_v2.py:117-128) as ONE autograd node: offset conv -> split / sigmoid -> deformable conv.  The input feeds
    both the offset conv and the deformable conv, so as separate nodes autograd adds two full-size input gradients per layer
    (16 additions per step, 30 us each on the 64-channel 96x320 maps); here the offset conv's input gradient is accumulated into
    the deformable conv's by the Winograd kernel's output transform (`dcd_conv3x3(..., residual = output)`).  Same kernels, same
    arithmetic otherwise (the one addition happens in the other order)."""

    @staticmethod
    def forward(ctx, input, w_off, b_off, weight, bias, geometry):
        from dcd_amd import ops
        input, w_off, weight = input.contiguous(), w_off.contiguous(), weight.contiguous()
        ctx.cprec = ops._conv_prec(input)                      # the backward runs outside the forward's precision scope
        p = _precision_scope()
        ctx.precision = {} if p == "f32" else {"precision": p}
        tw, ctx.tw_back = (ops.conv3x3_step_weights(w_off, input) if ops._PREP_BOTH and ctx.needs_input_grad[0] else (None, None))
        out = ops._conv3x3_call(input, w_off, w_off.shape[0], False, b_off.contiguous(), transformed=tw, prec=ctx.cprec)
        offset, mask = _offset_mask_split(out)
        ctx.geometry = geometry
        ctx.save_for_backward(input, offset, mask, weight, bias, w_off)
        return _backend.dcn_v2_forward(input, weight, bias, offset, mask, *geometry, **ctx.precision)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        from dcd_amd import ops
        input, offset, mask, weight, bias, w_off = ctx.saved_tensors
        grad_input, grad_offset, grad_mask, grad_weight, grad_bias = _backend.dcn_v2_backward(
            input, weight, bias, offset, mask, grad_output.contiguous(), *ctx.geometry, **ctx.precision)
        gout = _offset_mask_merge(grad_offset, grad_mask, mask)
        if ctx.needs_input_grad[0]:
            grad_input = ops._conv3x3_call(gout, w_off, w_off.shape[1], True, residual=grad_input.contiguous(), transformed=ctx.tw_back,
                                           prec=ctx.cprec)
        gw_off = ops._conv3x3_wrw_call(input, gout, w_off.shape, ctx.cprec) if ctx.needs_input_grad[1] else None
        gb_off = ops.channel_sums(gout) if ctx.needs_input_grad[2] else None
        return grad_input, gw_off, gb_off, grad_weight, grad_bias, None


def dcn_v2_conv(input, offset, mask, weight, bias, stride, padding, dilation, deformable_groups):
    """`_DCNv2.apply` with the reference's arguments.  Inside a torch.autocast region (someone else's: MODEL.FP16 itself is a
    precision scope on the device, model/detector.py) the op stays an fp32 op at its boundary like the reference's (its extension
    reads `.data<float>()`, cuda/dcn_v2_cuda.cu:58): half-precision activations are cast back to fp32, and the weight contraction
    takes the mixed-precision matrix path (DCD_PREC_BF16: fp32 in, fp32 out, operands rounded to bf16, fp32 accumulate)."""
    dev = input.device.type
    if torch.is_autocast_enabled(dev):
        with torch.autocast(device_type=dev, enabled=False):
            args = (input.float(), offset.float(), mask.float(), weight.float(), bias.float(), stride, padding, dilation,
                    deformable_groups)
            # the split-bf16 contraction exists on the device only; the host-logic tests run the fp32 oracle behind the cast
            return _DCNv2.apply(*args, "bf16") if input.is_cuda else _DCNv2.apply(*args)
    return _DCNv2.apply(input, offset, mask, weight, bias, stride, padding, dilation, deformable_groups)


class DCNv2(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, deformable_groups=1):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride = _pair(stride)
        self.padding = _pair(padding)
        self.dilation = _pair(dilation)
        self.deformable_groups = deformable_groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, *self.kernel_size))
        self.bias = nn.Parameter(torch.empty(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        # U(-1/sqrt(fan_in), 1/sqrt(fan_in)) weight, zero bias (dcn_v2.py:75-81)
        fan_in = self.in_channels * self.kernel_size[0] * self.kernel_size[1]
        bound = 1.0 / math.sqrt(fan_in)
        self.weight.data.uniform_(-bound, bound)
        self.bias.data.zero_()

    # The library keeps a per-layer launch policy keyed by the weight's device address (include/dcd_hip.h, dcd_dcn_v2_forget): the
    # module owns that address, so it says when the key dies -- when its parameters move (.to / .cuda / .float: `_apply`) and when
    # it is destroyed.  Another tensor the caching allocator later places at the same address then starts as "unknown".
    def _forget_policy(self):
        w = self._parameters.get("weight") if hasattr(self, "_parameters") else None
        if w is None or not w.is_cuda:
            return
        try:
            from dcd_amd import _lib
            with torch.cuda.device(w.device):
                _lib.lib().dcd_dcn_v2_forget(w.data_ptr())
        except Exception:                         # interpreter shutdown, library not built: nothing to forget
            pass

    def _apply(self, fn, *args, **kwargs):
        self._forget_policy()
        return super()._apply(fn, *args, **kwargs)

    def __del__(self):
        self._forget_policy()

    def forward(self, input, offset, mask):
        taps = self.deformable_groups * self.kernel_size[0] * self.kernel_size[1]
        assert offset.shape[1] == 2 * taps and mask.shape[1] == taps
        return dcn_v2_conv(input, offset, mask, self.weight, self.bias, self.stride, self.padding, self.dilation,
                           self.deformable_groups)


class DCN(DCNv2):
    """DCNv2 that predicts its own offsets and mask with a zero-initialised conv (dcn_v2.py:97-128)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation=1, deformable_groups=1):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, deformable_groups)
        taps = self.deformable_groups * self.kernel_size[0] * self.kernel_size[1]
        self.conv_offset_mask = Conv2d(self.in_channels, 3 * taps, kernel_size=self.kernel_size,
                                       stride=self.stride, padding=self.padding, bias=True)
        self.init_offset()

    def init_offset(self):
        self.conv_offset_mask.weight.data.zero_()
        self.conv_offset_mask.bias.data.zero_()

    def forward(self, input):
        from dcd_amd import ops
        com = self.conv_offset_mask
        if (_ONE_NODE and _FUSED_GLUE and input.dtype == torch.float32 and not torch.is_autocast_enabled(input.device.type)
                and com.bias is not None and com.groups == 1 and com.padding_mode == "zeros" and not isinstance(com.padding, str)
                and ops.conv3x3_bias_supported(input, com.weight, com.stride, com.padding, com.dilation)
                and ops._WRW_ENABLED and ops._OFFSET_CONV_FWD):
            geometry = (*self.kernel_size, *self.stride, *self.padding, *self.dilation, int(self.deformable_groups))
            return _DCNWithOffsets.apply(input, com.weight, com.bias, self.weight, self.bias, geometry)
        out = com(input)
        taps2 = out.shape[1] // 3 * 2
        # chunk(3) then cat(o1, o2) is the identity on the first 2/3 of the channels (dcn_v2.py:120-121)
        if out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and _FUSED_GLUE:
            offset, mask = _OffsetMask.apply(out)               # one launch (and one in the backward) instead of eight
        else:
            offset = out[:, :taps2]
            mask = torch.sigmoid(out[:, taps2:])
        return dcn_v2_conv(input, offset, mask, self.weight, self.bias, self.stride, self.padding, self.dilation,
                           self.deformable_groups)
