"""nn.Conv2d whose 3x3 / stride 1 / pad 1 case runs on the Winograd MFMA kernels (csrc/conv.hip), with or without a bias
(with: DCN's `conv_offset_mask`, DGDE/model/backbone/DCNv2/dcn_v2.py:107-116; other biased layers keep the stock op for
everything but the bias gradient).

Same parameters and state-dict keys as `torch.nn.Conv2d`; every other configuration (and CPU tensors -- the CPU test
suite) takes the stock op, which is what the reference uses everywhere (DGDE/model/backbone/dla_dcn.py:76-82,
DGDE/model/head/detector_predictor.py:52-58).  `DCD_CONV_WINOGRAD=0` switches the kernel off (A/B timing)."""
import os

import torch
from torch import nn

from dcd_amd import ops

_ENABLED = os.environ.get("DCD_CONV_WINOGRAD", "1") != "0"
_SKIP = os.environ.get("DCD_CONV_SKIP_NODE", "1") != "0"     # 0: conv + identity skip as two consumers of x (A/B timing)
_STEM = os.environ.get("DCD_CONV_STEM", "1") != "0"          # 0: the two stem convolutions stay on the stock op (A/B timing)


class Conv2d(nn.Conv2d):
    def forward(self, x):
        if (_ENABLED and self.bias is None and self.kernel_size == (3, 3) and self.stride == (1, 1) and self.padding == (1, 1)
                and self.dilation == (1, 1) and self.groups == 1 and self.padding_mode == "zeros"
                and ops.conv3x3_supported(x, self.weight)):
            return ops.conv3x3(x, self.weight)
        if (_ENABLED and self.bias is None and self.kernel_size == (3, 3) and self.stride == (1, 1) and self.padding == (1, 1)
                and self.dilation == (1, 1) and self.groups == 1 and self.padding_mode == "zeros" and torch.is_grad_enabled()
                and self.weight.requires_grad and ops.conv3x3_wrw_only_supported(x, self.weight)):
            return ops.conv3x3_stock_forward(x, self.weight)
        if (_ENABLED and self.bias is None and self.kernel_size == (3, 3) and self.stride == (2, 2) and self.padding == (1, 1)
                and self.dilation == (1, 1) and self.groups == 1 and self.padding_mode == "zeros"
                and ops.conv3x3_stride2_native_supported(x, self.weight)):
            return ops.conv3x3_stride2_native(x, self.weight)              # exact fp32: csrc/conv_s2_f32.inc (round 6)
        if (_ENABLED and self.bias is None and self.kernel_size == (3, 3) and self.stride == (2, 2) and self.padding == (1, 1)
                and self.dilation == (1, 1) and self.groups == 1 and self.padding_mode == "zeros"
                and ops.conv3x3_stride2_supported(x, self.weight)):
            return ops.conv3x3_stride2(x, self.weight)
        if (_ENABLED and _STEM and self.bias is None and self.padding_mode == "zeros" and not isinstance(self.padding, str)
                and ops.conv_stem_supported(x, self.weight, self.stride, self.padding, self.dilation, self.groups)
                and (self.weight.shape[1] == 16 or not x.requires_grad)):
            return ops.conv_stem(x, self.weight)
        if (_ENABLED and self.bias is not None and x.is_cuda and x.dtype == torch.float32 and self.groups == 1
                and self.padding_mode == "zeros" and not isinstance(self.padding, str) and torch.is_grad_enabled()
                and self.bias.requires_grad
                and (x.shape[0] * x.shape[2] * x.shape[3] >= 1 << 16
                     or ops.conv3x3_bias_supported(x, self.weight, self.stride, self.padding, self.dilation))):
            return ops.conv2d_bias(x, self.weight, self.bias, self.stride, self.padding, self.dilation)
        ops.stock_conv_in_capture(self, x)                  # raises for a stride-2 3x3 layer inside a stream capture
        return super().forward(x)


    def forward_with_skip(self, x):
        """(self(x), x): for a block whose skip connection starts at this convolution's input.  On the Winograd path the skip's
        gradient is added inside the input-gradient kernel (ops.conv3x3_with_skip); otherwise x itself is returned."""
        if (_ENABLED and _SKIP and self.bias is None and self.kernel_size == (3, 3) and self.stride == (1, 1) and self.padding == (1, 1)
                and self.dilation == (1, 1) and self.groups == 1 and self.padding_mode == "zeros" and torch.is_grad_enabled()
                and x.requires_grad and ops.conv3x3_supported(x, self.weight)):
            return ops.conv3x3_with_skip(x, self.weight)
        return self.forward(x), x


class DepthwiseUpsample(nn.ConvTranspose2d):
    """`nn.ConvTranspose2d(o, o, 2f, stride=f, padding=f//2, output_padding=0, groups=o, bias=False)` of IDAUp
    (DGDE/model/backbone/dla_dcn.py:416-418) on csrc/upsample.hip; same parameter / state-dict key (`weight`).  MIOpen has no
    solver for this shape and runs naive / im2col kernels (5.1 ms per step for the eight layers)."""

    def forward_add(self, x, skip):
        """up(x) + skip; on the HIP path the sum leaves the up-sampling kernel (IDAUp: node(up(proj(x)) + skip))."""
        f = self.stride[0]
        if self._hip_ok(x, None) and skip.is_cuda and skip.dtype == torch.float32:
            return ops.upsample_dw(x, self.weight, f, skip)
        return self.forward(x) + skip

    def _hip_ok(self, x, output_size):
        f = self.stride[0]
        return (_ENABLED and output_size is None and x.is_cuda and x.dtype == torch.float32 and self.bias is None
                and self.groups == self.in_channels == self.out_channels and self.stride == (f, f) and f in (2, 4, 8)
                and self.kernel_size == (2 * f, 2 * f) and self.padding == (f // 2, f // 2) and self.output_padding == (0, 0)
                and self.dilation == (1, 1) and (x.shape[3] * f) % 4 == 0 and x.shape[0] * x.shape[1] <= 65535)

    def forward(self, x, output_size=None):
        f = self.stride[0]
        if (_ENABLED and output_size is None and x.is_cuda and x.dtype == torch.float32 and self.bias is None
                and self.groups == self.in_channels == self.out_channels and self.stride == (f, f) and f in (2, 4, 8)
                and self.kernel_size == (2 * f, 2 * f) and self.padding == (f // 2, f // 2) and self.output_padding == (0, 0)
                and self.dilation == (1, 1) and (x.shape[3] * f) % 4 == 0 and x.shape[0] * x.shape[1] <= 65535):
            return ops.upsample_dw(x, self.weight, f)
        return super().forward(x, output_size)


class MaxPool2x2(nn.MaxPool2d):
    """`nn.MaxPool2d(2, stride=2)` of the DLA trees (DGDE/model/backbone/dla_dcn.py:228) on csrc/upsample.hip; anything else the
    module is configured for, and inputs the kernel does not take, go to the stock op."""

    def forward(self, x):
        def two(v):
            return v == 2 or v == (2, 2)
        if (_ENABLED and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and two(self.kernel_size) and two(self.stride)
                and self.padding in (0, (0, 0)) and self.dilation in (1, (1, 1)) and not self.ceil_mode and not self.return_indices
                and x.shape[2] % 2 == 0 and x.shape[3] % 4 == 0):
            return ops.maxpool2x2(x)
        return super().forward(x)
