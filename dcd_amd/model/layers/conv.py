"""nn.Conv2d whose 3x3 / stride 1 / pad 1 / no-bias case runs on the Winograd MFMA kernel (csrc/conv.hip).

Same parameters and state-dict keys as `torch.nn.Conv2d`; every other configuration (and CPU tensors -- the CPU test
suite) takes the stock op, which is what the reference uses everywhere (DGDE/model/backbone/dla_dcn.py:76-82,
DGDE/model/head/detector_predictor.py:52-58).  `DCD_CONV_WINOGRAD=0` switches the kernel off (A/B timing)."""
import os

from torch import nn

from dcd_amd import ops

_ENABLED = os.environ.get("DCD_CONV_WINOGRAD", "1") != "0"


class Conv2d(nn.Conv2d):
    def forward(self, x):
        if (_ENABLED and self.bias is None and self.kernel_size == (3, 3) and self.stride == (1, 1) and self.padding == (1, 1)
                and self.dilation == (1, 1) and self.groups == 1 and self.padding_mode == "zeros"
                and ops.conv3x3_supported(x, self.weight)):
            return ops.conv3x3(x, self.weight)
        return super().forward(x)
