"""Small layer helpers (DGDE/model/make_layers.py:34-49).  Unlike the reference these do not read a global cfg
at import time; the group count is passed in."""
from torch import nn


def group_norm(out_channels, num_groups=32):
    return nn.GroupNorm(num_groups if out_channels % 32 == 0 else num_groups // 2, out_channels)


def _fill_fc_weights(layers, value=0):
    for m in layers.modules():
        if isinstance(m, nn.Conv2d) and m.bias is not None:
            nn.init.constant_(m.bias, value)
