"""DLA-34 encoder + DCN up-sampling decoder (mirrors DGDE/model/backbone/dla_dcn.py:20-465).

Module tree and parameter names equal the reference's so its checkpoints load:
`base.{base_layer,level0..5}`, `dla_up.ida_{0,1,2}.{proj,up,node}_k`, `ida_up.{proj,up,node}_k`, with
`DeformConv = {conv: DCN, actf: [BN, ReLU]}`.  Ordinary convolutions / BN / transposed convolutions run on
stock PyTorch-ROCm (MIOpen); every DeformConv runs the HIP DCNv2 kernels and every BN (+residual) (+ReLU) chain
runs the fused HIP normalisation kernels (csrc/norm.hip).
"""
import math
import os

import numpy as np
import torch
from torch import nn

from .DCNv2.dcn_v2 import DCN
from dcd_amd import ops
from dcd_amd.model.layers.norm import BatchNorm2d
from dcd_amd.model.layers.conv import Conv2d, DepthwiseUpsample, MaxPool2x2

BN_MOMENTUM = 0.1
_ROOT_SPLIT = os.environ.get("DCD_ROOT_SPLIT", "1") != "0"      # 0: torch.cat + stock 1x1 convolution in Root (A/B timing)


def _bn(c, relu=False):
    """The reference's `nn.BatchNorm2d(c, momentum=BN_MOMENTUM)`; `relu=True` folds the ReLU (and, at call time, the
    residual add) that follows it into the same HIP kernel (dcd_amd/model/layers/norm.py)."""
    return BatchNorm2d(c, momentum=BN_MOMENTUM, fuse_relu=relu)


def _relu_slot():
    """Placeholder where the reference has `nn.ReLU(inplace=True)` after a BN inside an nn.Sequential: the ReLU runs inside
    the preceding BN kernel; the slot keeps the Sequential indices (state-dict keys) of the reference."""
    return nn.Identity()


class BasicBlock(nn.Module):
    """Two 3x3 convs with an externally supplied residual (dla_dcn.py:71-101)."""

    def __init__(self, inplanes, planes, stride=1, dilation=1):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, 3, stride=stride, padding=dilation, bias=False, dilation=dilation)
        self.bn1 = _bn(planes, relu=True)
        self.relu = nn.ReLU(inplace=True)       # kept for structural parity; executed inside bn1 / bn2
        self.conv2 = Conv2d(planes, planes, 3, stride=1, padding=dilation, bias=False, dilation=dilation)
        self.bn2 = _bn(planes, relu=True)
        self.stride = stride

    def forward(self, x, residual=None):
        if residual is None or residual is x:
            y, residual = self.conv1.forward_with_skip(x)   # identity skip: its gradient joins conv1's inside the kernel
        else:
            y = self.conv1(x)
        out = self.bn1(y)                                  # bn + relu
        return self.bn2(self.conv2(out), residual)         # bn + residual + relu


class Root(nn.Module):
    """1x1 conv over the concatenated children (dla_dcn.py:187-207)."""

    def __init__(self, in_channels, out_channels, kernel_size, residual):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, 1, stride=1, bias=False, padding=(kernel_size - 1) // 2)
        self.bn = _bn(out_channels, relu=True)
        self.relu = nn.ReLU(inplace=True)       # executed inside self.bn
        self.residual = residual

    def forward(self, *x):
        c = self.conv
        if (_ROOT_SPLIT and x[0].is_cuda and x[0].dtype == torch.float32 and c.kernel_size == (1, 1) and c.stride == (1, 1)
                and c.padding == (0, 0) and c.bias is None and c.groups == 1):
            y = ops.conv1x1_of_cat(x, c.weight)                 # sum_i W_i x_i: the concatenation is never formed
        else:
            y = c(torch.cat(x, 1))
        return self.bn(y, x[0] if self.residual else None)


class Tree(nn.Module):
    """Recursive aggregation node (dla_dcn.py:210-260)."""

    def __init__(self, levels, block, in_channels, out_channels, stride=1, level_root=False, root_dim=0,
                 root_kernel_size=1, dilation=1, root_residual=False):
        super().__init__()
        if root_dim == 0:
            root_dim = 2 * out_channels
        if level_root:
            root_dim += in_channels
        if levels == 1:
            self.tree1 = block(in_channels, out_channels, stride, dilation=dilation)
            self.tree2 = block(out_channels, out_channels, 1, dilation=dilation)
            self.root = Root(root_dim, out_channels, root_kernel_size, root_residual)
        else:
            self.tree1 = Tree(levels - 1, block, in_channels, out_channels, stride, root_dim=0,
                              root_kernel_size=root_kernel_size, dilation=dilation, root_residual=root_residual)
            self.tree2 = Tree(levels - 1, block, out_channels, out_channels, root_dim=root_dim + out_channels,
                              root_kernel_size=root_kernel_size, dilation=dilation, root_residual=root_residual)
        self.level_root = level_root
        self.root_dim = root_dim
        self.levels = levels
        self.downsample = MaxPool2x2(stride, stride=stride) if stride > 1 else None
        self.project = None
        if in_channels != out_channels:
            self.project = nn.Sequential(nn.Conv2d(in_channels, out_channels, 1, stride=1, bias=False), _bn(out_channels))
        # For levels > 1 the projected residual is handed to `tree1`, which is itself a Tree and recomputes its own
        # (dla_dcn.py:246-250 of the reference): this module's `project` then never receives a gradient.  It is still
        # executed so its BN buffers evolve exactly as in the reference; DDP is told about it (engine/trainer.py).
        self.dead_project = levels > 1 and self.project is not None

    def forward(self, x, residual=None, children=None):
        children = [] if children is None else children
        bottom = self.downsample(x) if self.downsample else x
        if self.project is not None and _ROOT_SPLIT and bottom.is_cuda and bottom.dtype == torch.float32:
            # the 1x1 projection as a (one-input) `conv1x1_of_cat` like the Roots: batched GEMMs in exact fp32, the bf16 pointwise
            # kernel (csrc/conv1x1_bf16.inc) under MODEL.FP16 -- no MIOpen call, no NHWC transposes
            residual = self.project[1](ops.conv1x1_of_cat([bottom], self.project[0].weight))
        else:
            residual = self.project(bottom) if self.project else bottom
        if self.level_root:
            children.append(bottom)
        x1 = self.tree1(x, residual)
        if self.levels == 1:
            return self.root(self.tree2(x1), x1, *children)
        children.append(x1)
        return self.tree2(x1, children=children)


class DLA(nn.Module):
    """Encoder: returns the six maps at strides 1..32 (dla_dcn.py:263-332)."""

    def __init__(self, levels, channels, num_classes=1000, block=BasicBlock, residual_root=False):
        super().__init__()
        self.channels = channels
        self.num_classes = num_classes
        self.base_layer = nn.Sequential(Conv2d(3, channels[0], 7, stride=1, padding=3, bias=False),
                                        _bn(channels[0], relu=True), _relu_slot())
        self.level0 = self._make_conv_level(channels[0], channels[0], levels[0])
        self.level1 = self._make_conv_level(channels[0], channels[1], levels[1], stride=2)
        self.level2 = Tree(levels[2], block, channels[1], channels[2], 2, level_root=False, root_residual=residual_root)
        self.level3 = Tree(levels[3], block, channels[2], channels[3], 2, level_root=True, root_residual=residual_root)
        self.level4 = Tree(levels[4], block, channels[3], channels[4], 2, level_root=True, root_residual=residual_root)
        self.level5 = Tree(levels[5], block, channels[4], channels[5], 2, level_root=True, root_residual=residual_root)

    @staticmethod
    def _make_conv_level(inplanes, planes, convs, stride=1, dilation=1):
        mods = []
        for i in range(convs):
            mods += [Conv2d(inplanes, planes, 3, stride=stride if i == 0 else 1, padding=dilation, bias=False,
                            dilation=dilation), _bn(planes, relu=True), _relu_slot()]
            inplanes = planes
        return nn.Sequential(*mods)

    def forward(self, x):
        maps = []
        x = self.base_layer(x)
        for i in range(6):
            x = getattr(self, "level{}".format(i))(x)
            maps.append(x)
        return maps

    def load_pretrained_model(self, pretrain_path):
        """Load ImageNet / DD3D encoder weights from a local file (dla_dcn.py:333-357).  A checkpoint that carries
        the ImageNet classifier gets the unused `fc` attached, like the reference, so keys line up."""
        weights = torch.load(pretrain_path, map_location="cpu")
        if isinstance(weights, dict) and "model" in weights:  # DD3D layout (:333-341)
            prefix = "backbone.bottom_up."
            weights = {k.replace(prefix, ""): v for k, v in weights["model"].items() if prefix in k}
        elif any(k.startswith("fc.") for k in weights):
            n_cls = len(weights[list(weights.keys())[-1]])
            self.fc = nn.Conv2d(self.channels[-1], n_cls, kernel_size=1, stride=1, padding=0, bias=True)
        self.load_state_dict(weights)


def dla34(pretrained=True, pretrain_path=None, **kwargs):
    model = DLA([1, 1, 1, 2, 2, 1], [16, 32, 64, 128, 256, 512], block=BasicBlock, **kwargs)
    if pretrained:
        if pretrain_path is None:
            raise RuntimeError("MODEL.PRETRAIN needs MODEL.PRETRAIN_PATH: this environment has no network to fetch "
                               "the ImageNet DLA-34 weights the reference downloads (dla_dcn.py:61-62)")
        model.load_pretrained_model(pretrain_path)
    return model


CONV_BODIES = {"dla34": dla34}


def fill_up_weights(up):
    """Bilinear-kernel initialisation of a depthwise ConvTranspose2d (dla_dcn.py:386-395)."""
    w = up.weight.data
    f = math.ceil(w.size(2) / 2)
    c = (2 * f - 1 - f % 2) / (2.0 * f)
    for i in range(w.size(2)):
        for j in range(w.size(3)):
            w[0, 0, i, j] = (1 - math.fabs(i / f - c)) * (1 - math.fabs(j / f - c))
    w[1:, 0] = w[0, 0]


class DeformConv(nn.Module):
    def __init__(self, chi, cho):
        super().__init__()
        self.actf = nn.Sequential(_bn(cho, relu=True), _relu_slot())
        self.conv = DCN(chi, cho, kernel_size=(3, 3), stride=1, padding=1, dilation=1, deformable_groups=1)

    def forward(self, x):
        return self.actf(self.conv(x))


class IDAUp(nn.Module):
    """Iterative deep aggregation: project (DCN) -> upsample (depthwise deconv) -> add -> node (DCN)
    (dla_dcn.py:412-438).  Mutates `layers` in place, exactly as the reference does."""

    def __init__(self, o, channels, up_f):
        super().__init__()
        for i in range(1, len(channels)):
            f = int(up_f[i])
            up = DepthwiseUpsample(o, o, f * 2, stride=f, padding=f // 2, output_padding=0, groups=o, bias=False)
            fill_up_weights(up)
            # registration order proj, up, node == the reference's state_dict key order
            setattr(self, "proj_" + str(i), DeformConv(channels[i], o))
            setattr(self, "up_" + str(i), up)
            setattr(self, "node_" + str(i), DeformConv(o, o))

    def forward(self, layers, startp, endp):
        for i in range(startp + 1, endp):
            k = str(i - startp)
            # layers[i] = up(proj(layers[i])); layers[i] = node(layers[i] + layers[i - 1])   (dla_dcn.py:433-436): the sum is
            # formed inside the up-sampling kernel
            up = getattr(self, "up_" + k)
            summed = up.forward_add(getattr(self, "proj_" + k)(layers[i]), layers[i - 1])
            layers[i] = getattr(self, "node_" + k)(summed)


class DLAUp(nn.Module):
    def __init__(self, startp, channels, scales, in_channels=None):
        super().__init__()
        self.startp = startp
        if in_channels is None:
            in_channels = channels
        self.channels = channels
        channels = list(channels)
        in_channels = list(in_channels)
        scales = np.array(scales, dtype=int)
        for i in range(len(channels) - 1):
            j = -i - 2
            setattr(self, "ida_{}".format(i), IDAUp(channels[j], in_channels[j:], scales[j:] // scales[j]))
            scales[j + 1:] = scales[j]
            in_channels[j + 1:] = [channels[j] for _ in channels[j + 1:]]

    def forward(self, layers):
        out = [layers[-1]]
        for i in range(len(layers) - self.startp - 1):
            getattr(self, "ida_{}".format(i))(layers, len(layers) - i - 2, len(layers))
            out.insert(0, layers[-1])
        return out


class DLASeg(nn.Module):
    def __init__(self, base_name, pretrained, pretrain_path, down_ratio, last_level):
        super().__init__()
        assert down_ratio in [2, 4, 8, 16]
        self.first_level = int(np.log2(down_ratio))
        self.last_level = last_level
        self.base = CONV_BODIES[base_name](pretrained=pretrained, pretrain_path=pretrain_path)
        channels = self.base.channels
        scales = [2 ** i for i in range(len(channels[self.first_level:]))]
        self.dla_up = DLAUp(self.first_level, channels[self.first_level:], scales)
        self.out_channels = channels[self.first_level]
        self.ida_up = IDAUp(self.out_channels, channels[self.first_level:self.last_level],
                            [2 ** i for i in range(self.last_level - self.first_level)])

    def forward(self, x):
        x = self.dla_up(self.base(x))
        # the reference clones the three maps here (dla_dcn.py:56); IDAUp only REBINDS list entries (layers[i] = ...), it never
        # writes into them, so the copies (110 MB at bs 8) are left out: same values, same gradients
        y = [x[i] for i in range(self.last_level - self.first_level)]
        self.ida_up(y, 0, len(y))
        return y[-1]


def build_backbone(cfg):
    return DLASeg(base_name=cfg.MODEL.BACKBONE.CONV_BODY, pretrained=cfg.MODEL.PRETRAIN,
                  pretrain_path=cfg.MODEL.PRETRAIN_PATH, down_ratio=cfg.MODEL.BACKBONE.DOWN_RATIO, last_level=5)
