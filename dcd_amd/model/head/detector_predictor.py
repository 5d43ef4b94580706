"""Prediction head (mirrors `_predictor`, DGDE/model/head/detector_predictor.py:18-203).

Module names / shapes equal the reference's so state dicts are interchangeable:
`class_head` (3x3 conv, BN, ReLU, 1x1 conv), `reg_features[i]` (3x3 conv, BN, ReLU), `reg_heads[i][j]` (1x1 conv),
`trunc_heatmap_conv`, `trunc_offset_conv` (Conv1d k3 replicate-pad, BN1d, [ReLU], Conv1d k1).
Differences in execution only: the per-image Python scatter loop of the edge fusion (:193-196, one host sync per
image) is a single masked `index_put_(accumulate=True)`.
"""
import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from dcd_amd.structures.params_3d import stack_field
from dcd_amd import ops
from dcd_amd.model import registry
from dcd_amd.model.layers.utils import select_point_of_interest, sigmoid_hm
from dcd_amd.model.make_layers import group_norm, _fill_fc_weights
from dcd_amd.model.backbone.DCNv2.dcn_v2 import DCN
from dcd_amd.model.layers.norm import BatchNorm2d
from dcd_amd.model.layers.conv import Conv2d
from dcd_amd.model.head import trunk_moments

import os

_EDGE_FAST = os.environ.get("DCD_EDGE_BRANCH_GEMM", "1") != "0"      # 0: the edge-fusion branches on the stock Conv1d / BatchNorm1d (A/B timing)


class EdgeBranch(nn.Sequential):
    """`nn.Sequential(Conv1d(k3, replicate), BatchNorm1d | Identity, ReLU | Identity, Conv1d(k1))` of the edge fusion
    (DGDE/model/head/detector_predictor.py:124-131) -- same modules, same state-dict keys.  On the device the first convolution
    is one GEMM over the unfolded border row (ops.conv1d_k3_replicate) and a training-mode BatchNorm1d + ReLU runs on
    csrc/norm.hip (one launch each way: (B, C, K) is a (B, C, K, 1) map to it); anything else -- host tensors, another kernel
    size, a converted SyncBatchNorm, eval mode -- takes the modules as they are."""

    def forward(self, x):
        conv1, bn, act, conv2 = self[0], self[1], self[2], self[3]
        if not (_EDGE_FAST and x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and isinstance(conv1, nn.Conv1d)
                and conv1.kernel_size == (3,) and conv1.stride == (1,) and conv1.dilation == (1,) and conv1.groups == 1
                and conv1.padding_mode == "replicate" and conv1.padding == (1,) and conv1.bias is not None):
            return super().forward(x)
        y = ops.conv1d_k3_replicate(x, conv1.weight, conv1.bias)
        relu = isinstance(act, nn.ReLU)
        if (type(bn) is nn.BatchNorm1d and bn.training and bn.affine and bn.track_running_stats and bn.momentum is not None
                and isinstance(act, (nn.ReLU, nn.Identity))):
            y = ops.batch_norm_act(y.unsqueeze(-1), None, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                   bn.num_batches_tracked, bn.momentum, bn.eps, relu).squeeze(-1)
        else:
            y = act(bn(y))
        if (isinstance(conv2, nn.Conv1d) and conv2.kernel_size == (1,) and conv2.stride == (1,) and conv2.padding == (0,)
                and conv2.dilation == (1,) and conv2.groups == 1 and conv2.bias is not None):
            # the k1 convolution as the batched GEMM it is: the stock solver ran its weight gradient as an NHWC implicit GEMM between two
            # transposes and a zero fill, with atomics (the last MIOpen rows of the step, and not reproducible run to run)
            w2 = conv2.weight.squeeze(-1)
            return torch.baddbmm(conv2.bias.view(1, -1, 1), w2.unsqueeze(0).expand(y.shape[0], -1, -1), y)
        return conv2(y)
_HEAD_FUSED = os.environ.get("DCD_HEAD_FUSED", "1") != "0"      # 0: stock 1x1 conv + separate gather (A/B timing)
_HEAD_ROWS = os.environ.get("DCD_HEAD_ROWS", "1") != "0"        # 0: one F.linear per regression head (A/B timing, CPU tests)


@registry.PREDICTOR.register("Base_Predictor")
class _predictor(nn.Module):
    def __init__(self, cfg, in_channels):
        super().__init__()
        classes = cfg.DATASETS.MAX_CLASSES_NUM
        if classes != len(cfg.DATASETS.DETECT_CLASSES):
            print('ATTENTION, classes!=len(cfg.DATASETS.DETECT_CLASSES)', classes, len(cfg.DATASETS.DETECT_CLASSES))

        self.regression_head_cfg = cfg.MODEL.HEAD.REGRESSION_HEADS
        self.regression_channel_cfg = cfg.MODEL.HEAD.REGRESSION_CHANNELS
        self.output_width = cfg.INPUT.WIDTH_TRAIN // cfg.MODEL.BACKBONE.DOWN_RATIO
        self.output_height = cfg.INPUT.HEIGHT_TRAIN // cfg.MODEL.BACKBONE.DOWN_RATIO
        self.head_conv = cfg.MODEL.HEAD.NUM_CHANNEL
        self.active_func = cfg.MODEL.HEAD.ACTIVE_FUNC

        use_norm = cfg.MODEL.HEAD.USE_NORMALIZATION
        if use_norm == 'BN':
            # BN with the following ReLU fused in (csrc/norm.hip); the activation slot of the Sequential becomes Identity
            fuse = self.active_func == 'relu'
            self.norm_func = lambda c: BatchNorm2d(c, fuse_relu=fuse)
        elif use_norm == 'GN':
            self.norm_func = lambda c: group_norm(c, cfg.MODEL.GROUP_NORM.NUM_GROUPS)
        else:
            self.norm_func = nn.Identity
        self.fused_bn_relu = use_norm == 'BN' and self.active_func == 'relu'
        self.bn_momentum = cfg.MODEL.HEAD.BN_MOMENTUM

        self.deeper_head = cfg.MODEL.HEAD.DEEPER_HEAD
        self.stacked_convs = cfg.MODEL.HEAD.STACKED_CONVS
        self.dcn_on_last_conv = cfg.MODEL.HEAD.DCN_ON_LAST_CONV
        self.in_channels = in_channels
        trunk_in = self.head_conv if self.deeper_head else self.in_channels

        # --- classification branch
        if self.deeper_head:
            self.cls_head_pre = self._deep_stem()
        self.class_head = nn.Sequential(
            Conv2d(trunk_in, self.head_conv, kernel_size=3, padding=1, bias=False),
            self.norm_func(self.head_conv), self._get_active_func(),
            nn.Conv2d(self.head_conv, classes, kernel_size=1, padding=0, bias=True))
        self.class_head[-1].bias.data.fill_(- np.log(1 / cfg.MODEL.HEAD.INIT_P - 1))

        # --- regression branches: one 3x3 feature conv per head group, one 1x1 conv per key
        if self.deeper_head:
            self.reg_head_pre = self._deep_stem()
        self.reg_features = nn.ModuleList()
        self.reg_heads = nn.ModuleList()
        for idx, keys in enumerate(self.regression_head_cfg):
            self.reg_features.append(nn.Sequential(
                Conv2d(trunk_in, self.head_conv, kernel_size=3, padding=1, bias=False),
                self.norm_func(self.head_conv), self._get_active_func()))
            heads = nn.ModuleList()
            for key_index, key in enumerate(keys):
                out_head = nn.Conv2d(self.head_conv, self.regression_channel_cfg[idx][key_index], kernel_size=1,
                                     padding=0, bias=True)
                if key.find('uncertainty') >= 0 and cfg.MODEL.HEAD.UNCERTAINTY_INIT:
                    torch.nn.init.xavier_normal_(out_head.weight, gain=0.01)
                if key == '3d_offset':   # the edge fusion is applied to this branch
                    self.offset_index = [idx, key_index]
                _fill_fc_weights(out_head, 0)
                heads.append(out_head)
            self.reg_heads.append(heads)

        # --- edge (truncation) feature fusion
        self.enable_edge_fusion = cfg.MODEL.HEAD.ENABLE_EDGE_FUSION
        self.edge_fusion_kernel_size = cfg.MODEL.HEAD.EDGE_FUSION_KERNEL_SIZE
        self.edge_fusion_relu = cfg.MODEL.HEAD.EDGE_FUSION_RELU
        self.exact_edge_gather = True   # False -> F.grid_sample exactly as the reference
        # Training evaluates the regression heads ONLY at the annotated object centres (the loss reads nothing else,
        # detector_loss.py:231-233): the 11 dense 1x1 convolutions, the 415-channel concatenation and -- in backward -- a
        # zero-filled 415-channel gradient map, its 12 slices, 12 dense 1x1 data/weight gradients all disappear.  The result
        # is `reg_pois` (B, MAX_OBJECTS, 415); `reg` is then None.  False restores the reference's dense training output.
        self.sparse_training_heads = True
        # Inference needs the regression heads at the TEST.DETECTIONS_PER_IMG top-scoring cells only (detector_infer.py:101-110).
        # True: the class map is computed densely, NMS + top-K run HERE, and the trunks / output layers are evaluated at those
        # cells (`reg_pois` (B, K, 415) + the top-K result under 'topk'; `reg` is None) -- the PostProcessor takes both as they
        # are.  False (default): the reference's dense (B, 415, H, W) map.
        self.sparse_eval_heads = False
        self.max_detection = cfg.TEST.DETECTIONS_PER_IMG
        if self.enable_edge_fusion:
            norm1d = nn.BatchNorm1d if cfg.MODEL.HEAD.EDGE_FUSION_NORM == 'BN' else nn.Identity
            k = self.edge_fusion_kernel_size

            def edge_branch(out_ch):
                act = nn.ReLU(inplace=True) if self.edge_fusion_relu else nn.Identity()
                return EdgeBranch(
                    nn.Conv1d(self.head_conv, self.head_conv, kernel_size=k, padding=k // 2, padding_mode='replicate'),
                    norm1d(self.head_conv), act, nn.Conv1d(self.head_conv, out_ch, kernel_size=1))
            self.trunc_heatmap_conv = edge_branch(classes)
            self.trunc_offset_conv = edge_branch(2)

    def _get_active_func(self):
        if self.active_func == 'relu' and self.fused_bn_relu:
            return nn.Identity()
        if self.active_func == 'relu':
            return nn.ReLU(inplace=True)
        if self.active_func == 'leaky_relu':
            return nn.LeakyReLU(inplace=True)
        raise ValueError('No such activate func')

    def _deep_stem(self):
        """Optional conv + DCN stem used when MODEL.HEAD.DEEPER_HEAD is set (detector_predictor.py:134-151)."""
        return nn.Sequential(
            Conv2d(self.in_channels, self.head_conv, kernel_size=3, padding=1, bias=False),
            self.norm_func(self.head_conv), self._get_active_func(),
            DCN(self.head_conv, self.head_conv, kernel_size=(3, 3), stride=1, padding=1, dilation=1, deformable_groups=1),
            self.norm_func(self.head_conv), self._get_active_func())

    def _edge_fusion(self, feature_cls, reg_feature, output_cls, output_reg, targets):
        """Sample both feature maps along the image border, run the two Conv1d branches and add the result back
        at the border cells (detector_predictor.py:172-196)."""
        b = feature_cls.shape[0]
        edge_indices = stack_field(targets, "edge_indices")          # B x K x 2 (x, y)
        edge_lens = stack_field(targets, "edge_len").view(b, 1)       # B x 1
        out_w = stack_field(targets, "final_output_w").float().view(-1, 1, 1)
        out_h = stack_field(targets, "final_output_h").float().view(-1, 1, 1)

        K = edge_indices.shape[1]
        bi = torch.arange(b, device=edge_indices.device).view(b, 1, 1)
        yi = edge_indices[:, :, 1].long().view(b, 1, K)
        xi = edge_indices[:, :, 0].long().view(b, 1, K)
        # final_output_w/h are the configured output size for every sample (kitti.py sets them from the same cfg keys);
        # compared as Python ints so that no device->host sync is needed here
        if self.exact_edge_gather and self.output_width == feature_cls.shape[3] and self.output_height == feature_cls.shape[2]:
            # The reference samples with F.grid_sample(align_corners=True) at grid points that ARE integer cells
            # (detector_predictor.py:178-184): a gather.  Reading the cells directly skips the 512-channel full-resolution
            # scatter of grid_sample's backward (2.2 ms + a 0.5 GB zero-fill per step at bs 8); it differs from the
            # reference only by the ~1e-5 interpolation leakage of the fp32 grid normalisation.
            ci = torch.arange(self.head_conv, device=feature_cls.device).view(1, -1, 1)
            edge_cls_feature = feature_cls[bi, ci, yi, xi]
            edge_offset_feature = reg_feature[bi, ci, yi, xi]
        else:
            grid = edge_indices.view(b, -1, 1, 2).float()
            grid = torch.stack((grid[..., 0] / (out_w - 1) * 2 - 1, grid[..., 1] / (out_h - 1) * 2 - 1), dim=-1)
            fused = torch.cat((feature_cls, reg_feature), dim=1)
            edge_features = F.grid_sample(fused, grid.type_as(fused), align_corners=True).squeeze(-1)
            edge_cls_feature = edge_features[:, :self.head_conv, ...]
            edge_offset_feature = edge_features[:, self.head_conv:, ...]
        edge_cls_output = self.trunc_heatmap_conv(edge_cls_feature)
        edge_offset_output = self.trunc_offset_conv(edge_offset_feature)
        valid = (torch.arange(K, device=edge_indices.device).view(1, K) < edge_lens).to(edge_cls_output.dtype)
        for out, vals in ((output_cls, edge_cls_output), (output_reg, edge_offset_output)):
            ci = torch.arange(out.shape[1], device=out.device).view(1, -1, 1)
            out.index_put_((bi, ci, yi, xi), vals * valid.unsqueeze(1), accumulate=True)

    def _edge_fusion_at_pois(self, edge_feature, edge_lin, valid, centers_lin):
        """Edge-fusion term of the 3d_offset head at the object centres: what `_edge_fusion` would have accumulated into the
        dense map at those cells (detector_predictor.py:172-196, duplicates of a border cell included).
        edge_feature (B, 256, K): trunk output at the border cells."""
        edge_offset_output = self.trunc_offset_conv(edge_feature)                            # B x 2 x K
        hit = (centers_lin.unsqueeze(2) == edge_lin.unsqueeze(1)) & valid.unsqueeze(1)       # B x M x K
        return torch.bmm(hit.to(edge_offset_output.dtype), edge_offset_output.transpose(1, 2))  # B x M x 2

    def _forward_sparse(self, features, targets):
        """Training forward with the regression heads evaluated at the object centres only.  The trunks' BN + ReLU is
        evaluated at those positions too (`BatchNorm2d.forward_at`): its dense output map is never written, and its backward
        needs one read of the conv output and one write of the gradient instead of seven tensor passes."""
        # one feature map feeds all twelve trunks: its gradient is summed by one kernel instead of eleven pairwise additions
        n_reg = len(self.reg_features)
        if self.deeper_head:
            feat_cls_in = self.cls_head_pre(features)
            feat_reg_in = self.reg_head_pre(features)
            reg_inputs = ops.fan_out(feat_reg_in, n_reg) if feat_reg_in.is_cuda and feat_reg_in.requires_grad else [feat_reg_in] * n_reg
        elif trunk_moments.ENABLED and trunk_moments.usable(self.reg_features, features):
            feat_cls_in, reg_inputs = features, [features] * n_reg          # two consumers only: the class trunk and the moments
        elif features.is_cuda and features.requires_grad:
            fans = ops.fan_out(features, n_reg + 1)
            feat_cls_in, reg_inputs = fans[0], fans[1:]
        else:
            feat_cls_in, reg_inputs = features, [features] * n_reg
        feature_cls = self.class_head[:-1](feat_cls_in)
        b, _, h, w = feature_cls.shape
        last = self.class_head[-1]
        edge_cls_feature = None
        if (_HEAD_FUSED and self.enable_edge_fusion and feature_cls.is_cuda and feature_cls.dtype == torch.float32 and isinstance(last, nn.Conv2d)
                and last.kernel_size == (1, 1) and last.stride == (1, 1) and last.padding == (0, 0) and last.groups == 1):
            # the class head's output layer and the edge-fusion gather read the same feature map: one node, one gradient write
            ei = stack_field(targets, "edge_indices")
            output_cls, edge_cls_feature = ops.head_out_and_gather(feature_cls, last.weight, last.bias,
                                                                   ei[:, :, 1].long() * w + ei[:, :, 0].long())
        else:
            output_cls = last(feature_cls)
        centers = stack_field(targets, "target_centers")              # B x M x 2 (x, y)
        centers_lin = centers[:, :, 1].long() * w + centers[:, :, 0].long()                  # B x M
        M = centers_lin.shape[1]
        if self.enable_edge_fusion:
            edge_indices = stack_field(targets, "edge_indices")      # B x K x 2 (x, y)
            edge_lens = stack_field(targets, "edge_len").view(b, 1)
            K = edge_indices.shape[1]
            edge_lin = edge_indices[:, :, 1].long() * w + edge_indices[:, :, 0].long()       # B x K
            edge_valid = torch.arange(K, device=edge_lin.device).view(1, K) < edge_lens      # B x K
        outs = []
        # All trunks at once from the Gram matrix of the shared input's 3x3 patches: no dense trunk output exists (trunk_moments.py)
        moments = None
        frozen_bn = (not self.deeper_head and _HEAD_ROWS and features.is_cuda and features.dtype == torch.float32
                     and not torch.is_grad_enabled() and self._head_rows_ok() and trunk_moments.frozen(self.reg_features, reg_inputs[0]))
        if frozen_bn or (not self.deeper_head and all(isinstance(fl[2], nn.Identity) for fl in self.reg_features)
                         and trunk_moments.usable(self.reg_features, reg_inputs[0])):
            extra = (self.offset_index[0], edge_lin) if self.enable_edge_fusion else None
            if frozen_bn or (_HEAD_ROWS and features.is_cuda and features.dtype == torch.float32 and self._head_rows_ok()):
                # every 1x1 output layer at the object centres in ONE launch (ops.head_rows) instead of a GEMM per head, their
                # concatenation and -- backward -- two GEMMs, a bias sum and a slice copy per head
                if frozen_bn:      # --generate_for_GMW pass: BatchNorm frozen, no gradients -> no statistics, no dense trunk output
                    at_centres, at_extra = trunk_moments.trunks_at_frozen(reg_inputs[0], self.reg_features, centers_lin, extra)
                else:
                    at_centres, at_extra = trunk_moments.trunks_at(reg_inputs[0], self.reg_features, centers_lin, extra, stacked=True)
                heads = [(i, h) for i, hs in enumerate(self.reg_heads) for h in hs]
                reg_pois = ops.head_rows(at_centres.reshape(n_reg, b * M, self.head_conv), [i for i, _ in heads],
                                         [h.weight for _, h in heads], [h.bias for _, h in heads]).view(b, M, -1)
                if self.enable_edge_fusion:
                    first = sum(len(hs) for hs in self.reg_heads[:self.offset_index[0]]) + self.offset_index[1]
                    ch0 = sum(h.out_channels for _, h in heads[:first])              # first channel of the 3d_offset head
                    edge = self._edge_fusion_at_pois(at_extra.transpose(1, 2), edge_lin, edge_valid, centers_lin)     # B x M x 2
                    reg_pois = reg_pois + F.pad(edge, (ch0, reg_pois.shape[2] - ch0 - edge.shape[2]))
                    output_cls = self._edge_fusion_cls(feature_cls, output_cls, targets, edge_cls_feature)
                output_cls = sigmoid_hm(output_cls)
                return {'cls': output_cls.float(), 'reg': None, 'reg_pois': reg_pois}
            moments = trunk_moments.trunks_at(reg_inputs[0], self.reg_features, centers_lin, extra)
        for i, feat_layer in enumerate(self.reg_features):
            fused = i == self.offset_index[0] and self.enable_edge_fusion
            pos = torch.cat((centers_lin, edge_lin), dim=1) if fused else centers_lin
            conv, norm = feat_layer[0], feat_layer[1]
            if moments is not None:
                at_all = moments[i]
            elif hasattr(norm, "forward_at") and isinstance(feat_layer[2], nn.Identity):
                at_all = norm.forward_at(conv(reg_inputs[i]), pos)                           # B x N x 256
            else:                                                                             # GN / leaky-relu configurations
                at_all = select_point_of_interest(b, pos, feat_layer(reg_inputs[i]))
            at = at_all[:, :M].reshape(b * M, self.head_conv)
            for j, out_head in enumerate(self.reg_heads[i]):
                o = F.linear(at, out_head.weight.view(out_head.out_channels, self.head_conv), out_head.bias).view(b, M, -1)
                if fused and j == self.offset_index[1]:
                    o = o + self._edge_fusion_at_pois(at_all[:, M:].transpose(1, 2), edge_lin, edge_valid, centers_lin)
                    # the class map gets its edge term densely (it is consumed densely by the focal loss)
                    output_cls = self._edge_fusion_cls(feature_cls, output_cls, targets, edge_cls_feature)
                outs.append(o)
        output_cls = sigmoid_hm(output_cls)
        return {'cls': output_cls.float(), 'reg': None, 'reg_pois': torch.cat(outs, dim=2).float()}

    def _head_rows_ok(self):
        """The shapes ops.head_rows takes: 1x1 output layers over a trunk of at most 256 channels, at most 16 heads."""
        heads = [h for hs in self.reg_heads for h in hs]
        return (len(heads) <= 16 and len(self.reg_features) <= 16 and self.head_conv <= 256 and self.head_conv % 4 == 0
                and all(isinstance(h, nn.Conv2d) and h.kernel_size == (1, 1) and h.groups == 1 and h.in_channels == self.head_conv
                        for h in heads) and sum(h.out_channels for h in heads) <= 1024)

    def _edge_fusion_cls(self, feature_cls, output_cls, targets, gathered=None):
        """The class-map half of `_edge_fusion` (dense: the focal loss reads every cell).  gathered: feature_cls at the border
        cells (B, K, C) when the caller already has it (ops.head_out_and_gather)."""
        b = feature_cls.shape[0]
        edge_indices = stack_field(targets, "edge_indices")
        edge_lens = stack_field(targets, "edge_len").view(b, 1)
        K = edge_indices.shape[1]
        bi = torch.arange(b, device=edge_indices.device).view(b, 1, 1)
        yi = edge_indices[:, :, 1].long().view(b, 1, K)
        xi = edge_indices[:, :, 0].long().view(b, 1, K)
        valid = torch.arange(K, device=edge_indices.device).view(1, K) < edge_lens
        if feature_cls.is_cuda and output_cls.dtype == torch.float32 and output_cls.is_contiguous():
            # gather / scatter-add at linear cell indices on the HIP kernels (one launch each)
            lin = (yi * feature_cls.shape[3] + xi).view(b, K)
            at_edges = gathered if gathered is not None else ops.select_point_of_interest(b, lin, feature_cls)
            edge_cls_output = self.trunc_heatmap_conv(at_edges.transpose(1, 2))
            vals = (edge_cls_output * valid.unsqueeze(1).to(edge_cls_output.dtype)).transpose(1, 2)
            return ops.scatter_add_at(output_cls, vals, lin)
        ci = torch.arange(self.head_conv, device=feature_cls.device).view(1, -1, 1)
        edge_cls_output = self.trunc_heatmap_conv(feature_cls[bi, ci, yi, xi])
        valid = valid.to(edge_cls_output.dtype)
        co = torch.arange(output_cls.shape[1], device=output_cls.device).view(1, -1, 1)
        output_cls.index_put_((bi, co, yi, xi), edge_cls_output * valid.unsqueeze(1), accumulate=True)
        return output_cls

    def _forward_sparse_eval(self, features, targets):
        """Inference forward with the regression heads at the top-K cells only (see `sparse_eval_heads`)."""
        from dcd_amd.model.layers.utils import select_topk
        feature_cls = self.class_head[:-1](features)
        output_cls = self.class_head[-1](feature_cls)
        b, _, h, w = feature_cls.shape
        if self.enable_edge_fusion:
            output_cls = self._edge_fusion_cls(feature_cls, output_cls, targets)
        heat = sigmoid_hm(output_cls).float()
        topk = select_topk(heat, K=self.max_detection, fuse_nms=True)                 # scores, cell index, classes, ys, xs: (B, K) each
        centers_lin = topk[1].long().view(b, -1)
        M = centers_lin.shape[1]
        extra = None
        if self.enable_edge_fusion:
            edge_indices = stack_field(targets, "edge_indices")
            edge_lens = stack_field(targets, "edge_len").view(b, 1)
            K = edge_indices.shape[1]
            edge_lin = edge_indices[:, :, 1].long() * w + edge_indices[:, :, 0].long()
            edge_valid = torch.arange(K, device=edge_lin.device).view(1, K) < edge_lens
            extra = (self.offset_index[0], edge_lin)
        at_centres, at_extra = trunk_moments.trunks_at_frozen(features, self.reg_features, centers_lin, extra)
        heads = [(i, hd) for i, hs in enumerate(self.reg_heads) for hd in hs]
        reg_pois = ops.head_rows(at_centres.reshape(len(self.reg_features), b * M, self.head_conv), [i for i, _ in heads],
                                 [hd.weight for _, hd in heads], [hd.bias for _, hd in heads]).view(b, M, -1)
        if self.enable_edge_fusion:
            first = sum(len(hs) for hs in self.reg_heads[:self.offset_index[0]]) + self.offset_index[1]
            ch0 = sum(hd.out_channels for _, hd in heads[:first])
            edge = self._edge_fusion_at_pois(at_extra.transpose(1, 2), edge_lin, edge_valid, centers_lin)
            reg_pois = reg_pois + F.pad(edge, (ch0, reg_pois.shape[2] - ch0 - edge.shape[2]))
        return {'cls': heat, 'reg': None, 'reg_pois': reg_pois.float(), 'topk': topk}

    def forward(self, features, targets):
        if (self.training and self.sparse_training_heads and targets is not None and self.exact_edge_gather
                and self.output_width == features.shape[3] and self.output_height == features.shape[2]):
            return self._forward_sparse(features, targets)
        if (not self.training and self.sparse_eval_heads and not self.deeper_head and features.is_cuda and features.dtype == torch.float32
                and not torch.is_grad_enabled() and self._head_rows_ok() and trunk_moments.frozen(self.reg_features, features)
                and self.output_width == features.shape[3] and self.output_height == features.shape[2]):
            return self._forward_sparse_eval(features, targets)
        feat_cls_in = self.cls_head_pre(features) if self.deeper_head else features
        feature_cls = self.class_head[:-1](feat_cls_in)
        output_cls = self.class_head[-1](feature_cls)

        feat_reg_in = self.reg_head_pre(features) if self.deeper_head else features
        output_regs = []
        for i, feat_layer in enumerate(self.reg_features):
            reg_feature = feat_layer(feat_reg_in)
            for j, out_head in enumerate(self.reg_heads[i]):
                output_reg = out_head(reg_feature)
                if self.enable_edge_fusion and i == self.offset_index[0] and j == self.offset_index[1]:
                    self._edge_fusion(feature_cls, reg_feature, output_cls, output_reg, targets)
                output_regs.append(output_reg)

        output_cls = sigmoid_hm(output_cls)
        return {'cls': output_cls.float(), 'reg': torch.cat(output_regs, dim=1).float()}


def make_predictor(cfg, in_channels):
    return registry.PREDICTOR[cfg.MODEL.HEAD.PREDICTOR](cfg, in_channels)
