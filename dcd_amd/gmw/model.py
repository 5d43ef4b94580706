"""GMW model (GMW/model/model.py:103-207) and its point-feature extractor (GMW/model/yi2018cvpr/model.py, ops.py).

Same module tree, parameter names and construction order as the reference (so `torch.manual_seed(s); GMW()` gives the same
weights and checkpoints are interchangeable): two extractors (4-D edges of the 2-D keypoints, 6-D edges of the 3-D ones),
each `conv_in` + 12 residual blocks of 1x1 Conv1d with context normalisation; L2-normalised features -> pairwise distance
matrix M (2628 x 2628 per object) -> `1 / diag(M)` as the regression weights and the Sinkhorn transport plan P.

What differs from the reference is execution only: `edge_expand` gathers the 2628 upper-triangle pairs through a
precomputed index (the reference expands to (B,73,73,C), transposes and `masked_select`s twice, model.py:128-152).
The regression weights are `1 / diag(M)` of the same M that feeds Sinkhorn (the transport layer needs every entry).
"""
import os

import torch
from torch import nn
from torch.nn import functional as F

from .optimal_transport import RegularisedTransport


def pairwise_l2_dist(x1, x2):
    """res[..., i, j] = ||x1[..., i, :] - x2[..., j, :]||  (GMW/model/model.py:17-36: ||a||^2 + ||b||^2 - 2 a.b, clamped)."""
    x1_norm2 = x1.pow(2).sum(dim=-1, keepdim=True)
    x2_norm2 = x2.pow(2).sum(dim=-1, keepdim=True)
    res = torch.baddbmm(x2_norm2.transpose(-2, -1), x1, x2.transpose(-2, -1), alpha=-2).add_(x1_norm2)
    return res.clamp_min_(1e-30).sqrt_()


class ContextNorm(nn.Module):
    """`gcn` of yi2018cvpr/ops.py:5-17: per-sample, per-channel normalisation over the points (unbiased variance, eps 1e-3)."""

    def forward(self, x):                                   # x: (B, C, K)
        if x.is_cuda and x.dtype == torch.float32 and x.shape[2] >= 2:
            from dcd_amd import ops
            return ops.context_norm(x, 1e-3)                 # csrc/heads.hip: one launch instead of seven (+ fifteen backward)
        m = torch.mean(x, 2, keepdim=True)
        v = torch.var(x, 2, keepdim=True)
        return (x - m) * (1.0 / torch.sqrt(v + 1e-3))


class PointwiseConv1d(nn.Conv1d):
    """`nn.Conv1d(cin, cout, 1)` (same parameters / state-dict keys); device tensors take the GEMM it is -- W (cout x cin)
    times x (B, cin, K) + bias through rocBLAS -- instead of the convolution library's 1x1 path: its data / weight / bias
    gradient calls cost 3 ms more per step than the two batched GEMMs autograd derives from this one (the forward alone is
    1 ms slower: 4.1 vs 3.0 ms for the two extractors at 8 objects; tools/time_gmw.py, DCD_GMW_CONV_GEMM=0 to compare)."""

    def forward(self, x):
        if x.is_cuda and _POINTWISE_AS_GEMM:
            return torch.baddbmm(self.bias.view(1, -1, 1), self.weight.squeeze(-1).unsqueeze(0).expand(x.shape[0], -1, -1), x)
        return super().forward(x)


_POINTWISE_AS_GEMM = __import__("os").environ.get("DCD_GMW_CONV_GEMM", "1") != "0"


def _conv1d_layer(cin, cout, context_norm):
    layers = [PointwiseConv1d(cin, cout, 1)]
    if context_norm:
        layers.append(ContextNorm())
    return nn.Sequential(*layers)


class ResBlock(nn.Module):
    """conv1d_resnet_block (ops.py:67-131) as configured by Net: preconv, conv1 + gcn, conv2 + gcn, ReLU, + input.
    (`perform_bn` is hard-wired to False inside the block, ops.py:87-116, whatever `net_batchnorm` says.)"""

    def __init__(self, channels, context_norm=True):
        super().__init__()
        self.preconv = _conv1d_layer(channels, channels, False)
        self.conv1 = _conv1d_layer(channels, channels, context_norm)
        self.conv2 = _conv1d_layer(channels, channels, context_norm)

    def forward(self, x):
        return F.relu(self.conv2(self.conv1(self.preconv(x)))) + x


class FeatureExtractor(nn.Module):
    """yi2018cvpr `Net` with its defaults (config.py:70-84): depth 12, 128 channels, context norm on."""

    def __init__(self, in_channel, depth=12, channels=128):
        super().__init__()
        self.numlayer = depth
        self.conv_in = _conv1d_layer(in_channel, channels, False)
        for i in range(depth):
            setattr(self, "conv_%d" % i, ResBlock(channels))

    def forward(self, x):                                   # (B, C_in, K) -> (B, 128, K)
        x = self.conv_in(x)
        for i in range(self.numlayer):
            x = getattr(self, "conv_%d" % i)(x)
        return x


class GMW(nn.Module):
    def __init__(self, num_kpts=73, sinkhorn_lambda=10.0, sinkhorn_tolerance=1e-9):
        super().__init__()
        self.FeatureExtractor4d = FeatureExtractor(4)
        self.FeatureExtractor6d = FeatureExtractor(6)
        self.sinkhorn = RegularisedTransport(sinkhorn_lambda, sinkhorn_tolerance)
        self.num_kpts = num_kpts
        iu = torch.triu_indices(num_kpts, num_kpts, offset=1)       # row-major upper triangle = masked_select order
        self.register_buffer("pair_i", iu[0], persistent=False)
        self.register_buffer("pair_j", iu[1], persistent=False)

    def edge_expand(self, f):
        """(B, n, c) -> (B, n(n-1)/2, 2c): [f_i, f_j] for every pair i < j  (model.py:139-152)."""
        return torch.cat((f.index_select(1, self.pair_i), f.index_select(1, self.pair_j)), dim=-1)

    def _extract(self, which, x):
        """FeatureExtractor4d / 6d; on the GPU in a single-process training run each one is replayed from HIP graphs (forward and
        backward captured once per input shape by torch.cuda.make_graphed_callables): 37 Conv1d + 24 context norms and their
        ~250 backward kernels are launch-bound, and since the transport layer's backward moved to csrc/spd.hip the step waits
        for the host.  DCD_GMW_GRAPH=0 switches it off; under torch.distributed the plain modules run."""
        mod = getattr(self, which)
        if not (x.is_cuda and self.training and torch.is_grad_enabled() and os.environ.get("DCD_GMW_GRAPH", "1") != "0"
                and not (torch.distributed.is_available() and torch.distributed.is_initialized())
                and not torch.cuda.is_current_stream_capturing()):
            return mod(x)
        key = (which, tuple(x.shape), x.dtype, x.requires_grad)
        cache = self.__dict__.setdefault("_graphed", {})
        if key not in cache:
            if len(cache) >= 4:
                cache.pop(next(iter(cache)))
            sample = x.detach().clone().requires_grad_(x.requires_grad)
            cache[key] = torch.cuda.make_graphed_callables(mod, (sample,))
        return cache[key](x.contiguous())

    def graph_matching(self, f4d, f6d):
        f4d = self._extract("FeatureExtractor4d", f4d.transpose(-2, -1)).transpose(-2, -1)         # B x m x 128
        f6d = self._extract("FeatureExtractor6d", f6d.transpose(-2, -1)).transpose(-2, -1)
        f4d = F.normalize(f4d, p=2, dim=-1)
        f6d = F.normalize(f6d, p=2, dim=-1)
        M = pairwise_l2_dist(f4d, f6d)
        diag_feat = 1.0 / M.diagonal(offset=0, dim1=-2, dim2=-1)                         # graph_extract, model.py:154-157
        b, m, n = M.size()
        r = M.new_ones((b, m)) / m
        c = M.new_ones((b, n)) / n
        return self.sinkhorn(M, r, c), diag_feat

    def forward(self, kpts_2d, kpts_3d, pred_rot=None, args=None):
        """(B,73,2) K-normalised keypoints, (B,73,3) object-frame keypoints -> (reg_weights (B,2628), edge_P (B,2628,2628))."""
        edge_P, reg_weights = self.graph_matching(self.edge_expand(kpts_2d), self.edge_expand(kpts_3d))
        return reg_weights, edge_P
