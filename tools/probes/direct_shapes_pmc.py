"""The direct bf16 convolution kernels alone (forward + weight gradient, three DGDE shapes at bs 8) for the counter passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dcd_amd import ops, _ext
dev = torch.device("cuda:0")
for C, K, H, W in [(64, 64, 96, 320), (64, 256, 96, 320), (128, 128, 48, 160)]:
    x = torch.randn(8, C, H, W, device=dev)
    gy = torch.randn(8, K, H, W, device=dev)
    w = torch.randn(K, C, 3, 3, device=dev) / (C * 9) ** 0.5
    with _ext.precision_scope("bf16"):
        tf, tb = ops.conv3x3_transform_weights(w)
    for _ in range(6):
        y = ops._conv3x3_call(x, w, K, False, transformed=tf)
        gw = ops._conv3x3_wrw_call(x, gy, w.shape, ops.PREC_BF16)
    torch.cuda.synchronize()
