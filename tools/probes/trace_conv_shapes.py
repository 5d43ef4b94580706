"""Kernel-trace durations per shape: runs each shape's prepared forward 12 times with a marker kernel (a tiny fill of n elements,
n = shape index + 1) between shapes; prints per shape the median duration of the convolution kernels.  Run under
rocprofv3 --kernel-trace and parse the csv with tools/probes/trace_conv_parse.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dcd_amd import ops, _ext
dev = torch.device("cuda:0")
B = int(sys.argv[1])
SH = [(16, 16, 384, 1280), (64, 256, 96, 320), (256, 64, 96, 320), (64, 64, 96, 320), (128, 128, 48, 160), (256, 256, 24, 80), (512, 512, 12, 40),
      (64, 27, 96, 320), (27, 64, 96, 320)]
for C, K, H, W in SH:
    x = torch.randn(B, C, H, W, device=dev)
    w = torch.randn(K, C, 3, 3, device=dev) / (C * 9) ** 0.5
    with _ext.precision_scope("bf16"):
        tf, tb = ops.conv3x3_transform_weights(w)
    torch.cuda.synchronize()
    for _ in range(12):
        y = ops._conv3x3_call(x, w, K, False, transformed=tf)
    torch.cuda.synchronize()
