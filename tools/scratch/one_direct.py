import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dcd_amd import ops, _ext
dev = torch.device("cuda:0")
C, K, H, W, B = [int(v) for v in sys.argv[1:6]]
x = torch.randn(B, C, H, W, device=dev)
w = torch.randn(K, C, 3, 3, device=dev) / (C * 9) ** 0.5
with _ext.precision_scope("bf16"):
    tf, tb = ops.conv3x3_transform_weights(w)
for _ in range(10):
    y = ops._conv3x3_call(x, w, K, False, transformed=tf)
torch.cuda.synchronize()
