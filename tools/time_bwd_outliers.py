"""64->64 @ 96x320, bs 8: smooth 0.5-px offsets plus one 16x16 patch per image that jumps 5 px (call-wide radius > the tiled
grad_input kernel's window).  Backward ms with the per-tile dispatch (DCD_BI_HYBRID=1, default) vs the per-call one (=0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dcd_amd import _ext
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, C, H, W = 8, 64, 96, 320
x = torch.randn(B, C, H, W, device=dev)
off = torch.randn(B, 18, H, W, device=dev) * 0.5
off[:, :, 40:56, 100:116] += 5.0
m = torch.sigmoid(torch.randn(B, 9, H, W, device=dev))
w = torch.randn(C, C, 3, 3, device=dev) / (C * 9) ** 0.5
b = torch.zeros(C, device=dev)
gy = torch.randn(B, C, H, W, device=dev)
a = (3, 3, 1, 1, 1, 1, 1, 1, 1)
for _ in range(3):
    _ext.dcn_v2_backward(x, w, b, off, m, gy, *a)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    _ext.dcn_v2_backward(x, w, b, off, m, gy, *a)
e1.record()
torch.cuda.synchronize()
print("DCD_BI_HYBRID=%s: backward %.3f ms" % (os.environ.get("DCD_BI_HYBRID", "1"), e0.elapsed_time(e1) / 10))
