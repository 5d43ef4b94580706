"""Backward time of one DCN layer through the C ABI: python tools/time_bwd_one.py C Co H W [B] [off_scale] [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dcd_amd import _ext
C, Co, H, W = [int(v) for v in sys.argv[1:5]]
B = int(sys.argv[5]) if len(sys.argv) > 5 else 8
osc = float(sys.argv[6]) if len(sys.argv) > 6 else 0.5
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 10
dev = torch.device("cuda:0")
x = torch.randn(B, C, H, W, device=dev)
off = torch.randn(B, 18, H, W, device=dev) * osc
m = torch.sigmoid(torch.randn(B, 9, H, W, device=dev))
w = torch.randn(Co, C, 3, 3, device=dev) / (C * 9) ** 0.5
b = torch.zeros(Co, device=dev)
gy = torch.randn(B, Co, H, W, device=dev)
a = (3, 3, 1, 1, 1, 1, 1, 1, 1)
for _ in range(3):
    _ext.dcn_v2_backward(x, w, b, off, m, gy, *a, precision=os.environ.get("DCD_PREC", "f32"))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(iters):
    _ext.dcn_v2_backward(x, w, b, off, m, gy, *a, precision=os.environ.get("DCD_PREC", "f32"))
e1.record()
torch.cuda.synchronize()
print("%d->%d @%dx%d B=%d off=%.2g  bwd %.3f ms" % (C, Co, H, W, B, osc, e0.elapsed_time(e1) / iters))
