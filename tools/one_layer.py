"""One DCN layer, a few forward+backward calls (for rocprofv3 --kernel-trace): python tools/one_layer.py C Co H W [B] [off_scale]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dcd_amd import _ext
C, Co, H, W = [int(v) for v in sys.argv[1:5]]
B = int(sys.argv[5]) if len(sys.argv) > 5 else 8
osc = float(sys.argv[6]) if len(sys.argv) > 6 else 0.5
dev = torch.device("cuda:0")
x = torch.randn(B, C, H, W, device=dev)
off = torch.randn(B, 18, H, W, device=dev) * osc
m = torch.sigmoid(torch.randn(B, 9, H, W, device=dev))
w = torch.randn(Co, C, 3, 3, device=dev) / (C * 9) ** 0.5
b = torch.zeros(Co, device=dev)
gy = torch.randn(B, Co, H, W, device=dev)
a = (3, 3, 1, 1, 1, 1, 1, 1, 1)
prec = os.environ.get("DCD_PREC", "f32")
for _ in range(4):
    _ext.dcn_v2_forward(x, w, b, off, m, *a, precision=prec)
    _ext.dcn_v2_backward(x, w, b, off, m, gy, *a, precision=prec)
torch.cuda.synchronize()
