"""Repeated dcn_v2_backward calls on identical inputs: max |out_k - out_0| / max |out_0| per output, per precision."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from dcd_amd import _ext

dev = torch.device("cuda:0")
torch.manual_seed(0)
for (C, Co, H, W) in ((256, 64, 24, 80), (256, 256, 24, 80), (512, 256, 12, 40), (256, 128, 24, 80)):
    B = 8
    x = torch.randn(B, C, H, W, device=dev)
    off = torch.randn(B, 18, H, W, device=dev) * float(os.environ.get('OFF', '0.5'))
    far = torch.rand(B, 18, H, W, device=dev) < 0.004
    off = torch.where(far, off * 20, off)
    m = torch.sigmoid(torch.randn(B, 9, H, W, device=dev))
    w = torch.randn(Co, C, 3, 3, device=dev) / (C * 9) ** 0.5
    b = torch.zeros(Co, device=dev)
    gy = torch.randn(B, Co, H, W, device=dev)
    a = (3, 3, 1, 1, 1, 1, 1, 1, 1)
    for prec in ("f32", "bf16x3", "bf16"):
        outs = []
        for k in range(6):
            outs.append([t.clone() for t in _ext.dcn_v2_backward(x, w, b, off, m, gy, *a, precision=prec)])
            torch.cuda.synchronize()
        line = []
        for k in range(1, 6):
            line.append(" ".join("%.1e" % ((outs[k][i] - outs[0][i]).abs().max().item() / max(outs[0][i].abs().max().item(), 1e-12)) for i in range(5)))
        print("%d->%d@%dx%d %-6s [gi goff gmask gw gb] vs call 0: %s" % (C, Co, H, W, prec, " | ".join(line)), flush=True)
