"""Runs every distinct DCN layer shape of DLA-34-DCN once (forward + backward) at batch 8 -- the PMC / trace target."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dcd_amd import _ext
LAYERS = [(512, 256, 12, 40, 1), (256, 256, 24, 80, 1), (256, 128, 24, 80, 2), (128, 128, 48, 160, 2),
          (128, 64, 48, 160, 4), (64, 64, 96, 320, 5), (256, 64, 24, 80, 1)]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
off_scale = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
dev = torch.device("cuda:0")
torch.manual_seed(0)
a = (3, 3, 1, 1, 1, 1, 1, 1, 1)
REPS = 2                      # identical passes; summaries divide by REPS
for rep in range(REPS):
    for (C, Co, H, W, mult) in [l for l in LAYERS for _ in range(l[4])]:   # every one of the 16 layers
        x = torch.randn(B, C, H, W, device=dev)
        off = torch.randn(B, 18, H, W, device=dev) * off_scale
        m = torch.sigmoid(torch.randn(B, 9, H, W, device=dev))
        w = torch.randn(Co, C, 3, 3, device=dev) / (C * 9) ** 0.5
        b = torch.zeros(Co, device=dev)
        gy = torch.randn(B, Co, H, W, device=dev)
        torch.cuda.synchronize()
        _ext.dcn_v2_forward(x, w, b, off, m, *a)
        _ext.dcn_v2_backward(x, w, b, off, m, gy, *a)
        torch.cuda.synchronize()
print("done")
