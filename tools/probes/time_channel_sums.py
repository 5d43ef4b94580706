"""ops.channel_sums (bias gradient of DCN's offset convolutions) per shape; DCD_CHANNEL_SUM_ONE_LAUNCH=1|0 pins the form."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dcd_amd import ops
dev = torch.device("cuda:0")
for B in (8, 1):
    for C, H, W in ((27, 96, 320), (27, 48, 160), (27, 24, 80), (27, 12, 40), (256, 96, 320)):
        x = torch.randn(B, C, H, W, device=dev)
        for _ in range(3):
            r = ops.channel_sums(x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20):
            r = ops.channel_sums(x)
        e1.record(); torch.cuda.synchronize()
        err = (r.double() - x.double().sum((0, 2, 3))).abs().max().item()
        print("B %d C %3d @%3dx%3d  %.1f us  err %.2e" % (B, C, H, W, e0.elapsed_time(e1) / 20 * 1e3, err))
