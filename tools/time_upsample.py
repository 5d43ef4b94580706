"""Depthwise ConvTranspose2d (IDAUp.up_*: kernel 2f, stride f, padding f/2, groups = channels) forward + backward: the stock op,
or with `ours` as first argument csrc/upsample.hip (DCD_UP_BWD_OLD=1: its first backward kernel)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch import nn
SHAPES = [(256, 12, 40, 2), (128, 24, 80, 2), (128, 24, 80, 2), (64, 48, 160, 2), (64, 48, 160, 2), (64, 48, 160, 2), (64, 48, 160, 2), (64, 24, 80, 4)]
dev = torch.device("cuda:0")
tot = 0
for C, H, W, f in SHAPES:
    if len(sys.argv) > 1 and sys.argv[1] == "ours":
        from dcd_amd.model.layers.conv import DepthwiseUpsample
        up = DepthwiseUpsample(C, C, f * 2, stride=f, padding=f // 2, groups=C, bias=False).to(dev)
    else:
        up = nn.ConvTranspose2d(C, C, f * 2, stride=f, padding=f // 2, groups=C, bias=False).to(dev)
    x = torch.randn(8, C, H, W, device=dev, requires_grad=True)
    def step():
        y = up(x)
        y.backward(torch.ones_like(y))
    for _ in range(3):
        step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        step()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    tot += t
    print("C=%3d %3dx%3d f=%d  fwd+bwd %.3f ms" % (C, H, W, f, t))
print("total %.3f ms" % tot)
