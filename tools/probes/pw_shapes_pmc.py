"""The pointwise bf16 kernels alone (conv1x1_of_cat forward + backward inside the bf16 scope, three Root shapes at bs 8) for the
counter passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dcd_amd import ops, _ext
dev = torch.device("cuda:0")
for cs, O, H, W in [((64, 64), 64, 96, 320), ((128, 128, 64, 128), 128, 48, 160), ((256, 256), 256, 24, 80)]:
    xs = [torch.randn(8, c, H, W, device=dev, requires_grad=True) for c in cs]
    C = sum(cs)
    w = (torch.randn(O, C, 1, 1, device=dev) / C ** 0.5).requires_grad_(True)
    g = torch.randn(8, O, H, W, device=dev)
    for _ in range(6):
        with _ext.precision_scope("bf16"):
            y = ops.conv1x1_of_cat(xs, w)
        torch.autograd.grad(y, xs + [w], g)
    torch.cuda.synchronize()
