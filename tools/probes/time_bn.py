"""HBM rate of the BatchNorm(+ReLU) kernels on DLA-34's tensor sizes (bs 8): forward = read x (stats) + read x, write y; backward =
read gy, y, x (sums) + read gy, y, x, write gx."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from dcd_amd import ops
dev = torch.device("cuda:0")


def t(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for C, H, W in [(16, 384, 1280), (32, 192, 640), (64, 96, 320), (128, 48, 160), (256, 24, 80), (512, 12, 40)]:
    x = torch.randn(8, C, H, W, device=dev, requires_grad=True)
    w = torch.ones(C, device=dev, requires_grad=True)
    b = torch.zeros(C, device=dev, requires_grad=True)
    rm, rv, nb = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros((), dtype=torch.long, device=dev)
    y = ops.batch_norm_act(x, None, w, b, rm, rv, nb, 0.1, 1e-5, True)
    gy = torch.randn_like(y)
    nbytes = x.numel() * 4
    f = t(lambda: ops.batch_norm_act(x, None, w, b, rm, rv, nb, 0.1, 1e-5, True))
    bw = t(lambda: torch.autograd.grad(y, (x, w, b), gy, retain_graph=True))
    print("%3d ch @%4dx%4d  %6.1f MB  fwd %6.3f ms (%4.2f TB/s for 3 passes)   bwd %6.3f ms (%4.2f TB/s for 7 passes)" % (
        C, H, W, nbytes / 1e6, f, 3 * nbytes / f / 1e9, bw, 7 * nbytes / bw / 1e9))
