"""Stock-op cost of DLA-34's low-channel full-resolution layers (base_layer 7x7 3->16, level0 3x3 16->16, level1 3x3 16->32 s2,
DGDE/model/backbone/dla_dcn.py:236-246) at bs 8, 384x1280: forward, input gradient, weight gradient, and their BN."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.nn import functional as F


def t(fn, iters=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for name, Ci, Co, k, s, p in [("base 7x7 3->16", 3, 16, 7, 1, 3), ("level0 3x3 16->16", 16, 16, 3, 1, 1), ("level1 3x3 16->32 s2", 16, 32, 3, 2, 1)]:
    x = torch.randn(B, Ci, 384, 1280, device=dev)
    w = torch.randn(Co, Ci, k, k, device=dev) * 0.05
    y = F.conv2d(x, w, stride=s, padding=p)
    gy = torch.randn_like(y)
    a = t(lambda: F.conv2d(x, w, stride=s, padding=p))
    args = (gy, x, w, None, [s, s], [p, p], [1, 1], False, [0, 0], 1)
    b = t(lambda: torch.ops.aten.convolution_backward(*args, [True, False, False]))
    c = t(lambda: torch.ops.aten.convolution_backward(*args, [False, True, False]))
    mb = (x.numel() + y.numel()) * 4 / 1e6
    if s == 1:
        from dcd_amd import ops
        a2 = t(lambda: ops._conv_stem_call(x, w, False))
        b2 = t(lambda: ops._conv_stem_call(gy, w, True)) if Ci == 16 else float("nan")
        xg, wg = x.clone().requires_grad_(False), w.clone().requires_grad_()
        yy = ops.conv_stem(xg, wg)
        c2 = t(lambda: torch.autograd.grad(yy, wg, gy, retain_graph=True))
        print("%-22s hip: fwd %.3f ms  bwd-data %.3f ms  wrw %.3f ms" % (name, a2, b2, c2))
    print("%-22s fwd %.3f ms  bwd-data %.3f ms  wrw %.3f ms   (x+y = %.0f MB -> %.3f ms at 4 TB/s)" % (name, a, b, c, mb, mb / 4e3))
