"""One-product (DCD_PREC_BF16) Winograd convolution of csrc/conv.hip at DGDE's big shapes, against MIOpen's bf16 solvers on
bf16 tensors (NCHW and channels_last): what a direct bf16 implicit GEMM achieves on this part.  DCD_PROBE_ONLY=1: our kernel only
(for the PMC passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.nn import functional as F
from dcd_amd import _ext, ops

dev = torch.device("cuda:0")


def t(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


only = os.environ.get("DCD_PROBE_ONLY") == "1"
for C, K, H, W in [(64, 64, 96, 320), (128, 128, 48, 160), (256, 256, 24, 80), (512, 512, 12, 40)]:
    B = 8
    x = torch.randn(B, C, H, W, device=dev)
    w = torch.randn(K, C, 3, 3, device=dev) / (C * 9) ** 0.5
    gy = torch.randn(B, K, H, W, device=dev)
    with _ext.precision_scope("bf16"):
        tf, tb = ops.conv3x3_transform_weights(w)
        a = t(lambda: ops._conv3x3_call(x, w, K, False, transformed=tf))
    e = t(lambda: ops._conv3x3_wrw_call(x, gy, w.shape, ops.PREC_BF16))
    line = "%4d->%3d @%3dx%3d  ours fwd %.1f us  wrw %.1f us" % (C, K, H, W, a, e)
    if not only:
        xb, wb = x.bfloat16(), w.bfloat16()
        b = t(lambda: F.conv2d(xb, wb, padding=1))
        xc, wc = xb.contiguous(memory_format=torch.channels_last), wb.contiguous(memory_format=torch.channels_last)
        c = t(lambda: F.conv2d(xc, wc, padding=1))
        gb = gy.bfloat16()
        d = t(lambda: torch.ops.aten.convolution_backward(gb, xb, wb, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False]))
        line += " | miopen bf16 nchw %.1f  nhwc %.1f  wrw nchw %.1f" % (b, c, d)
    print(line, flush=True)
