"""Winograd MFMA conv (csrc/conv.hip) vs the stock MIOpen solver, forward and input gradient, DGDE's 3x3 shapes at bs 8."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.nn import functional as F
from dcd_amd import ops

SHAPES = [(64, 256, 96, 320), (256, 64, 96, 320), (64, 64, 96, 320), (128, 128, 48, 160), (256, 256, 24, 80), (512, 512, 12, 40)]
if os.environ.get("DCD_TIME_OFFSET_CONVS"):          # the DCN layers' conv_offset_mask shapes (27 outputs)
    SHAPES = [(64, 27, 96, 320), (128, 27, 48, 160), (256, 27, 24, 80), (512, 27, 12, 40)]


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
if os.environ.get("DCD_PRECISION"):                       # bf16x3 / bf16: the same table with the products in that precision
    from dcd_amd import _ext
    _ext.set_precision(os.environ["DCD_PRECISION"])
PREC = {"f32": 0, "bf16x3": 1, "bf16": 2}[os.environ.get("DCD_PRECISION", "f32")]
for C, K, H, W in SHAPES:
    x = torch.randn(B, C, H, W, device=dev)
    w = torch.randn(K, C, 3, 3, device=dev) / (C * 9) ** 0.5
    gy = torch.randn(B, K, H, W, device=dev)
    fl = 2.0 * B * K * C * 9 * H * W
    a = t(lambda: ops._conv3x3_call(x, w, K, False))
    b = t(lambda: F.conv2d(x, w, padding=1))
    c = t(lambda: ops._conv3x3_call(gy, w, C, True))
    d = t(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False]))
    e = t(lambda: ops._conv3x3_wrw_call(x, gy, w.shape, PREC))
    f = t(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False]))
    print("%4d->%3d @%3dx%3d  wrw hip %.3f ms (%.0f TF eff) | miopen %.3f (%.0f)" % (C, K, H, W, e, fl / e / 1e9, f, fl / f / 1e9))
    print("%4d->%3d @%3dx%3d  fwd hip %.3f ms (%.0f TF eff) | miopen %.3f (%.0f)   bwd-data hip %.3f (%.0f) | miopen %.3f (%.0f)" % (
        C, K, H, W, a, fl / a / 1e9, b, fl / b / 1e9, c, fl / c / 1e9, d, fl / d / 1e9))
