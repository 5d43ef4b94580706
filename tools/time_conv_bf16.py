"""One-product bf16 3x3 convolution (DCD_PREC_BF16), forward and backward-data with PREPARED weights: kernel time only.
DCD_CONV_BF16_DIRECT=0 times the Winograd form of the same calls (round 5: direct implicit GEMM vs Winograd on the bf16 pipe)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dcd_amd import ops, _ext

SHAPES = [(16, 16, 384, 1280), (64, 256, 96, 320), (256, 64, 96, 320), (64, 64, 96, 320), (128, 128, 48, 160), (256, 256, 24, 80),
          (512, 512, 12, 40), (64, 64, 192, 640), (128, 128, 96, 320)]


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


dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for C, K, H, W in SHAPES:
    x = torch.randn(B, C, H, W, device=dev)
    w = torch.randn(K, C, 3, 3, device=dev) / (C * 9) ** 0.5
    gy = torch.randn(B, K, H, W, device=dev)
    with _ext.precision_scope("bf16"):
        tf, tb = ops.conv3x3_transform_weights(w)
    fl = 2.0 * B * K * C * 9 * H * W
    by = 4.0 * B * (K + C) * H * W
    a = t(lambda: ops._conv3x3_call(x, w, K, False, transformed=tf))
    c = t(lambda: ops._conv3x3_call(gy, w, C, True, transformed=tb))
    print("%4d->%3d @%3dx%4d  fwd %7.1f us (%4.0f TF, %4.2f TB/s)   bwd-data %7.1f us (%4.0f TF, %4.2f TB/s)" % (
        C, K, H, W, a, fl / a / 1e6, by / a / 1e6, c, fl / c / 1e6, by / c / 1e6))
