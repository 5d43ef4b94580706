"""Forward time of the Winograd 3x3 convolution, exact fp32 vs split-bf16 products (same shapes as tools/wino_layers.py)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from dcd_amd import _ext, ops
dev = torch.device("cuda:0")


def t(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for C, K, H, W in [(64, 64, 96, 320), (64, 256, 96, 320), (128, 128, 48, 160), (256, 256, 24, 80), (512, 512, 12, 40), (64, 27, 96, 320),
                   (128, 27, 48, 160), (256, 27, 24, 80)]:
    B = int(os.environ.get("B", "8"))
    x = torch.randn(B, C, H, W, device=dev)
    w = torch.randn(K, C, 3, 3, device=dev) * 0.05
    out = []
    for prec in ("f32", "bf16x3"):
        _ext.set_precision(prec)
        tf, tb = ops.conv3x3_transform_weights(w)
        out.append(t(lambda: ops._conv3x3_call(x, w, K, False, transformed=tf)))
    _ext.set_precision("f32")
    fl = 2.0 * B * H * W * C * K * 9
    print("%4d->%3d @%3dx%3d  f32 %6.1f us   split %6.1f us   (%.2fx; split = %.0f TF/s of direct-conv flops)" % (
        C, K, H, W, out[0], out[1], out[0] / out[1], fl / out[1] / 1e6))
