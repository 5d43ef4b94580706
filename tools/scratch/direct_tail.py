import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dcd_amd import ops, _ext
dev = torch.device("cuda:0")
def t(fn, iters=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for C, K, H, W in [(64, 64, 96, 320), (64, 256, 96, 320), (128, 128, 48, 160)]:
    for B in (2, 4, 6, 8, 10, 12, 16, 24):
        x = torch.randn(B, C, H, W, device=dev)
        w = torch.randn(K, C, 3, 3, device=dev) / (C * 9) ** 0.5
        with _ext.precision_scope("bf16"):
            tf, tb = ops.conv3x3_transform_weights(w)
        a = t(lambda: ops._conv3x3_call(x, w, K, False, transformed=tf))
        wgs = ((H + 7) // 8) * ((W + 31) // 32) * B * ((K + 63) // 64)
        print("%d->%d @%dx%d B=%2d  wgs %5d (%.2f rounds of 768)  %.1f us  %.2f us/image  %.2f TB/s" % (C, K, H, W, B, wgs, wgs / 768, a, a / B, 4.0 * B * (C + K) * H * W / a / 1e6), flush=True)
