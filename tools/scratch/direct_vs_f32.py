"""Direct bf16 conv (fwd + bwd-data) against the exact fp32 kernel at DGDE's full-size shapes: relative error of each."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dcd_amd import ops, _ext
dev = torch.device("cuda:0")
B = 8
SH = [(64, 27, 96, 320), (128, 27, 48, 160), (256, 27, 24, 80), (512, 27, 12, 40), (64, 64, 96, 320), (64, 256, 96, 320), (256, 256, 24, 80),
      (512, 512, 12, 40), (1024, 512, 12, 40), (16, 16, 384, 1280), (128, 128, 48, 160), (64, 128, 48, 160)]
for C, K, H, W in SH:
    g = torch.Generator(device=dev).manual_seed(C + K)
    x = torch.randn(B, C, H, W, device=dev, generator=g)
    w = torch.randn(K, C, 3, 3, device=dev, generator=g) / (C * 9) ** 0.5
    gy = torch.randn(B, K, H, W, device=dev, generator=g)
    res = torch.randn(B, C, H, W, device=dev, generator=g)
    y32 = ops._conv3x3_call(x, w, K, False, prec=ops.PREC_F32)
    gx32 = ops._conv3x3_call(gy, w, C, True, residual=res.clone(), prec=ops.PREC_F32)
    with _ext.precision_scope("bf16"):
        tf, tb = ops.conv3x3_transform_weights(w)
        y = ops._conv3x3_call(x, w, K, False, transformed=tf)
        gx = ops._conv3x3_call(gy, w, C, True, residual=res.clone(), transformed=tb)
    ey = ((y - y32).abs().max() / y32.abs().max()).item()
    eg = ((gx - gx32).abs().max() / gx32.abs().max()).item()
    ry = ((y - y32).pow(2).mean().sqrt() / y32.pow(2).mean().sqrt()).item()
    rg = ((gx - gx32).pow(2).mean().sqrt() / gx32.pow(2).mean().sqrt()).item()
    print("%4d->%3d @%3dx%4d  %s  fwd max %.2e rms %.2e   bwd-data max %.2e rms %.2e" % (C, K, H, W, type(tf).__name__, ey, ry, eg, rg), flush=True)
