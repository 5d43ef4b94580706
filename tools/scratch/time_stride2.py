"""The five stride-2 3x3 convolutions of DLA-34 (stock MIOpen): fp32 NCHW as the step runs them today vs bf16 (NCHW and
channels_last) incl. the casts a mixed-precision region would pay around them -- is a local autocast worth it?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.nn import functional as F

SHAPES = [(16, 32, 384, 1280), (32, 64, 192, 640), (64, 128, 96, 320), (128, 256, 48, 160), (256, 512, 24, 80)]


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
B = 8
tot = {}
for C, K, H, W in SHAPES:
    x = torch.randn(B, C, H, W, device=dev)
    w = torch.randn(K, C, 3, 3, device=dev) / (C * 9) ** 0.5
    gy = torch.randn(B, K, H // 2, W // 2, device=dev)

    def f32():
        y = F.conv2d(x, w, None, 2, 1)
        return torch.ops.aten.convolution_backward(gy, x, w, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1, [True, True, False])

    def bf16():
        xb, wb, gb = x.bfloat16(), w.bfloat16(), gy.bfloat16()
        y = F.conv2d(xb, wb, None, 2, 1).float()
        gx, gw, _ = torch.ops.aten.convolution_backward(gb, xb, wb, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1, [True, True, False])
        return gx.float(), gw.float()

    def bf16_cl():
        xb, wb, gb = (x.bfloat16().contiguous(memory_format=torch.channels_last), w.bfloat16().contiguous(memory_format=torch.channels_last),
                      gy.bfloat16().contiguous(memory_format=torch.channels_last))
        y = F.conv2d(xb, wb, None, 2, 1).float().contiguous()
        gx, gw, _ = torch.ops.aten.convolution_backward(gb, xb, wb, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1, [True, True, False])
        return gx.float().contiguous(), gw.float().contiguous()

    r = {n: t(f) for n, f in (("f32", f32), ("bf16", bf16), ("bf16_cl", bf16_cl))}
    for n, v in r.items():
        tot[n] = tot.get(n, 0.0) + v
    print("%4d->%3d @%3dx%4d s2  fwd+bwd: f32 %.3f ms | bf16 + casts %.3f | bf16 channels_last + casts %.3f" % (C, K, H, W, r["f32"], r["bf16"], r["bf16_cl"]))
print("sum", {k: round(v, 3) for k, v in tot.items()})

# ---- the same layers through space-to-depth on our stride-1 Winograd kernels (ops.conv3x3_stride2), exact fp32 and bf16 operands
from dcd_amd import _ext, ops
for prec in ("f32", "bf16"):
    tot_s = 0.0
    for C, K, H, W in SHAPES:
        x = torch.randn(B, C, H, W, device=dev, requires_grad=True)
        w = (torch.randn(K, C, 3, 3, device=dev) / (C * 9) ** 0.5).requires_grad_(True)
        gy = torch.randn(B, K, H // 2, W // 2, device=dev)

        def ours():
            with _ext.precision_scope(prec):
                y = ops.conv3x3_stride2(x, w)
            return torch.autograd.grad(y, (x, w), gy)

        ref = F.conv2d(x, w, None, 2, 1)
        with _ext.precision_scope(prec):
            got = ops.conv3x3_stride2(x, w)
        err = (got - ref).abs().max().item() / ref.abs().max().item()
        v = t(ours)
        tot_s += v
        print("%4d->%3d @%3dx%4d s2  space-to-depth %s fwd+bwd %.3f ms   (rel err vs stock %.1e)" % (C, K, H, W, prec, v, err))
    print("sum s2d", prec, round(tot_s, 3))
