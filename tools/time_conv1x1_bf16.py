"""ops.conv1x1_of_cat forward + backward at DLA's root shapes, fp32 library GEMMs against the bf16 scope's own kernels
(csrc/conv1x1_bf16.inc); whole autograd call (forward, backward), GPU-bound shapes only are meaningful."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dcd_amd import ops, _ext

def t(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
SH = [((64, 64), 64, 96, 320), ((128, 128), 128, 48, 160), ((128, 128, 64, 128), 128, 48, 160), ((64,), 128, 48, 160),
      ((256, 256), 256, 24, 80), ((256, 256, 128, 256), 256, 24, 80), ((128,), 256, 24, 80), ((512, 512, 256), 512, 12, 40), ((256,), 512, 12, 40)]
tot = {"f32": [0, 0], "bf16": [0, 0]}
for cs, O, H, W in SH:
    xs = [torch.randn(B, c, H, W, device=dev, requires_grad=True) for c in cs]
    C = sum(cs)
    w = (torch.randn(O, C, 1, 1, device=dev) / C ** 0.5).requires_grad_(True)
    g = torch.randn(B, O, H, W, device=dev)
    line = "%-22s -> %3d @%3dx%3d " % (cs, O, H, W)
    for mode in ("f32", "bf16"):
        def fwd():
            with _ext.precision_scope(mode):
                return ops.conv1x1_of_cat(xs, w)
        y = fwd()
        def bwd():
            torch.autograd.grad(y, xs + [w], g, retain_graph=True)
        a, b = t(fwd), t(bwd)
        tot[mode][0] += a; tot[mode][1] += b
        line += " | %s fwd %6.1f bwd %6.1f us" % (mode, a, b)
    print(line + " | floor %.0f / %.0f" % (4.0 * B * (C + O) * H * W / 5e6, 2 * 4.0 * B * (C + O) * H * W / 5e6), flush=True)
print("total", tot)
