"""The 1x1 convolutions of DLA's Roots / projections (ops.conv1x1_of_cat) at bs 8 in exact fp32: forward and backward (input
gradients + weight gradient) on the own pointwise kernels (csrc/conv1x1_f32.inc, round 6) against the batched library GEMMs of
rounds 1-5 (DCD_CONV1X1_F32=0), with the HBM floor of one pass (bytes / 5 TB/s) beside them.   python tools/time_conv1x1.py [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dcd_amd import ops
ops._PW_F32_MAX_WEIGHTS = 1 << 30          # time the own kernels on every shape (the dispatch keeps the wide ones on the library)


def t(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
# (input channel groups, Cout, H, W): DLA-34 roots (cat of children [+ level root]) and the 1x1 projections
SH = [((64, 64), 64, 96, 320), ((32,), 64, 96, 320), ((128, 128), 128, 48, 160), ((128, 128, 64, 128), 128, 48, 160), ((64,), 128, 48, 160),
      ((256, 256), 256, 24, 80), ((256, 256, 128, 256), 256, 24, 80), ((128,), 256, 24, 80), ((512, 512, 256), 512, 12, 40), ((256,), 512, 12, 40)]
tot = {}
if os.environ.get("SHAPE"):                       # one shape only (kernel traces)
    SH = [SH[int(os.environ["SHAPE"])]]
for cs, O, H, W in SH:
    xs = [torch.randn(B, c, H, W, device=dev, requires_grad=True) for c in cs]
    C = sum(cs)
    w = (torch.randn(O, C, 1, 1, device=dev) / C ** 0.5).requires_grad_(True)
    g = torch.randn(B, O, H, W, device=dev)
    row = []
    for own in (True, False):
        ops._PW_F32 = own
        fwd = lambda: ops.conv1x1_of_cat(xs, w)
        y = fwd()
        def bwd():
            torch.autograd.grad(y, xs + [w], g, retain_graph=True)
        a, b = t(fwd), t(bwd)
        row += [a, b]
        key = "own" if own else "lib"
        tot[key] = tot.get(key, 0.0) + a + b
    by = 4.0 * B * (C + O) * H * W
    print("%-22s -> %3d @%3dx%3d  own fwd %6.1f bwd %6.1f | library fwd %6.1f bwd %6.1f us   floor %5.1f us per pass" % (
        (cs, O, H, W) + tuple(row) + (by / 5e6,)), flush=True)
print("total fwd+bwd: own %.0f us, library %.0f us (launches under %d pixels stay on the library in both)" % (
    tot["own"], tot["lib"], ops._PW_F32_MIN_PIXELS))
