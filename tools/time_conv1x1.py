"""The 1x1 convolutions of DLA's Roots / projections (ops.conv1x1_of_cat) at bs 8: forward, input gradient and weight gradient
timed separately against their HBM floor (bytes / 5 TB/s)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dcd_amd import ops

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
SH = [((64, 64), 64, 96, 320), ((128, 128), 128, 48, 160), ((128, 128, 64, 128), 128, 48, 160), ((64,), 128, 48, 160),
      ((256, 256), 256, 24, 80), ((256, 256, 128, 256), 256, 24, 80), ((128,), 256, 24, 80), ((512, 512, 256), 512, 12, 40), ((256,), 512, 12, 40)]
tot = [0.0, 0.0, 0.0, 0.0]
for cs, O, H, W in SH:
    xs = [torch.randn(B, c, H, W, device=dev) for c in cs]
    C = sum(cs)
    w = torch.randn(O, C, 1, 1, device=dev) / C ** 0.5
    g = torch.randn(B, O, H, W, device=dev)
    w2 = w.reshape(O, C)
    HW = H * W
    g3 = g.view(B, O, HW)
    def fwd():
        return ops.conv1x1_of_cat(xs, w)
    def dgrad():
        c0 = 0
        for x in xs:
            Ci = x.shape[1]
            torch.bmm(w2[:, c0:c0 + Ci].t().unsqueeze(0).expand(B, Ci, O), g3)
            c0 += Ci
    gw = torch.empty_like(w2)
    def wgrad():
        c0 = 0
        for x in xs:
            Ci = x.shape[1]
            torch.sum(torch.bmm(g3, x.view(B, Ci, HW).transpose(1, 2)), 0, out=gw[:, c0:c0 + Ci])
            c0 += Ci
    a, b, c = t(fwd), t(dgrad), t(wgrad)
    by = 4.0 * B * (C + O) * HW
    fl = by / 5e6
    print("%-22s -> %3d @%3dx%3d  fwd %6.1f  dgrad %6.1f  wgrad %6.1f us   floor %5.1f us each" % (cs, O, H, W, a, b, c, fl), flush=True)
    tot[0] += a; tot[1] += b; tot[2] += c; tot[3] += fl
print("total fwd %.0f dgrad %.0f wgrad %.0f us; floor %.0f each" % tuple(tot))
