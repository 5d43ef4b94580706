"""Times the 16 DCN layers of DLA-34-DCN at 384x1280 through the C ABI."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dcd_amd import _ext

LAYERS = [(512, 256, 12, 40, 1), (256, 256, 24, 80, 1), (256, 128, 24, 80, 2), (128, 128, 48, 160, 2),
          (128, 64, 48, 160, 4), (64, 64, 96, 320, 5), (256, 64, 24, 80, 1)]


def main(B=8, prec="f32", iters=5, off_scale=2.0):
    dev = torch.device("cuda:0")
    tot_f = tot_b = 0.0
    for (C, Co, H, W, mult) in LAYERS:
        x = torch.randn(B, C, H, W, device=dev)
        off = torch.randn(B, 18, H, W, device=dev) * off_scale
        m = torch.sigmoid(torch.randn(B, 9, H, W, device=dev))
        w = torch.randn(Co, C, 3, 3, device=dev) / (C * 9) ** 0.5
        b = torch.zeros(Co, device=dev)
        gy = torch.randn(B, Co, H, W, device=dev)
        a = (3, 3, 1, 1, 1, 1, 1, 1, 1)
        for _ in range(2):
            _ext.dcn_v2_forward(x, w, b, off, m, *a, precision=prec)
            _ext.dcn_v2_backward(x, w, b, off, m, gy, *a, precision=prec)
            torch.cuda.synchronize()       # the backward's hand-over policy reads what the layer's previous call reported
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        torch.cuda.synchronize()
        e[0].record()
        for _ in range(iters):
            _ext.dcn_v2_forward(x, w, b, off, m, *a, precision=prec)
        e[1].record()
        for _ in range(iters):
            _ext.dcn_v2_backward(x, w, b, off, m, gy, *a, precision=prec)
        e[2].record()
        torch.cuda.synchronize()
        tf, tb = e[0].elapsed_time(e[1]) / iters, e[1].elapsed_time(e[2]) / iters
        fl = 2.0 * B * Co * 9 * C * H * W
        print("%4d->%3d @%3dx%3d x%d  fwd %.3f ms (%.1f TF)  bwd %.3f ms (%.1f TF)" % (
            C, Co, H, W, mult, tf, fl / tf / 1e9, tb, 2 * fl / tb / 1e9))
        tot_f += tf * mult
        tot_b += tb * mult
    print("off_scale=%g " % off_scale, end="")
    print("B=%d prec=%s TOTAL fwd %.3f ms  bwd %.3f ms  fwd+bwd %.3f ms" % (B, prec, tot_f, tot_b, tot_f + tot_b))


if __name__ == "__main__":
    main(B=int(sys.argv[1]) if len(sys.argv) > 1 else 8, prec=sys.argv[2] if len(sys.argv) > 2 else "f32",
         off_scale=float(sys.argv[3]) if len(sys.argv) > 3 else 2.0)
