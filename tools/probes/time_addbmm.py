"""Root weight gradient: bmm(g, x^T).sum(0) against torch.addbmm (one op) on DLA's Root shapes at bs 8 / 1."""
import sys, torch
dev = torch.device("cuda:0")
def t(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for B in (8, 1):
    for O, C, H, W in ((64, 64, 96, 320), (128, 128, 48, 160), (256, 256, 24, 80), (512, 512, 12, 40)):
        g = torch.randn(B, O, H * W, device=dev); x = torch.randn(B, C, H * W, device=dev)
        out = torch.empty(O, C, device=dev)
        a = t(lambda: torch.sum(torch.bmm(g, x.transpose(1, 2)), 0, out=out))
        ref = out.clone()
        b = t(lambda: torch.addbmm(out, g, x.transpose(1, 2), beta=0, out=out))
        err = (out - ref).abs().max().item() / ref.abs().max().item()
        print("B %d O %3d C %3d @%3dx%3d  bmm+sum %.1f us  addbmm %.1f us  rel diff %.1e" % (B, O, C, H, W, a, b, err))

# third form: split-K shifted-view GEMM of csrc/sgemm_f32.inc (dcd_sgemm_shifted: A = g k-contiguous, B rows = x's channel rows)
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dcd_amd import _lib
L = _lib.lib()
for B in (8, 1):
    for O, C, H, W in ((64, 64, 96, 320), (128, 128, 48, 160), (256, 256, 24, 80), (512, 512, 12, 40), (64, 128, 96, 320)):
        HW = H * W
        g = torch.randn(B, O, HW, device=dev); x = torch.randn(B, C, HW, device=dev)
        ref = torch.bmm(g, x.transpose(1, 2)).sum(0)
        off = torch.arange(C, device=dev, dtype=torch.int64) * HW
        for nsplit in (1, 4, 8, 16):
            if HW // nsplit < 64:
                continue
            part = torch.empty(B * nsplit, O, C, device=dev)
            st = torch.cuda.current_stream().cuda_stream
            def run():
                r = L.dcd_sgemm_shifted(st, g.data_ptr(), HW, O * HW, x.data_ptr(), off.data_ptr(), C * HW, 1, None, part.data_ptr(), C,
                                        nsplit * O * C, O * C, O, C, HW, B, nsplit)
                assert r == 0, r
                return part.sum(0)
            us = t(run)
            err = (run() - ref).abs().max().item() / ref.abs().max().item()
            print("B %d O %3d C %3d @%3dx%3d nsplit %2d  sgemm_shifted+sum %.1f us  rel diff %.1e" % (B, O, C, H, W, nsplit, us, err))
