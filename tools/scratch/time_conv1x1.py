"""1x1 convolution as a batched GEMM: torch.bmm (hipBLASLt) vs dcd_sgemm (csrc/sgemm_f32.inc) on DLA's Root shapes at bs 8."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dcd_amd import _lib


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
L = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for O, C, H, W in ((64, 64, 96, 320), (64, 128, 96, 320), (128, 128, 48, 160), (128, 256, 48, 160), (128, 448, 48, 160),
                   (256, 256, 24, 80), (256, 512, 24, 80), (256, 896, 24, 80), (512, 512, 12, 40), (512, 1280, 12, 40)):
    HW = H * W
    x = torch.randn(B, C, HW, device=dev)
    w = torch.randn(O, C, device=dev) / C ** 0.5
    out = torch.empty(B, O, HW, device=dev)
    out2 = torch.empty(B, O, HW, device=dev)
    we = w.unsqueeze(0).expand(B, O, C)
    a = t(lambda: torch.bmm(we, x, out=out))
    st = torch.cuda.current_stream().cuda_stream

    def ours():
        r = L.dcd_sgemm(st, w.data_ptr(), C, 0, 1, x.data_ptr(), HW, C * HW, 0, out2.data_ptr(), HW, O * HW, O, HW, C, B, 1.0, 0, 0)
        assert r == 0, r
    b = t(ours)
    err = (out - out2).abs().max().item()
    mb = B * HW * (C + O) * 4 / 1e6
    print("O %4d C %4d @%3dx%3d  bmm %7.1f us (%.2f TB/s)  sgemm %7.1f us (%.2f TB/s)  maxdiff %.2e" % (O, C, H, W, a, mb / a, b, mb / b, err))
