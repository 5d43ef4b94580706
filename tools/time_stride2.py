"""The five stride-2 3x3 convolutions of DLA-34 at 384x1280 in exact fp32: own kernels (csrc/conv_s2_f32.inc) against the stock
solver and the space-to-depth Winograd form, per direction; checks the own forward against fp64 on the way.
    python tools/time_stride2.py [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.nn import functional as F
from dcd_amd import _lib, ops


def t(fn, iters=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L = _lib.lib()
tot = {}
for C, K, H, W in [(16, 32, 384, 1280), (32, 64, 192, 640), (64, 128, 96, 320), (128, 256, 48, 160), (256, 512, 24, 80)]:
    x = torch.randn(B, C, H, W, device=dev)
    w = torch.randn(K, C, 3, 3, device=dev) / (C * 9) ** 0.5
    gy = torch.randn(B, K, H // 2, W // 2, device=dev)
    y = torch.empty(B, K, H // 2, W // 2, device=dev)
    st = _lib.stream_of(x)
    def own_fwd():
        _lib.check(L.dcd_conv3x3_s2_f32(st, x.data_ptr(), w.data_ptr(), y.data_ptr(), B, C, H, W, K), "dcd_conv3x3_s2_f32")
    gx = torch.empty_like(x)
    gw = torch.empty_like(w)
    nws = L.dcd_conv3x3_s2_f32_wrw_workspace_bytes(B, C, H, W, K)
    ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=dev)
    def own_dgrad():
        _lib.check(L.dcd_conv3x3_s2_f32_backward_data(st, gy.data_ptr(), w.data_ptr(), gx.data_ptr(), B, C, H, W, K), "s2 dgrad")
    def own_wgrad():
        _lib.check(L.dcd_conv3x3_s2_f32_wrw(st, x.data_ptr(), gy.data_ptr(), gw.data_ptr(), B, C, H, W, K, ws.data_ptr(), nws), "s2 wrw")
    own_fwd(); own_dgrad(); own_wgrad()
    ref = F.conv2d(x[:1].double(), w.double(), None, 2, 1)
    err = float((y[:1].double() - ref).abs().max() / ref.abs().max())
    rgx = torch.nn.grad.conv2d_input(x[:1].shape, w.double(), gy[:1].double(), stride=2, padding=1)
    err_gx = float((gx[:1].double() - rgx).abs().max() / rgx.abs().max())
    rgw = torch.nn.grad.conv2d_weight(x.double(), w.shape, gy.double(), stride=2, padding=1)
    err_gw = float((gw.double() - rgw).abs().max() / rgw.abs().max())
    row = {"own fwd": t(own_fwd), "own dgrad": t(own_dgrad), "own wgrad": t(own_wgrad), "stock fwd": t(lambda: F.conv2d(x, w, None, 2, 1)),
           "stock dgrad": t(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])),
           "stock wgrad": t(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False]))}
    fl = 2.0 * B * K * C * 9 * (H // 2) * (W // 2)
    print("%3d->%3d @%3dx%4d  " % (C, K, H, W) + "  ".join("%s %6.1f us (%3.0f TF)" % (k, v, fl / v / 1e6) for k, v in row.items()) + "  | err fwd %.1e dgrad %.1e wgrad %.1e" % (err, err_gx, err_gw), flush=True)
    for k, v in row.items(): tot[k] = tot.get(k, 0) + v
print("total: " + "  ".join("%s %.0f us" % kv for kv in tot.items()))
