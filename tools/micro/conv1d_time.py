"""GMW extractor's 1x1 Conv1d (128 -> 128 on (8, 128, 2628)): stock conv1d / matmul against dcd_sgemm (tools/micro)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.nn import functional as F
from dcd_amd import _lib

dev = torch.device("cuda:0")
B, C, K = 8, 128, 2628
x = torch.randn(B, C, K, device=dev)
w = torch.randn(C, C, 1, device=dev) / C ** 0.5
b = torch.randn(C, device=dev)
gy = torch.randn(B, C, K, device=dev)


def t(fn, iters=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


L = _lib.lib()
y = torch.empty(B, C, K, device=dev)
w2 = w.reshape(C, C).contiguous()


def ours_fwd():
    st = L.dcd_sgemm(_lib.stream_of(x), w2.data_ptr(), C, 0, 1, x.data_ptr(), K, C * K, 0, y.data_ptr(), K, C * K, C, K, C, B, 1.0, 0, 0)
    assert st == 0


gx = torch.empty(B, C, K, device=dev)


def ours_bwd_data():           # gx = W^T gy: A(m,k) = W[k][m] -> a_kcontig = 0
    st = L.dcd_sgemm(_lib.stream_of(x), w2.data_ptr(), C, 0, 0, gy.data_ptr(), K, C * K, 0, gx.data_ptr(), K, C * K, C, K, C, B, 1.0, 0, 0)
    assert st == 0


print("conv1d fwd  %.1f us" % t(lambda: F.conv1d(x, w, b)))
print("matmul fwd  %.1f us" % t(lambda: torch.matmul(w2, x)))
print("ours   fwd  %.1f us" % t(ours_fwd))
print("ours   bwd-data %.1f us" % t(ours_bwd_data))
ref = torch.matmul(w2, x)
ours_fwd()
print("max err", (y - ref).abs().max().item(), "scale", ref.abs().max().item())
print("stock bwd (data+weight) %.1f us" % t(lambda: torch.ops.aten.convolution_backward(gy, x, w, [C], [1], [0], [1], False, [0], 1, [True, True, True])))
print("wrw as matmul %.1f us" % t(lambda: torch.matmul(gy, x.transpose(1, 2)).sum(0)))
for mask in ([True, False, False], [False, True, False], [False, False, True], [True, True, False]):
    print("stock bwd", mask, "%.1f us" % t(lambda: torch.ops.aten.convolution_backward(gy, x, w, [C], [1], [0], [1], False, [0], 1, mask)))
from dcd_amd import ops
print("channel_sums %.1f us" % t(lambda: ops.channel_sums(gy)))
print("wrw bmm+sum: bmm %.1f us" % t(lambda: torch.bmm(gy, x.transpose(1, 2))))
gy2, x2 = gy.permute(1, 0, 2).reshape(C, B * K), x.permute(1, 0, 2).reshape(C, B * K)
print("wrw single mm on permuted copies (copies excluded) %.1f us" % t(lambda: torch.mm(gy2, x2.t())))
