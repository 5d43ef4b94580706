"""Pins the C oracle of DCNv2 with the reference's OWN tests, restated (DCN/testcpu.py), plus a cross-check of
its hand-written backward against torch autograd of the closed-form formula (oracle/dcn_torch.py)."""
import torch
from torch.autograd import gradcheck

from oracle.dcn_torch import dcn_v2_reference

N, inC, inH, inW, outC, kH, kW = 2, 2, 4, 4, 2, 3, 3      # DCN/testcpu.py:15-17


def test_check_zero_offset(oracle_dcn):
    """DCN/testcpu.py:32-67: zero offsets, mask = sigmoid(0) = 0.5, identity weight  =>  2*out == in (1e-10)."""
    torch.manual_seed(0)
    weight = torch.zeros(outC, inC, kH, kW)
    for p in range(inC):
        weight[p, p, kH // 2, kW // 2] = 1.0
    bias = torch.zeros(outC)
    x = torch.randn(N, inC, inH, inW)
    offset = torch.zeros(N, 2 * kH * kW, inH, inW)
    mask = torch.sigmoid(torch.zeros(N, kH * kW, inH, inW))
    out = oracle_dcn.dcn_v2_forward(x, weight, bias, offset, mask, kH, kW, 1, 1, 1, 1, 1, 1, 1) * 2
    assert (x - out).abs().max().item() < 1e-10


class _OracleDCN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, offset, mask, weight, bias):
        from oracle import dcn_oracle
        ctx.save_for_backward(x, offset, mask, weight, bias)
        return dcn_oracle.dcn_v2_forward(x, weight, bias, offset, mask, 3, 3, 1, 1, 1, 1, 1, 1, 1)

    @staticmethod
    def backward(ctx, gy):
        from oracle import dcn_oracle
        x, offset, mask, weight, bias = ctx.saved_tensors
        gi, go, gm, gw, gb = dcn_oracle.dcn_v2_backward(x, weight, bias, offset, mask, gy.contiguous(), 3, 3, 1, 1, 1, 1, 1, 1, 1)
        return gi, go, gm, gw, gb


def test_check_gradient_dconv(oracle_dcn):
    """DCN/testcpu.py:69-97: gradcheck(dcn_v2_conv, eps=1e-3, atol=1e-4, rtol=1e-2) with input ~ U(0,0.01),
    offset ~ 2*N(0,1), mask = sigmoid(U(0,1)).  Run on the float64 instantiation: in fp32 the finite differences
    across bilinear kinks make the reference's own check flaky (its comment at testcpu.py:266-270)."""
    torch.manual_seed(3)
    x = (torch.rand(N, inC, inH, inW, dtype=torch.float64) * 0.01).requires_grad_()
    offset = (torch.randn(N, 2 * kW * kH, inH, inW, dtype=torch.float64) * 2)
    # keep samples away from integer coordinates so central differences do not straddle a kink
    frac = offset - torch.floor(offset)
    offset = (torch.floor(offset) + frac.clamp(0.05, 0.95)).requires_grad_()
    mask = torch.sigmoid(torch.rand(N, kW * kH, inH, inW, dtype=torch.float64)).requires_grad_()
    weight = torch.randn(outC, inC, kH, kW, dtype=torch.float64, requires_grad=True)
    bias = torch.rand(outC, dtype=torch.float64, requires_grad=True)
    assert gradcheck(_OracleDCN.apply, (x, offset, mask, weight, bias), eps=1e-3, atol=1e-4, rtol=1e-2)


def test_oracle_matches_autograd_formula(oracle_dcn):
    """Hand-written backward of the oracle == autograd of the restated formula, f32 and f64, incl. samples that
    leave the image."""
    for dtype, tol in ((torch.float32, 2e-5), (torch.float64, 1e-12)):
        torch.manual_seed(1)
        B, C, Co, H, W = 2, 5, 4, 7, 9
        x = torch.randn(B, C, H, W, dtype=dtype, requires_grad=True)
        off = (torch.randn(B, 18, H, W, dtype=dtype) * 3).requires_grad_()
        m = torch.sigmoid(torch.randn(B, 9, H, W, dtype=dtype)).requires_grad_()
        w = torch.randn(Co, C, 3, 3, dtype=dtype, requires_grad=True)
        b = torch.randn(Co, dtype=dtype, requires_grad=True)
        ref = dcn_v2_reference(x, off, m, w, b)
        got = oracle_dcn.dcn_v2_forward(x, w, b, off, m, 3, 3, 1, 1, 1, 1, 1, 1, 1)
        assert (ref - got).abs().max().item() <= tol * ref.abs().max().item()
        gy = torch.randn_like(ref)
        ref.backward(gy)
        grads = oracle_dcn.dcn_v2_backward(x, w, b, off, m, gy, 3, 3, 1, 1, 1, 1, 1, 1, 1)
        for g, r in zip(grads, (x.grad, off.grad, m.grad, w.grad, b.grad)):
            assert (g - r).abs().max().item() <= 10 * tol * r.abs().max().item()


def test_example_dconv_shapes(oracle_dcn):
    """DCN/testcpu.py:169-180 (scaled down): deformable_groups=2 runs forward and backward."""
    torch.manual_seed(2)
    x = torch.randn(2, 8, 16, 16)
    w, b = torch.randn(8, 8, 3, 3) * 0.1, torch.zeros(8)
    off, m = torch.randn(2, 36, 16, 16), torch.sigmoid(torch.randn(2, 18, 16, 16))
    out = oracle_dcn.dcn_v2_forward(x, w, b, off, m, 3, 3, 1, 1, 1, 1, 1, 1, 2)
    assert tuple(out.shape) == (2, 8, 16, 16)
    grads = oracle_dcn.dcn_v2_backward(x, w, b, off, m, torch.ones_like(out), 3, 3, 1, 1, 1, 1, 1, 1, 2)
    assert [tuple(g.shape) for g in grads] == [(2, 8, 16, 16), (2, 36, 16, 16), (2, 18, 16, 16), (8, 8, 3, 3), (8,)]
