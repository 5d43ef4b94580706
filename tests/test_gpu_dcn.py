"""GPU parity: HIP DCNv2 (through the C ABI, dcd_amd._ext) vs the CPU oracle on identical inputs.

Tolerance: north_star asks for 1e-3 relative in fp32.  The f32 MFMA path is an exact-fp32 fmaf chain, so
it is held to 2e-5 of the output scale (summation-order noise only); the bound is written in `close()`.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _fresh_launch_policy(cuda):
    """The DCN backward's per-layer launch policy is keyed by the weight's device address, and the caching allocator hands a freed
    address to the next test: a test would inherit the far-sample history of whichever test used that block before it (results
    never depend on it, launch SEQUENCES do -- and some tests here assert on the sequence).  Every test starts with no history."""
    from dcd_amd import _lib
    _lib.lib().dcd_dcn_v2_forget(None)
    yield


def close(got, ref, rel, what):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = ref.abs().max().item() + 1e-12
    err = (got - ref).abs().max().item()
    assert err <= rel * scale, "%s: max abs err %.3e vs scale %.3e (rel %.2e > %.1e)" % (what, err, scale, err / scale, rel)


def make_case(B, C, Co, H, W, dg=1, k=3, off_scale=2.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, C, H, W, generator=g)
    off = torch.randn(B, 2 * dg * k * k, H, W, generator=g) * off_scale
    m = torch.sigmoid(torch.randn(B, dg * k * k, H, W, generator=g))
    w = torch.randn(Co, C, k, k, generator=g) / (C * k * k) ** 0.5
    b = torch.randn(Co, generator=g)
    gy = torch.randn(B, Co, H, W, generator=g)
    return x, w, b, off, m, gy


CASES = [
    # B, C, Co, H, W, dg, off_scale      (reference test shapes: DCN/testcpu.py:15-17; DGDE layer shapes scaled down)
    (2, 2, 2, 4, 4, 1, 2.0),
    (2, 5, 4, 7, 9, 1, 3.0),
    (1, 64, 64, 24, 40, 1, 2.0),
    (2, 128, 64, 12, 20, 1, 2.0),
    (1, 256, 128, 6, 10, 1, 1.0),
    (2, 64, 32, 9, 11, 2, 2.0),      # deformable_groups = 2 (DCN/testcpu.py:172-173)
    (1, 6, 3, 5, 6, 3, 6.0),         # odd channels per group, samples far outside the image
    (1, 96, 320, 8, 8, 1, 1.0),      # Cout > 256: streamed dY path
    # shapes that take the workgroup-tiled LDS kernels (3x3 s1 p1, dg 1, W % 4 == 0, H >= 16, W >= 32)
    (2, 64, 64, 24, 64, 1, 0.5),     # sub-pixel offsets: everything from the staged window
    (1, 12, 70, 17, 36, 1, 1.5),     # ragged: C % 8 != 0, Cout > 64 with a partial slice, partial tiles, some far samples
    (1, 32, 64, 16, 32, 1, 4.0),     # many samples beyond the 3-px window -> per-lane global fallback
    (2, 64, 64, 16, 64, 1, 12.0),    # offsets of many pixels everywhere: tiled kernels stand down (device-side switch)
]


@pytest.mark.parametrize("case", CASES)
def test_forward_matches_oracle(cuda, oracle_dcn, case):
    from dcd_amd import _ext
    B, C, Co, H, W, dg, osc = case
    x, w, b, off, m, _ = make_case(B, C, Co, H, W, dg, off_scale=osc)
    ref = oracle_dcn.dcn_v2_forward(x, w, b, off, m, 3, 3, 1, 1, 1, 1, 1, 1, dg)
    got = _ext.dcn_v2_forward(x.to(cuda), w.to(cuda), b.to(cuda), off.to(cuda), m.to(cuda), 3, 3, 1, 1, 1, 1, 1, 1, dg)
    close(got, ref, 2e-5, "forward %s" % (case,))


@pytest.mark.parametrize("case", CASES)
def test_backward_matches_oracle(cuda, oracle_dcn, case):
    from dcd_amd import _ext
    B, C, Co, H, W, dg, osc = case
    x, w, b, off, m, gy = make_case(B, C, Co, H, W, dg, off_scale=osc, seed=1)
    ref = oracle_dcn.dcn_v2_backward(x, w, b, off, m, gy, 3, 3, 1, 1, 1, 1, 1, 1, dg)
    got = _ext.dcn_v2_backward(x.to(cuda), w.to(cuda), b.to(cuda), off.to(cuda), m.to(cuda), gy.to(cuda),
                               3, 3, 1, 1, 1, 1, 1, 1, dg)
    for name, g_, r_ in zip(("grad_input", "grad_offset", "grad_mask", "grad_weight", "grad_bias"), got, ref):
        close(g_, r_, 5e-5, "%s %s" % (name, case))


# ---- DCD_PREC_BF16X3: split-bf16 contraction (hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16, fp32 accumulate).
# north_star's bound is 1e-3 relative; the split form loses ~2^-16 per product, so it is held to 1e-4 of the output scale
# (1e-3 / 10) against the SAME fp32 oracle.  Geometries without a split kernel run the exact fp32 kernels (allowed by the ABI).
BF16X3_TOL = 1e-4


@pytest.mark.parametrize("case", CASES)
def test_forward_matches_oracle_bf16x3(cuda, oracle_dcn, case):
    from dcd_amd import _ext
    B, C, Co, H, W, dg, osc = case
    x, w, b, off, m, _ = make_case(B, C, Co, H, W, dg, off_scale=osc)
    ref = oracle_dcn.dcn_v2_forward(x, w, b, off, m, 3, 3, 1, 1, 1, 1, 1, 1, dg)
    got = _ext.dcn_v2_forward(x.to(cuda), w.to(cuda), b.to(cuda), off.to(cuda), m.to(cuda), 3, 3, 1, 1, 1, 1, 1, 1, dg,
                              precision="bf16x3")
    close(got, ref, BF16X3_TOL, "bf16x3 forward %s" % (case,))


@pytest.mark.parametrize("case", CASES)
def test_backward_matches_oracle_bf16x3(cuda, oracle_dcn, case):
    from dcd_amd import _ext
    B, C, Co, H, W, dg, osc = case
    x, w, b, off, m, gy = make_case(B, C, Co, H, W, dg, off_scale=osc, seed=1)
    ref = oracle_dcn.dcn_v2_backward(x, w, b, off, m, gy, 3, 3, 1, 1, 1, 1, 1, 1, dg)
    got = _ext.dcn_v2_backward(x.to(cuda), w.to(cuda), b.to(cuda), off.to(cuda), m.to(cuda), gy.to(cuda),
                               3, 3, 1, 1, 1, 1, 1, 1, dg, precision="bf16x3")
    for name, g_, r_ in zip(("grad_input", "grad_offset", "grad_mask", "grad_weight", "grad_bias"), got, ref):
        close(g_, r_, BF16X3_TOL, "bf16x3 %s %s" % (name, case))


def test_bf16x3_path_is_the_split_kernel(cuda, oracle_dcn):
    """On a tiled geometry the split kernels must be what runs: the result differs from the exact-fp32 path in the low bits
    (it is not the f32 kernel under another name) yet stays ~2^-16-close; forward and grad_weight (the two split kernels)."""
    from dcd_amd import _ext
    x, w, b, off, m, gy = (t.to(cuda) for t in make_case(2, 64, 64, 24, 64, off_scale=0.5, seed=9))
    a = (3, 3, 1, 1, 1, 1, 1, 1, 1)
    y32 = _ext.dcn_v2_forward(x, w, b, off, m, *a, precision="f32")
    y16 = _ext.dcn_v2_forward(x, w, b, off, m, *a, precision="bf16x3")
    assert not torch.equal(y32, y16)
    rel = (y32 - y16).abs().max().item() / y32.abs().max().item()
    assert 1e-8 < rel < 3e-5, rel
    g32 = _ext.dcn_v2_backward(x, w, b, off, m, gy, *a, precision="f32")
    g16 = _ext.dcn_v2_backward(x, w, b, off, m, gy, *a, precision="bf16x3")
    assert not torch.equal(g32[3], g16[3])
    rel = (g32[3] - g16[3]).abs().max().item() / g32[3].abs().max().item()
    assert 1e-8 < rel < 3e-5, rel


def test_bf16x3_dense_path_is_the_split_gemm(cuda):
    """The column-buffer path (wide inputs) under DCD_PREC_BF16X3 runs its three products on sgemm_bf16x3.inc: forward, input /
    coordinate gradients (through T = W^T dY) and grad_weight all differ from the exact-fp32 result in the low bits and stay
    ~2^-16-close; odd sizes exercise the tile edges and the k tail of the split-K chunks."""
    from dcd_amd import _ext
    a = (3, 3, 1, 1, 1, 1, 1, 1, 1)
    for C, Co, H, W in ((256, 256, 24, 80), (256, 72, 12, 20)):
        x, w, b, off, m, gy = (t.to(cuda) for t in make_case(2, C, Co, H, W, off_scale=0.5, seed=11))
        y32 = _ext.dcn_v2_forward(x, w, b, off, m, *a, precision="f32")
        y16 = _ext.dcn_v2_forward(x, w, b, off, m, *a, precision="bf16x3")
        assert not torch.equal(y32, y16)
        rel = (y32 - y16).abs().max().item() / y32.abs().max().item()
        assert 1e-8 < rel < 3e-5, ("forward", C, Co, rel)
        g32 = _ext.dcn_v2_backward(x, w, b, off, m, gy, *a, precision="f32")
        g16 = _ext.dcn_v2_backward(x, w, b, off, m, gy, *a, precision="bf16x3")
        for name, p32, p16 in zip(("grad_input", "grad_offset", "grad_mask", "grad_weight"), g32, g16):
            assert not torch.equal(p32, p16), name
            rel = (p32 - p16).abs().max().item() / p32.abs().max().item()
            assert 1e-8 < rel < 3e-5, (name, C, Co, rel)
        assert torch.equal(g32[4], g16[4])                       # the bias gradient is a plain sum in both


@pytest.mark.parametrize("geom", [(64, 64, 96, 320), (128, 64, 48, 160), (256, 256, 24, 80)])
def test_full_size_bf16x3(cuda, oracle_dcn, geom):
    """PARITY (image 0 of 8 against the fp32 oracle at 1e-4).  BASELINE batch (8) in split precision: forward and all five gradients."""
    from dcd_amd import _ext
    C, Co, H, W = geom
    x, w, b, off, m, gy = make_case(8, C, Co, H, W, seed=7, off_scale=0.5)
    xd, wd, bd, od, md, gd = (t.to(cuda) for t in (x, w, b, off, m, gy))
    a = (3, 3, 1, 1, 1, 1, 1, 1, 1)
    y = _ext.dcn_v2_forward(xd, wd, bd, od, md, *a, precision="bf16x3")
    close(y[:1], oracle_dcn.dcn_v2_forward(x[:1], w, b, off[:1], m[:1], *a), BF16X3_TOL, "bf16x3 image 0 forward")
    g1 = _ext.dcn_v2_backward(xd[:1].contiguous(), wd, bd, od[:1].contiguous(), md[:1].contiguous(), gd[:1].contiguous(), *a,
                              precision="bf16x3")
    refg = oracle_dcn.dcn_v2_backward(x[:1], w, b, off[:1], m[:1], gy[:1], *a)
    for name, g_, r_ in zip(("grad_input", "grad_offset", "grad_mask", "grad_weight", "grad_bias"), g1, refg):
        close(g_, r_, BF16X3_TOL, "bf16x3 full-size image 0 " + name)


# ---- DCD_PREC_BF16: mixed precision (MODEL.FP16): both operands of the weight contraction rounded to bf16 (2^-9 each), ONE
# product on the bf16 matrix cores, fp32 accumulate; sampling / coordinate arithmetic and sums stay fp32.  Against the SAME fp32
# oracle the error is the operand rounding: ~2^-8 per product, random in sign, i.e. ~1e-3 of the output scale after a sum over
# 9 Cin terms -- held to 1e-2 of the output scale (max norm) and required to be ABOVE 1e-6 on the kernels that have the form
# (it is not the fp32 kernel under another name).
BF16_TOL = 1e-2


@pytest.mark.parametrize("case", CASES)
def test_forward_and_backward_match_oracle_bf16(cuda, oracle_dcn, case):
    from dcd_amd import _ext
    B, C, Co, H, W, dg, osc = case
    x, w, b, off, m, gy = make_case(B, C, Co, H, W, dg, off_scale=osc, seed=1)
    a = (3, 3, 1, 1, 1, 1, 1, 1, dg)
    dev = [t.to(cuda) for t in (x, w, b, off, m)]
    ref = oracle_dcn.dcn_v2_forward(x, w, b, off, m, *a)
    got = _ext.dcn_v2_forward(*dev, *a, precision="bf16")
    close(got, ref, BF16_TOL, "bf16 forward %s" % (case,))
    refg = oracle_dcn.dcn_v2_backward(x, w, b, off, m, gy, *a)
    gotg = _ext.dcn_v2_backward(*dev, gy.to(cuda), *a, precision="bf16")
    for name, g_, r_ in zip(("grad_input", "grad_offset", "grad_mask", "grad_weight", "grad_bias"), gotg, refg):
        close(g_, r_, BF16_TOL, "bf16 %s %s" % (name, case))


@pytest.mark.parametrize("shape", [(64, 64, 24, 64), (128, 128, 16, 32), (256, 256, 24, 80), (256, 72, 12, 20)])
def test_bf16_path_is_the_one_product_kernel(cuda, shape):
    """Tiled forward, one-pass backward (Cout <= 64 and the wide variant) and the dense path's products under DCD_PREC_BF16: every
    result that passes through a weight contraction differs from the exact-fp32 one by bf16 operand rounding -- more than the split
    form's 3e-5, less than 1e-2 of its scale; the bias gradient (a plain sum) is bitwise the same."""
    from dcd_amd import _ext
    C, Co, H, W = shape
    x, w, b, off, m, gy = (t.to(cuda) for t in make_case(2, C, Co, H, W, off_scale=0.5, seed=13))
    a = (3, 3, 1, 1, 1, 1, 1, 1, 1)
    y32 = _ext.dcn_v2_forward(x, w, b, off, m, *a, precision="f32")
    y16 = _ext.dcn_v2_forward(x, w, b, off, m, *a, precision="bf16")
    rel = (y32 - y16).abs().max().item() / y32.abs().max().item()
    assert 1e-4 < rel < BF16_TOL, ("forward", rel)
    g32 = _ext.dcn_v2_backward(x, w, b, off, m, gy, *a, precision="f32")
    g16 = _ext.dcn_v2_backward(x, w, b, off, m, gy, *a, precision="bf16")
    for name, p32, p16 in zip(("grad_input", "grad_offset", "grad_mask", "grad_weight"), g32, g16):
        rel = (p32 - p16).abs().max().item() / p32.abs().max().item()
        assert 1e-4 < rel < BF16_TOL, (name, shape, rel)
    # the bias gradient is a plain fp32 sum of dY in both (partial sums meet in atomics: equal to summation order)
    assert (g32[4] - g16[4]).abs().max().item() <= 1e-5 * g32[4].abs().max().item()


@pytest.mark.parametrize("geom", [(64, 64, 96, 320), (256, 128, 24, 80)])
def test_full_size_bf16(cuda, oracle_dcn, geom):
    """PARITY (image 0 of 8 against the fp32 oracle at 1e-2) + PROPERTY (additivity of the batch for the other seven).
    BASELINE batch (8) in mixed precision: image 0 against the fp32 oracle, forward and all five gradients; the other images
    through additivity of the batch (grad_weight / grad_bias of the batch = sum over single-image calls, to summation order)."""
    from dcd_amd import _ext
    C, Co, H, W = geom
    x, w, b, off, m, gy = make_case(8, C, Co, H, W, seed=7, off_scale=0.5)
    xd, wd, bd, od, md, gd = (t.to(cuda) for t in (x, w, b, off, m, gy))
    a = (3, 3, 1, 1, 1, 1, 1, 1, 1)
    y = _ext.dcn_v2_forward(xd, wd, bd, od, md, *a, precision="bf16")
    close(y[:1], oracle_dcn.dcn_v2_forward(x[:1], w, b, off[:1], m[:1], *a), BF16_TOL, "bf16 image 0 forward")
    g1 = _ext.dcn_v2_backward(xd[:1].contiguous(), wd, bd, od[:1].contiguous(), md[:1].contiguous(), gd[:1].contiguous(), *a,
                              precision="bf16")
    refg = oracle_dcn.dcn_v2_backward(x[:1], w, b, off[:1], m[:1], gy[:1], *a)
    for name, g_, r_ in zip(("grad_input", "grad_offset", "grad_mask", "grad_weight", "grad_bias"), g1, refg):
        close(g_, r_, BF16_TOL, "bf16 full-size image 0 " + name)
    gall = _ext.dcn_v2_backward(xd, wd, bd, od, md, gd, *a, precision="bf16")
    close(gall[0][:1], g1[0], 1e-4, "bf16 batch vs single image: grad_input")
    close(gall[1][:1], g1[1], 1e-4, "bf16 batch vs single image: grad_offset")
    gw = sum(_ext.dcn_v2_backward(xd[i:i + 1].contiguous(), wd, bd, od[i:i + 1].contiguous(), md[i:i + 1].contiguous(),
                                  gd[i:i + 1].contiguous(), *a, precision="bf16")[3] for i in range(8))
    close(gall[3], gw, 1e-4, "bf16 batch grad_weight = sum of the images'")


@pytest.mark.parametrize("geom", [(3, 3, 2, 2, 1, 1, 1, 1), (3, 3, 1, 1, 2, 2, 2, 2), (1, 1, 1, 1, 0, 0, 1, 1),
                                  (3, 1, 1, 2, 1, 0, 1, 1)])
def test_general_geometry(cuda, oracle_dcn, geom):
    """stride / dilation / non-square kernels: the reference op supports them (dcn_v2_cuda.cu:86-87)."""
    from dcd_amd import _ext
    kh, kw, sh, sw, ph, pw, dh, dw = geom
    B, C, Co, H, W = 2, 8, 6, 11, 13
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, C, H, W, generator=g)
    off = torch.randn(B, 2 * kh * kw, Ho, Wo, generator=g) * 2
    m = torch.rand(B, kh * kw, Ho, Wo, generator=g)
    w = torch.randn(Co, C, kh, kw, generator=g) * 0.2
    b = torch.randn(Co, generator=g)
    gy = torch.randn(B, Co, Ho, Wo, generator=g)
    args = (kh, kw, sh, sw, ph, pw, dh, dw, 1)
    ref = oracle_dcn.dcn_v2_forward(x, w, b, off, m, *args)
    got = _ext.dcn_v2_forward(x.to(cuda), w.to(cuda), b.to(cuda), off.to(cuda), m.to(cuda), *args)
    close(got, ref, 2e-5, "forward geom %s" % (geom,))
    refg = oracle_dcn.dcn_v2_backward(x, w, b, off, m, gy, *args)
    gotg = _ext.dcn_v2_backward(x.to(cuda), w.to(cuda), b.to(cuda), off.to(cuda), m.to(cuda), gy.to(cuda), *args)
    for name, g_, r_ in zip(("grad_input", "grad_offset", "grad_mask", "grad_weight", "grad_bias"), gotg, refg):
        close(g_, r_, 5e-5, "%s geom %s" % (name, geom))


def test_zero_offset_known_answer(cuda):
    """The reference's check_zero_offset (DCN/testcpu.py:32-67): identity weight, mask 0.5 -> 2*out == in."""
    from dcd_amd import _ext
    torch.manual_seed(0)
    N, inC, H, W, outC = 2, 2, 4, 4, 2
    w = torch.zeros(outC, inC, 3, 3)
    for p in range(inC):
        w[p, p, 1, 1] = 1.0
    x = torch.randn(N, inC, H, W)
    off = torch.zeros(N, 18, H, W)
    m = torch.sigmoid(torch.zeros(N, 9, H, W))
    out = _ext.dcn_v2_forward(x.to(cuda), w.to(cuda), torch.zeros(outC, device=cuda), off.to(cuda), m.to(cuda),
                              3, 3, 1, 1, 1, 1, 1, 1, 1)
    assert (x - 2 * out.cpu()).abs().max().item() < 1e-10


def test_boundary_samples(cuda, oracle_dcn):
    """Samples landing in (-1,0) and (H-1,H): partially outside, must follow the open-interval rule
    (cuda/dcn_v2_im2col_cuda.cu:180) and the per-corner bounds (:38-48)."""
    from dcd_amd import _ext
    B, C, Co, H, W = 1, 4, 4, 6, 7
    x, w, b, off, m, gy = make_case(B, C, Co, H, W, seed=3)
    off.zero_()
    off[:, 0::2] = torch.tensor([-0.5, -1.0, -1.5, 0.25, 6.5, 5.99, -0.999, 5.0, 7.0]).view(1, 9, 1, 1)
    off[:, 1::2] = torch.tensor([-0.5, -0.25, 7.5, 6.99, -1.0, 0.5, -0.999, 6.0, -2.0]).view(1, 9, 1, 1)
    ref = oracle_dcn.dcn_v2_forward(x, w, b, off, m, 3, 3, 1, 1, 1, 1, 1, 1, 1)
    got = _ext.dcn_v2_forward(x.to(cuda), w.to(cuda), b.to(cuda), off.to(cuda), m.to(cuda), 3, 3, 1, 1, 1, 1, 1, 1, 1)
    close(got, ref, 2e-5, "boundary forward")
    refg = oracle_dcn.dcn_v2_backward(x, w, b, off, m, gy, 3, 3, 1, 1, 1, 1, 1, 1, 1)
    gotg = _ext.dcn_v2_backward(x.to(cuda), w.to(cuda), b.to(cuda), off.to(cuda), m.to(cuda), gy.to(cuda),
                                3, 3, 1, 1, 1, 1, 1, 1, 1)
    for name, g_, r_ in zip(("grad_input", "grad_offset", "grad_mask", "grad_weight", "grad_bias"), gotg, refg):
        close(g_, r_, 5e-5, "boundary " + name)


@pytest.mark.parametrize("W,sweep", [(20, "1"), (48, "1"), (48, "0")])
def test_convergent_offsets_overflow_and_far_fallback(cuda, oracle_dcn, monkeypatch, W, sweep):
    """Three-pass backward: grad_input comes from inverse sample lists (capacity 10 per cell and tap, radius <= 3 px) with an
    atomic fallback.  Offsets that make every pixel of a 4x4 block sample the SAME location overflow the lists; offsets of
    4..7 px are 'far'.  Both fallbacks and their mix with the list path must reproduce the oracle.
    One-pass backward (W = 48): the same offsets make the four quarters of every pixel step meet in one cell -- every tap goes
    through the fix-up loop (LDS atomics) -- and the far half through the generic kernels' far-only pass."""
    from dcd_amd import _ext
    monkeypatch.setenv("DCD_BWD_SWEEP", sweep)
    B, C, Co, H = 2, 8, 8, 16
    x, w, b, off, m, gy = make_case(B, C, Co, H, W, seed=11)
    ys = torch.arange(H).view(1, 1, H, 1).float()
    xs = torch.arange(W).view(1, 1, 1, W).float()
    for k in range(9):
        i, j = k // 3, k % 3
        # un-deformed tap position is (y-1+i, x-1+j); send it to the centre of the pixel's 4x4 block (+ a fraction)
        off[:, 2 * k] = ((ys // 4) * 4 + 1.3) - (ys - 1 + i)
        off[:, 2 * k + 1] = ((xs // 4) * 4 + 1.6) - (xs - 1 + j)
    off[1, :, 8:] += 4.0                                   # second image: half the rows become far samples
    a = (3, 3, 1, 1, 1, 1, 1, 1, 1)
    ref = oracle_dcn.dcn_v2_backward(x, w, b, off, m, gy, *a)
    got = _ext.dcn_v2_backward(x.to(cuda), w.to(cuda), b.to(cuda), off.to(cuda), m.to(cuda), gy.to(cuda), *a)
    for name, g_, r_ in zip(("grad_input", "grad_offset", "grad_mask", "grad_weight", "grad_bias"), got, ref):
        close(g_, r_, 5e-5, "convergent " + name)


def test_backward_is_deterministic_without_fallback(cuda, monkeypatch):
    """Three-pass backward (inverse lists): with offsets inside the search radius nothing is scattered atomically, so
    grad_input is bit-reproducible (the reference's col2im is not)."""
    from dcd_amd import _ext
    monkeypatch.setenv("DCD_BWD_SWEEP", "0")
    x, w, b, off, m, gy = (t.to(cuda) for t in make_case(2, 64, 64, 24, 40, off_scale=0.25, seed=5))
    off.clamp_(-0.9, 0.9)      # |offset| < 1 px: a cell collects at most 9 samples of one tap <= list capacity 10
    a = (3, 3, 1, 1, 1, 1, 1, 1, 1)
    g1 = _ext.dcn_v2_backward(x, w, b, off, m, gy, *a)[0]
    g2 = _ext.dcn_v2_backward(x, w, b, off, m, gy, *a)[0]
    assert torch.equal(g1, g2)


def test_one_pass_backward_reproducibility(cuda):
    """One-pass backward (dcn_bwd_sweep.inc, round 3): grad_offset / grad_mask are sums of per-chunk planes in a fixed order
    -> bit-reproducible; grad_input leaves the LDS windows through global atomics where strips overlap, like the reference's
    col2im (cuda/dcn_v2_im2col_cuda.cu:249) -> reproducible to summation-order noise only."""
    from dcd_amd import _ext
    x, w, b, off, m, gy = (t.to(cuda) for t in make_case(2, 64, 64, 24, 40, off_scale=0.25, seed=5))
    a = (3, 3, 1, 1, 1, 1, 1, 1, 1)
    r1 = _ext.dcn_v2_backward(x, w, b, off, m, gy, *a)
    r2 = _ext.dcn_v2_backward(x, w, b, off, m, gy, *a)
    assert torch.equal(r1[1], r2[1]) and torch.equal(r1[2], r2[2])
    close(r1[0], r2[0], 2e-6, "grad_input run to run")
    close(r1[3], r2[3], 2e-6, "grad_weight run to run")


def test_one_pass_backward_equals_three_pass(cuda, monkeypatch):
    """Same call through both backward paths (A/B switch DCD_BWD_SWEEP), incl. a ragged channel count, a partial strip,
    offsets that make quarter collisions (jumps of >= 2 px over 4 px) and some far samples."""
    from dcd_amd import _ext
    for (B, C, Co, H, W, osc, seed) in ((2, 64, 64, 24, 64, 0.5, 11), (1, 40, 50, 19, 44, 1.2, 12), (1, 128, 64, 16, 48, 2.0, 13)):
        x, w, b, off, m, gy = (t.to(cuda) for t in make_case(B, C, Co, H, W, off_scale=osc, seed=seed))
        off[:, :, :, 1::4] += 1.4          # neighbouring pixel groups pushed towards each other
        off[:, :, :, 3::4] -= 1.4
        a = (3, 3, 1, 1, 1, 1, 1, 1, 1)
        monkeypatch.setenv("DCD_BWD_SWEEP", "1")
        one = _ext.dcn_v2_backward(x, w, b, off, m, gy, *a)
        monkeypatch.setenv("DCD_BWD_SWEEP", "0")
        three = _ext.dcn_v2_backward(x, w, b, off, m, gy, *a)
        for name, g_, r_ in zip(("grad_input", "grad_offset", "grad_mask", "grad_weight", "grad_bias"), one, three):
            close(g_, r_, 2e-5, "one-pass vs three-pass %s %s" % (name, (B, C, Co, H, W)))


def test_one_pass_backward_keeps_far_heavy_calls_once_the_layer_is_known_as_near(cuda, oracle_dcn):
    """Host-side hand-over policy (dcn_v2.hip, `handover_decide`): after a layer's sampled call stayed far below its limit the
    five generic launches are dropped and the one-pass kernel takes EVERY sample of the following calls itself -- also of a call
    whose offsets would have been handed to the generic kernels.  Same weight tensor (the policy's key): two near calls (the
    second reads the first one's reported count), a far-heavy one against the oracle (still on the one-pass kernel alone), and
    the same again (now handed over)."""
    from dcd_amd import _ext
    for (B, C, Co, H, W, seed) in ((1, 32, 64, 16, 48, 21), (1, 32, 128, 12, 36, 22)):
        x, w, b, off, m, gy = make_case(B, C, Co, H, W, off_scale=0.3, seed=seed)
        dev = [t.to(cuda) for t in (x, w, b, off, m, gy)]
        a = (3, 3, 1, 1, 1, 1, 1, 1, 1)
        first = _ext.dcn_v2_backward(*dev, *a)
        torch.cuda.synchronize()
        again = _ext.dcn_v2_backward(*dev, *a)                       # policy: hand-over off from here on (unless the env pins it)
        for g_, r_ in zip(first, again):
            close(g_, r_, 2e-6, "hand-over on vs off, near offsets")
        g = torch.Generator().manual_seed(seed + 100)
        off_far = torch.randn(off.shape, generator=g) * 2.5            # ~23 % of the coordinates beyond 3 px
        # forward side of the policy: after a near backward the tiled forward runs without its far count and rescue launch and
        # takes every far sample in the kernel -- near offsets and (the report still says "near") far-heavy ones
        torch.cuda.synchronize()
        for o_ in (off, off_far):
            close(_ext.dcn_v2_forward(dev[0], dev[1], dev[2], o_.to(cuda), dev[4], *a).cpu(), oracle_dcn.dcn_v2_forward(x, w, b, o_, m, *a),
                  2e-5, "forward without rescue launch %s" % ((B, C, Co, H, W),))
        ref = oracle_dcn.dcn_v2_backward(x, w, b, off_far, m, gy, *a)
        got = _ext.dcn_v2_backward(dev[0], dev[1], dev[2], off_far.to(cuda), dev[4], dev[5], *a)
        for name, g_, r_ in zip(("grad_input", "grad_offset", "grad_mask", "grad_weight", "grad_bias"), got, ref):
            close(g_.cpu(), r_, 5e-5, "far-heavy call on the one-pass kernel alone: %s %s" % (name, (B, C, Co, H, W)))
        torch.cuda.synchronize()                                       # that call reported its count: the next one is handed over
        got = _ext.dcn_v2_backward(dev[0], dev[1], dev[2], off_far.to(cuda), dev[4], dev[5], *a)
        for name, g_, r_ in zip(("grad_input", "grad_offset", "grad_mask", "grad_weight", "grad_bias"), got, ref):
            close(g_.cpu(), r_, 5e-5, "far-heavy call, hand-over armed again: %s %s" % (name, (B, C, Co, H, W)))


def test_far_dominated_wide_layers_take_the_column_buffer_path(cuda, oracle_dcn):
    """Round 6 (dcn_v2.hip, `handover_far_dominated`): a layer both the one-pass kernel and the column-buffer path take (Cin, Cout
    >= 128) whose PREVIOUS call reported more far coordinates than its hand-over limit runs the column-buffer backward (no
    `dcn_bwd_sweep` launch), reports again, and returns to the one-pass kernel once its offsets are small.  Same weight tensor
    (the policy's key) through four calls, each against the oracle; the pinned modes never take that route."""
    import ctypes
    from dcd_amd import _ext, _lib
    L = _lib.lib()
    B, C, Co, H, W = 1, 128, 128, 12, 36
    x, w, b, off, m, gy = make_case(B, C, Co, H, W, off_scale=0.3, seed=31)
    g = torch.Generator().manual_seed(131)
    off_far = torch.randn(off.shape, generator=g) * 2.5                   # ~23 % of the coordinates beyond 3 px
    dev = [t.to(cuda) for t in (x, w, b, off, m, gy)]
    a = (3, 3, 1, 1, 1, 1, 1, 1, 1)
    names = ("grad_input", "grad_offset", "grad_mask", "grad_weight", "grad_bias")

    def far_reported():
        torch.cuda.synchronize()
        far = ctypes.c_uint(0)
        assert L.dcd_dcn_v2_policy_state(dev[1].data_ptr(), ctypes.byref(far)) == 2
        return far.value

    def run(o, what):
        ref = oracle_dcn.dcn_v2_backward(x, w, b, o, m, gy, *a)
        got = _ext.dcn_v2_backward(dev[0], dev[1], dev[2], o.to(cuda), dev[4], dev[5], *a)
        for name, g_, r_ in zip(names, got, ref):
            close(g_.cpu(), r_, 5e-5, "%s: %s" % (what, name))

    limit = B * 18 * H * W // 8                                           # far_count_limit: wide outputs, fewer than four images
    try:
        L.dcd_dcn_v2_forget(dev[1].data_ptr())
        run(off_far, "first call (unknown layer: one-pass kernel, hand-over armed)")
        n_far = far_reported()
        assert n_far > limit, (n_far, limit)
        run(off_far, "second call (far-dominated: column buffer)")
        assert far_reported() == n_far                                    # the column-buffer call reported the same count
        run(off, "third call (still routed by the last report: column buffer, near offsets)")
        assert far_reported() == 0
        run(off, "fourth call (near again: one-pass kernel)")
        assert far_reported() == 0
        # the same route in the two bf16 precisions (the column-buffer path's GEMMs in split / one-product form): against the fp32
        # oracle at their own bars
        for prec, tol in (("bf16x3", 1e-4), ("bf16", 1e-2)):
            L.dcd_dcn_v2_forget(dev[1].data_ptr())
            ref = oracle_dcn.dcn_v2_backward(x, w, b, off_far, m, gy, *a)
            for call in ("armed", "column buffer"):
                got = _ext.dcn_v2_backward(dev[0], dev[1], dev[2], off_far.to(cuda), dev[4], dev[5], *a, precision=prec)
                for name, g_, r_ in zip(names, got, ref):
                    close(g_.cpu(), r_, tol, "%s, %s call: %s" % (prec, call, name))
                assert far_reported() == n_far
        L.dcd_dcn_v2_forget(dev[1].data_ptr())
        _ext.set_handover("never")                                        # pinned: the route is never taken, results unchanged
        run(off_far, "pinned never")
        _ext.set_handover("always")
        run(off_far, "pinned always")
    finally:
        _ext.set_handover(None)
        L.dcd_dcn_v2_forget(dev[1].data_ptr())


WIDE_SWEEP_CASES = [
    # B, C, Co, H, W, off_scale: one-pass backward with Cout > 64 (round 4): dcol over 2 / 4 blocks of 64 outputs, the masked
    # samples through the col buffer, grad_weight as one product
    (2, 32, 128, 12, 36, 0.5),       # two output blocks, partial strip (36 = 2 x 16 + 4)
    (1, 48, 200, 10, 40, 1.2),       # 200 outputs -> four blocks with zero-padded weights; some far samples
    (1, 16, 256, 9, 32, 0.7),        # four full blocks, one channel chunk, a map barely taller than the ring
    (2, 40, 72, 19, 44, 2.0),        # ragged channels (40 = 2.5 chunks), 72 outputs, many far samples + quarter collisions
]


@pytest.mark.parametrize("prec", ["f32", "bf16x3"])
@pytest.mark.parametrize("case", WIDE_SWEEP_CASES)
def test_one_pass_backward_wide_outputs(cuda, oracle_dcn, monkeypatch, case, prec):
    """the one-pass backward for Cout 128 / 256 (the layers rounds 2-3 ran through the column buffer).
    All five gradients against the oracle, and against the three-pass / dense path of the same library (DCD_SWEEP_WIDE=0)."""
    from dcd_amd import _ext
    B, C, Co, H, W, osc = case
    x, w, b, off, m, gy = make_case(B, C, Co, H, W, off_scale=osc, seed=31)
    off[:, :, :, 1::4] += 1.4              # neighbouring pixel groups pushed towards each other: quarter collisions
    off[:, :, :, 3::4] -= 1.4
    a = (3, 3, 1, 1, 1, 1, 1, 1, 1)
    ref = oracle_dcn.dcn_v2_backward(x, w, b, off, m, gy, *a)
    dev = [t.to(cuda) for t in (x, w, b, off, m, gy)]
    monkeypatch.setenv("DCD_SWEEP_WIDE", "1")
    got = _ext.dcn_v2_backward(*dev, *a, precision=prec)
    tol = 5e-5 if prec == "f32" else BF16X3_TOL
    for name, g_, r_ in zip(("grad_input", "grad_offset", "grad_mask", "grad_weight", "grad_bias"), got, ref):
        close(g_, r_, tol, "wide one-pass %s %s %s" % (prec, name, case))
    if prec == "f32":
        again = _ext.dcn_v2_backward(*dev, *a, precision=prec)
        assert torch.equal(got[1], again[1]) and torch.equal(got[2], again[2])      # per-chunk planes summed in a fixed order
        close(again[3], got[3], 2e-6, "grad_weight run to run")                     # product partials in a fixed order; far pass: atomics
        monkeypatch.setenv("DCD_SWEEP_WIDE", "0")                                   # rounds 2-3: dense / three-pass backward
        old = _ext.dcn_v2_backward(*dev, *a, precision=prec)
        for name, g_, r_ in zip(("grad_input", "grad_offset", "grad_mask", "grad_weight", "grad_bias"), got, old):
            close(g_, r_, 2e-5, "wide one-pass vs round-3 path %s %s" % (name, case))


def test_wide_outputs_take_the_one_pass_kernel(cuda, monkeypatch):
    """The A/B switch must switch something: with DCD_SWEEP_WIDE=0 a 128-output layer runs the round-3 paths, whose grad_input
    (inverse lists / dense col2im: plain stores, a different summation order) differs from the one-pass result in the low bits."""
    from dcd_amd import _ext
    x, w, b, off, m, gy = (t.to(cuda) for t in make_case(2, 128, 128, 16, 48, off_scale=0.5, seed=3))
    a = (3, 3, 1, 1, 1, 1, 1, 1, 1)
    monkeypatch.setenv("DCD_SWEEP_WIDE", "1")
    one = _ext.dcn_v2_backward(x, w, b, off, m, gy, *a)
    monkeypatch.setenv("DCD_SWEEP_WIDE", "0")              # read per call
    three = _ext.dcn_v2_backward(x, w, b, off, m, gy, *a)
    for name, g_, r_ in zip(("grad_input", "grad_offset", "grad_mask", "grad_weight", "grad_bias"), one, three):
        close(g_, r_, 2e-5, "wide one-pass vs round-3 path " + name)
    assert not torch.equal(one[0], three[0])


# the seven distinct DCN geometries of DLA-34 at 384x1280 (bench.py::DCN_LAYERS), at the BASELINE batch
DGDE_GEOMETRIES = [(512, 256, 12, 40), (256, 256, 24, 80), (256, 128, 24, 80), (128, 128, 48, 160), (128, 64, 48, 160),
                   (64, 64, 96, 320), (256, 64, 24, 80)]


@pytest.mark.parametrize("off_scale", [0.5, 2.0])
@pytest.mark.parametrize("geom", DGDE_GEOMETRIES)
def test_full_size_layer_properties(cuda, oracle_dcn, geom, off_scale):
    """PARITY (image 0 of 8 against the oracle: 2e-5 forward / 5e-5 gradients) + PROPERTY (additivity of the batch for the other seven).
    BASELINE size (every DGDE layer geometry, bs 8): size-independent properties instead of a full oracle run.
    The kernel dispatch depends on the size (tiled kernels need H >= 16, W >= 32; Cin <= 64 / Cout <= 64 pick other
    variants; the deep layers run the generic kernels and the far-only passes), so each geometry is its own case, at
    sub-pixel offsets (0.5 px: everything from the staged windows) and at 2 px (far samples, list fallbacks).
    (1) linearity in the weights, (2) image 0 of the batch equals the oracle, forward and all five gradients,
    (3) <dY, dcn(x)> adjoint identity between forward and grad_weight/grad_bias, (4) the batched run's per-image
    gradients equal the batch-1 run's (images are independent; grad_weight sums over them)."""
    from dcd_amd import _ext
    C, Co, H, W = geom
    B = 8
    x, w, b, off, m, gy = make_case(B, C, Co, H, W, seed=7, off_scale=off_scale)
    xd, wd, bd, od, md, gd = (t.to(cuda) for t in (x, w, b, off, m, gy))
    a = (3, 3, 1, 1, 1, 1, 1, 1, 1)
    y = _ext.dcn_v2_forward(xd, wd, bd, od, md, *a)
    y2 = _ext.dcn_v2_forward(xd, 2 * wd, 2 * bd, od, md, *a)
    close(y2, 2 * y, 1e-5, "linearity in (W,b)")
    ref0 = oracle_dcn.dcn_v2_forward(x[:1], w, b, off[:1], m[:1], *a)
    close(y[:1], ref0, 2e-5, "image 0 vs oracle")
    grads = _ext.dcn_v2_backward(xd, wd, bd, od, md, gd, *a)
    # <dY, y> = <dW, W> + <db, b>  because y is linear in (W, b)
    lhs = (gd.double() * y.double()).sum().item()
    rhs = (grads[3].double() * wd.double()).sum().item() + (grads[4].double() * bd.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), 1.0), (lhs, rhs)
    refg = oracle_dcn.dcn_v2_backward(x[:1], w, b, off[:1], m[:1], gy[:1], *a)
    g1 = _ext.dcn_v2_backward(xd[:1].contiguous(), wd, bd, od[:1].contiguous(), md[:1].contiguous(),
                              gd[:1].contiguous(), *a)
    for name, g_, r_ in zip(("grad_input", "grad_offset", "grad_mask", "grad_weight", "grad_bias"), g1, refg):
        close(g_, r_, 1e-4, "full-size image 0 " + name)
    for i, name in enumerate(("grad_input", "grad_offset", "grad_mask")):
        close(grads[i][:1], g1[i], 2e-5, "batched vs single image " + name)
    # grad_weight of the batch = sum over images: check against the last image's own contribution added to the rest
    g7 = _ext.dcn_v2_backward(xd[7:].contiguous(), wd, bd, od[7:].contiguous(), md[7:].contiguous(), gd[7:].contiguous(), *a)
    g07 = _ext.dcn_v2_backward(xd[:7].contiguous(), wd, bd, od[:7].contiguous(), md[:7].contiguous(), gd[:7].contiguous(), *a)
    close(grads[3], g07[3] + g7[3], 2e-5, "grad_weight additivity over images")
    close(grads[4], g07[4] + g7[4], 2e-5, "grad_bias additivity over images")


# ---- dense path (dcd_amd/csrc/dcn_dense.inc: column buffer + three GEMMs).  By default it takes the layers with Cin >= 256
# (CASES has one, DGDE_GEOMETRIES four); DCD_DCN_DENSE=1 sends every geometry the alignment rules allow through it.
DENSE_CASES = [
    # B, C, Co, H, W, dg, off_scale, (kh, kw, sh, sw, ph, pw, dh, dw)
    (2, 16, 8, 6, 10, 1, 2.0, (3, 3, 1, 1, 1, 1, 1, 1)),
    (2, 64, 32, 9, 12, 2, 2.0, (3, 3, 1, 1, 1, 1, 1, 1)),      # two deformable groups
    (1, 32, 40, 8, 8, 1, 12.0, (3, 3, 1, 1, 1, 1, 1, 1)),      # most samples beyond the 8-px list radius: scattered atomically
    (2, 16, 136, 11, 16, 1, 1.5, (3, 3, 1, 1, 1, 1, 1, 1)),    # Cout over one GEMM tile, partial second tile
    (1, 48, 24, 23, 24, 1, 3.0, (3, 3, 1, 1, 1, 1, 1, 1)),     # 552 pixels: partial pixel tiles, split-K forward
    (2, 16, 8, 15, 17, 1, 2.0, (3, 3, 2, 2, 1, 1, 1, 1)),      # stride 2 -> 8 x 9 outputs
    (2, 16, 8, 12, 12, 1, 2.0, (3, 3, 1, 1, 2, 2, 2, 2)),      # dilation 2
    (1, 32, 8, 6, 10, 1, 1.0, (1, 1, 1, 1, 0, 0, 1, 1)),       # 1x1 kernel
]


@pytest.mark.parametrize("case", DENSE_CASES)
def test_dense_path_matches_oracle(cuda, oracle_dcn, monkeypatch, case):
    from dcd_amd import _ext
    monkeypatch.setenv("DCD_DCN_DENSE", "1")
    B, C, Co, H, W, dg, osc, geo = case
    kh, kw, sh, sw, ph, pw, dh, dw = geo
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B, C, H, W, generator=g)
    off = torch.randn(B, 2 * dg * kh * kw, Ho, Wo, generator=g) * osc
    m = torch.rand(B, dg * kh * kw, Ho, Wo, generator=g)
    w = torch.randn(Co, C, kh, kw, generator=g) / (C * kh * kw) ** 0.5
    b = torch.randn(Co, generator=g)
    gy = torch.randn(B, Co, Ho, Wo, generator=g)
    args = geo + (dg,)
    dev = [t.to(cuda) for t in (x, w, b, off, m, gy)]
    ref = oracle_dcn.dcn_v2_forward(x, w, b, off, m, *args)
    got = _ext.dcn_v2_forward(*dev[:5], *args)
    close(got, ref, 2e-5, "dense forward %s" % (case,))
    refg = oracle_dcn.dcn_v2_backward(x, w, b, off, m, gy, *args)
    gotg = _ext.dcn_v2_backward(*dev, *args)
    for name, g_, r_ in zip(("grad_input", "grad_offset", "grad_mask", "grad_weight", "grad_bias"), gotg, refg):
        close(g_, r_, 5e-5, "dense %s %s" % (name, case))
    # and it is a different code path from the fused kernels, agreeing with them to summation-order noise
    monkeypatch.setenv("DCD_DCN_DENSE", "0")
    fused = _ext.dcn_v2_forward(*dev[:5], *args)
    close(got, fused, 2e-5, "dense vs fused forward %s" % (case,))


def test_dense_path_overflow_and_far(cuda, oracle_dcn, monkeypatch):
    """The convergent-offset case above through the dense path: overflowed lists and far samples are scattered by the
    coordinate kernel, the rest comes from the lists."""
    from dcd_amd import _ext
    monkeypatch.setenv("DCD_DCN_DENSE", "1")
    B, C, Co, H, W = 2, 16, 8, 16, 20
    x, w, b, off, m, gy = make_case(B, C, Co, H, W, seed=11)
    ys = torch.arange(H).view(1, 1, H, 1).float()
    xs = torch.arange(W).view(1, 1, 1, W).float()
    for k in range(9):
        i, j = k // 3, k % 3
        off[:, 2 * k] = ((ys // 4) * 4 + 1.3) - (ys - 1 + i)
        off[:, 2 * k + 1] = ((xs // 4) * 4 + 1.6) - (xs - 1 + j)
    off[1, :, 8:] += 9.0                                   # second image: half the rows beyond the list radius
    a = (3, 3, 1, 1, 1, 1, 1, 1, 1)
    ref = oracle_dcn.dcn_v2_backward(x, w, b, off, m, gy, *a)
    got = _ext.dcn_v2_backward(x.to(cuda), w.to(cuda), b.to(cuda), off.to(cuda), m.to(cuda), gy.to(cuda), *a)
    for name, g_, r_ in zip(("grad_input", "grad_offset", "grad_mask", "grad_weight", "grad_bias"), got, ref):
        close(g_, r_, 5e-5, "dense convergent " + name)


def test_dense_backward_is_deterministic(cuda, monkeypatch):
    """No atomics when the lists hold every sample: all five gradients are bit-reproducible (split-K partials are summed
    in a fixed order)."""
    from dcd_amd import _ext
    monkeypatch.setenv("DCD_DCN_DENSE", "1")
    x, w, b, off, m, gy = (t.to(cuda) for t in make_case(2, 64, 64, 24, 40, off_scale=0.25, seed=5))
    off.clamp_(-0.9, 0.9)
    a = (3, 3, 1, 1, 1, 1, 1, 1, 1)
    g1 = _ext.dcn_v2_backward(x, w, b, off, m, gy, *a)
    g2 = _ext.dcn_v2_backward(x, w, b, off, m, gy, *a)
    for u, v in zip(g1[:4], g2[:4]):
        assert torch.equal(u, v)


@pytest.mark.parametrize("jump", [3.0, 6.0, 7.5])
@pytest.mark.parametrize("dense", ["0", "1"])
def test_local_search_radius_of_the_inverse_lists(cuda, oracle_dcn, monkeypatch, jump, dense):
    """dcn_build_inverse searches as far as the offsets in the 3x3 neighbourhood of 8x8 blocks around a cell require
    (dcn_offset_blockmax), not as far as the call's largest offset.  A smooth field (0.2 px) with one 8x8 patch whose samples
    jump `jump` px to the right: the cells that receive them lie in blocks with small offsets of their own and must still find
    them (7.5 px: the local bound saturates and the call-wide radius is used)."""
    from dcd_amd import _ext
    monkeypatch.setenv("DCD_DCN_DENSE", dense)
    monkeypatch.setenv("DCD_BWD_SWEEP", "0")               # the inverse lists belong to the three-pass / dense backward
    B, C, Co, H, W = 2, 16, 8, 40, 40
    x, w, b, off, m, gy = make_case(B, C, Co, H, W, off_scale=0.2, seed=13)
    off[:, 1::2, 16:24, 16:24] += jump                     # x offsets of every tap inside the patch
    off[1, 0::2, 24:32, 8:16] -= jump                      # second image: another patch jumping up
    a = (3, 3, 1, 1, 1, 1, 1, 1, 1)
    ref = oracle_dcn.dcn_v2_backward(x, w, b, off, m, gy, *a)
    got = _ext.dcn_v2_backward(x.to(cuda), w.to(cuda), b.to(cuda), off.to(cuda), m.to(cuda), gy.to(cuda), *a)
    for name, g_, r_ in zip(("grad_input", "grad_offset", "grad_mask", "grad_weight", "grad_bias"), got, ref):
        close(g_, r_, 5e-5, "local radius %s jump %s dense %s" % (name, jump, dense))


@pytest.mark.parametrize("shape", [(2, 27, 12, 20), (3, 54, 7, 9), (8, 27, 96, 320)])
def test_offset_mask_glue_matches_stock_ops(cuda, shape):
    """DCN.forward's chunk / cat / sigmoid between conv_offset_mask and the deformable convolution as one kernel each way
    (dcn_offset_mask_split / _merge) against the stock slicing + sigmoid and their autograd."""
    from dcd_amd.model.backbone.DCNv2.dcn_v2 import _OffsetMask
    g = torch.Generator().manual_seed(17)
    out = (torch.randn(shape, generator=g) * 2).to(cuda)
    T = shape[1] // 3
    a = out.clone().requires_grad_(True)
    off_a, m_a = _OffsetMask.apply(a)
    b = out.clone().requires_grad_(True)
    off_b, m_b = b[:, :2 * T], torch.sigmoid(b[:, 2 * T:])
    assert off_a.is_contiguous() and torch.equal(off_a, off_b)
    assert (m_a - m_b).abs().max().item() <= 2e-7
    w1 = torch.randn(off_a.shape, generator=g).to(cuda)
    w2 = torch.randn(m_a.shape, generator=g).to(cuda)
    ((off_a * w1).sum() + (m_a * w2).sum()).backward()
    ((off_b * w1).sum() + (m_b * w2).sum()).backward()
    assert (a.grad - b.grad).abs().max().item() <= 1e-6 * b.grad.abs().max().item()


def test_errors_raise(cuda):
    from dcd_amd import _ext
    x, w, b, off, m, gy = (t.to(cuda) for t in make_case(1, 4, 4, 5, 5))
    with pytest.raises(RuntimeError):
        _ext.dcn_v2_forward(x, w[:, :3].contiguous(), b, off, m, 3, 3, 1, 1, 1, 1, 1, 1, 1)
    with pytest.raises(RuntimeError):
        _ext.dcn_v2_forward(x, w, b, off, m, 5, 5, 1, 1, 1, 1, 1, 1, 1)
    with pytest.raises(RuntimeError):
        _ext.dcn_v2_forward(x.cpu(), w, b, off, m, 3, 3, 1, 1, 1, 1, 1, 1, 1)
    with pytest.raises(RuntimeError):
        _ext.dcn_v2_backward(x.transpose(2, 3), w, b, off, m, gy, 3, 3, 1, 1, 1, 1, 1, 1, 1)


@pytest.mark.parametrize("B,C,Co,H,W", [(2, 64, 64, 24, 80), (1, 128, 64, 12, 40), (2, 64, 72, 14, 44)])
def test_dcn_module_as_one_node_equals_three_nodes(cuda, monkeypatch, B, C, Co, H, W):
    """`DCN.forward` as one autograd node (offset conv + split + deformable conv, the offset conv's input gradient accumulated
    into the deformable conv's inside the Winograd kernel) against the three separate nodes: same output, same gradients."""
    from dcd_amd.model.backbone.DCNv2 import dcn_v2
    torch.manual_seed(C + H)
    m = dcn_v2.DCN(C, Co, (3, 3), 1, 1).to(cuda)
    m.conv_offset_mask.weight.data.normal_(0, 0.02)
    m.conv_offset_mask.bias.data.normal_(0, 0.3)
    x = torch.randn(B, C, H, W, device=cuda)
    gy = torch.randn(B, Co, H, W, device=cuda)
    res = []
    for one in (True, False):
        monkeypatch.setattr(dcn_v2, "_ONE_NODE", one)
        xi = x.clone().requires_grad_()
        m.zero_grad()
        y = m(xi)
        assert (type(y.grad_fn).__name__ == "_DCNWithOffsetsBackward") == one
        y.backward(gy)
        res.append([y.detach(), xi.grad] + [p.grad.clone() for p in (m.weight, m.bias, m.conv_offset_mask.weight, m.conv_offset_mask.bias)])
    for a, b, what in zip(res[0], res[1], ("output", "grad_input", "grad_weight", "grad_bias", "grad_offset_weight", "grad_offset_bias")):
        scale = max(b.abs().max().item(), 1e-6)
        assert (a - b).abs().max().item() <= 2e-6 * scale, what


def test_launch_policy_is_forgotten_with_its_weight(cuda):
    """VERDICT r4 item 8 / ADVICE r4: the per-layer hand-over policy is keyed by the weight's device address.  A layer that has run
    backward calls has a report; once its module is destroyed (or moved) the key is dropped, so a NEW layer whose weight the caching
    allocator places at the SAME address starts as "unknown" (no entry) instead of inheriting the old layer's far history."""
    import ctypes
    import gc
    from dcd_amd import _lib
    from dcd_amd.model.backbone.DCNv2.dcn_v2 import DCN
    L = _lib.lib()

    def state(ptr):
        far = ctypes.c_uint(0)
        st = L.dcd_dcn_v2_policy_state(ptr, ctypes.byref(far))
        return st, far.value

    torch.cuda.synchronize()
    layer = DCN(64, 64, (3, 3), 1, 1).to(cuda)
    ptr = layer.weight.data_ptr()
    assert state(ptr)[0] == 0
    x = torch.randn(2, 64, 24, 64, device=cuda, requires_grad=True)
    for _ in range(2):
        layer(x).sum().backward()
    torch.cuda.synchronize()
    st, far = state(ptr)
    assert st == 2 and far == 0, (st, far)              # zero-initialised offset conv: no far samples, reported
    del layer
    gc.collect()
    assert state(ptr)[0] == 0, "the destroyed layer's policy entry survived"
    # same size, same allocator pool: the new weight lands on the freed block
    other = DCN(64, 64, (3, 3), 1, 1).to(cuda)
    if other.weight.data_ptr() == ptr:
        assert state(ptr)[0] == 0
        other(x).sum().backward()
        torch.cuda.synchronize()
        assert state(ptr)[0] == 2
    # a module whose parameters move forgets the old address
    moved = other.weight.data_ptr()
    other.double().float()
    assert state(moved)[0] == 0
    assert L.dcd_dcn_v2_forget(None) == 0
