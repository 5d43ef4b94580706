"""Winograd MFMA 3x3 convolution (csrc/conv.hip) against torch's conv2d evaluated in fp64 on the same inputs.
Tolerance 2e-5 of the output scale: F(2x2,3x3) in fp32 (the same algorithm class as the stock MIOpen solver)."""
import pytest
import torch
from torch.nn import functional as F

pytestmark = pytest.mark.gpu

CASES = [  # B, Cin, Cout, H, W
    (2, 64, 64, 16, 32),
    (1, 64, 256, 24, 64),      # head trunk shape, scaled down
    (2, 72, 80, 10, 36),       # ragged: partial regions, partial channel chunk / output slice
    (1, 256, 64, 8, 32),       # small map, few regions: the contraction is split, partial images summed in fixed order
    (2, 128, 64, 24, 80),      # 12 x 20 px regions (30 of 32 lanes live), exact cover
    (1, 512, 72, 12, 40),      # one region row, 8-way split contraction, partial output slice
    (2, 64, 64, 14, 44),       # 12 x 20 regions with ragged rows and columns
    (1, 128, 128, 48, 160),
    (3, 64, 64, 96, 176),      # 5.5 column strips, many splits of the weight-gradient tiles
]


def _close(a, ref, what, tol=2e-5):
    err = (a.double() - ref).abs().max().item()
    scale = max(ref.abs().max().item(), 1e-6)
    assert err <= tol * scale, "%s: max abs err %.3e vs scale %.3e" % (what, err, scale)


@pytest.mark.parametrize("B,C,K,H,W", CASES)
def test_conv3x3_forward_and_input_grad(cuda, B, C, K, H, W):
    from dcd_amd import ops
    g = torch.Generator().manual_seed(C * 7 + K)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(K, C, 3, 3, generator=g) / (C * 9) ** 0.5
    gy = torch.randn(B, K, H, W, generator=g)
    xd, wd = x.double().requires_grad_(), w.double().requires_grad_()
    ref = F.conv2d(xd, wd, padding=1)
    ref.backward(gy.double())
    xg, wg = x.to(cuda).requires_grad_(), w.to(cuda).requires_grad_()
    y = ops.conv3x3(xg, wg)
    y.backward(gy.to(cuda))
    _close(y.detach().cpu(), ref.detach(), "forward")
    _close(xg.grad.cpu(), xd.grad, "grad_input")
    _close(wg.grad.cpu(), wd.grad, "grad_weight", 2e-5)


@pytest.mark.parametrize("geom", ["0", "1"])
@pytest.mark.parametrize("B,C,K,H,W", [(2, 64, 64, 14, 44), (1, 72, 80, 24, 80), (1, 64, 64, 26, 36)])
def test_conv3x3_region_shapes(cuda, monkeypatch, geom, B, C, K, H, W):
    """Both region shapes of the forward / input-gradient kernel (8 x 32 and 12 x 20 px) on maps neither divides."""
    from dcd_amd import ops
    monkeypatch.setenv("DCD_CONV_GEOM", geom)
    g = torch.Generator().manual_seed(H * 100 + W)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(K, C, 3, 3, generator=g) / (C * 9) ** 0.5
    gy = torch.randn(B, K, H, W, generator=g)
    xd, wd = x.double().requires_grad_(), w.double()
    ref = F.conv2d(xd, wd, padding=1)
    ref.backward(gy.double())
    y = ops._conv3x3_call(x.to(cuda), w.to(cuda), K, False)
    gx = ops._conv3x3_call(gy.to(cuda), w.to(cuda), C, True)
    _close(y.cpu(), ref.detach(), "forward")
    _close(gx.cpu(), xd.grad, "grad_input")


def test_conv3x3_prepared_weights_equal_per_call_transform(cuda):
    """dcd_conv3x3_transform_weights (both directions, one launch) + dcd_conv3x3_prepared against dcd_conv3x3: bitwise equal."""
    from dcd_amd import ops
    g = torch.Generator(device=cuda).manual_seed(11)
    for B, C, K, H, W in ((2, 64, 96, 24, 80), (1, 128, 27, 12, 40), (1, 256, 64, 8, 32)):
        x = torch.randn(B, C, H, W, device=cuda, generator=g)
        w = torch.randn(K, C, 3, 3, device=cuda, generator=g) / (C * 9) ** 0.5
        gy = torch.randn(B, K, H, W, device=cuda, generator=g)
        tf, tb = ops.conv3x3_transform_weights(w)
        assert torch.equal(ops._conv3x3_call(x, w, K, False, transformed=tf), ops._conv3x3_call(x, w, K, False))
        assert torch.equal(ops._conv3x3_call(gy, w, C, True, transformed=tb), ops._conv3x3_call(gy, w, C, True))
        only_b = ops.conv3x3_transform_weights(w, forward=False)[1]
        assert torch.equal(only_b, tb)


def test_prepared_weight_table_one_launch_for_all_layers(cuda):
    """refresh_conv_weights: every registered layer's Winograd-domain weights from ONE launch, bitwise equal to the per-layer
    transform; an entry is used only while the weight is untouched since the refresh."""
    from dcd_amd import ops
    g = torch.Generator(device=cuda).manual_seed(3)
    shapes = [(64, 64), (128, 64), (72, 80), (256, 512), (64, 27)]
    ws = [torch.nn.Parameter(torch.randn(k, c, 3, 3, device=cuda, generator=g) / (c * 9) ** 0.5) for c, k in shapes]
    x = [torch.randn(1, c, 24, 80, device=cuda, generator=g, requires_grad=True) for c, _ in shapes]
    prepared = ops._PREPARED.setdefault((cuda.index if cuda.index is not None else torch.cuda.current_device(), False), ops._PreparedWeights())
    first = [ops.conv3x3(xi, wi) for xi, wi in zip(x, ws)]              # registers the five layers (each transforms its own copy)
    assert all(prepared.lookup(wi) is None for wi in ws)
    ops.refresh_conv_weights()
    for (c, k), wi in zip(shapes, ws):
        e = prepared.lookup(wi)
        assert e is not None
        tf, tb = ops.conv3x3_transform_weights(wi)
        assert torch.equal(e[2], tf) and torch.equal(e[3], tb)
    second = [ops.conv3x3(xi, wi) for xi, wi in zip(x, ws)]             # served from the table
    for a, b in zip(first, second):
        assert torch.equal(a, b)
    (second[0].sum() + second[3].sum()).backward()                      # backward-data from the table's other half
    assert torch.isfinite(x[0].grad).all() and torch.isfinite(x[3].grad).all()
    with torch.no_grad():
        ws[1].mul_(2.0)                                                 # an in-place write: the entry is stale until the next refresh
    assert prepared.lookup(ws[1]) is None and prepared.lookup(ws[0]) is not None
    y = ops.conv3x3(x[1], ws[1])
    assert torch.allclose(y, 2.0 * first[1], rtol=1e-5, atol=1e-6)
    ops.refresh_conv_weights()
    assert prepared.lookup(ws[1]) is not None
    assert torch.equal(ops.conv3x3(x[1], ws[1]), y)
    # the bf16 precision modes have tables of their own -- split-bf16 layout (bf16x3, maps of at least 7 680 px) and the direct
    # one-product kernel's (bf16): one launch, bitwise equal to the per-layer transform
    from dcd_amd import _ext
    idx = cuda.index if cuda.index is not None else torch.cuda.current_device()
    tables = []
    for mode, key, xs in (("bf16x3", True, [torch.randn(1, c, 96, 80, device=cuda, generator=g, requires_grad=True) for c, _ in shapes]),
                          ("bf16", "bf16" if ops._BF16_DIRECT else True, x)):
        with _ext.precision_scope(mode):
            split = ops._PREPARED.setdefault((idx, key), ops._PreparedWeights(key))
            tables.append(split)
            a = [ops.conv3x3(xi, wi) for xi, wi in zip(xs, ws)]             # registers the layers in the mode's table
            assert all(split.lookup(wi) is None for wi in ws)
            ops.refresh_conv_weights()
            for wi in ws:
                e = split.lookup(wi)
                assert e is not None
                tf, tb = ops.conv3x3_transform_weights(wi, like=xs[0])
                assert isinstance(tf, ops.Bf16Weights) == (key == "bf16")
                assert torch.equal(e[2], tf.tensor) and torch.equal(e[3], tb.tensor)
            b = [ops.conv3x3(xi, wi) for xi, wi in zip(xs, ws)]             # served from the table
            for u, v in zip(a, b):
                assert torch.equal(u, v)
    # entries nobody looks up any more leave their table after three refreshes (the fp32 ones were last used above)
    for _ in range(4):
        ops.refresh_conv_weights()
    assert not prepared.entries and not any(t.entries for t in tables)


def test_conv_with_skip_sums_both_gradients_in_the_kernel(cuda):
    """ops.conv3x3_with_skip: (conv(x), x) as one node; d/dx = input gradient of the convolution + the skip's gradient, added in the
    kernel's output transform -- against the two-consumer form (autograd's own addition), incl. a split contraction (small map)."""
    from dcd_amd import ops
    g = torch.Generator(device=cuda).manual_seed(17)
    for B, C, H, W in ((2, 64, 24, 80), (1, 256, 8, 32)):
        x = torch.randn(B, C, H, W, device=cuda, generator=g)
        w = torch.randn(C, C, 3, 3, device=cuda, generator=g) / (C * 9) ** 0.5
        gy, gs = torch.randn(B, C, H, W, device=cuda, generator=g), torch.randn(B, C, H, W, device=cuda, generator=g)
        xa, wa = x.clone().requires_grad_(), w.clone().requires_grad_()
        y, skip = ops.conv3x3_with_skip(xa, wa)
        gs_before = gs.clone()
        torch.autograd.backward([y, skip], [gy, gs])
        assert torch.equal(gs, gs_before)                     # the skip's gradient is read, not accumulated into
        xb, wb = x.clone().requires_grad_(), w.clone().requires_grad_()
        yb = ops.conv3x3(xb, wb)
        torch.autograd.backward([yb, xb * 1.0], [gy, gs])
        assert torch.equal(y, yb)
        _close(xa.grad, xb.grad.double(), "grad_input with skip", 1e-6)
        assert torch.equal(wa.grad, wb.grad)
        xc = x.clone().requires_grad_()
        ops.conv3x3_with_skip(xc, w)[1].backward(gs)           # only the skip used
        assert torch.equal(xc.grad, gs)


def test_conv_module_dispatch(cuda):
    from dcd_amd.model.layers.conv import Conv2d
    conv = Conv2d(64, 64, 3, padding=1, bias=False).to(cuda)
    x = torch.randn(1, 64, 48, 160, device=cuda)
    from dcd_amd import ops
    assert ops.conv3x3_supported(x, conv.weight)
    y = conv(x)
    ref = F.conv2d(x, conv.weight, padding=1)
    _close(y.detach().cpu(), ref.detach().double().cpu(), "module forward", 1e-4)     # vs the stock fp32 solver
    strided = Conv2d(64, 64, 3, stride=2, padding=1, bias=False).to(cuda)
    assert strided(x).shape == (1, 64, 24, 80)                                           # falls back to the stock op


def test_conv3x3_bad_arguments(cuda):
    from dcd_amd import _lib
    L = _lib.lib()
    x = torch.randn(1, 64, 9, 30, device=cuda)
    assert L.dcd_conv3x3(_lib.stream_of(x), x.data_ptr(), x.data_ptr(), None, None, x.data_ptr(), 1, 64, 9, 30, 64, 0, x.data_ptr(), 1 << 30) == 1
    x2 = torch.randn(1, 64, 8, 32, device=cuda)
    assert L.dcd_conv3x3(_lib.stream_of(x2), x2.data_ptr(), x2.data_ptr(), None, None, x2.data_ptr(), 1, 64, 8, 32, 64, 0, x2.data_ptr(), 16) == 2
    assert L.dcd_conv3x3(_lib.stream_of(x2), x2.data_ptr(), x2.data_ptr(), x2.data_ptr(), None, x2.data_ptr(), 1, 64, 8, 32, 64, 1, x2.data_ptr(), 1 << 30) == 1      # bias with backward_data


@pytest.mark.parametrize("B,C,H,W,f", [(2, 8, 5, 6, 2), (1, 64, 24, 80, 4), (2, 16, 12, 40, 2), (1, 3, 4, 2, 8)])
def test_depthwise_upsample_matches_conv_transpose(cuda, B, C, H, W, f):
    """IDAUp's depthwise ConvTranspose2d (csrc/upsample.hip) against torch's conv_transpose2d in fp64."""
    from dcd_amd import ops
    g = torch.Generator().manual_seed(C + f)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(C, 1, 2 * f, 2 * f, generator=g)
    gy = torch.randn(B, C, H * f, W * f, generator=g)
    xd, wd = x.double().requires_grad_(), w.double().requires_grad_()
    ref = F.conv_transpose2d(xd, wd, stride=f, padding=f // 2, groups=C)
    ref.backward(gy.double())
    xg, wg = x.to(cuda).requires_grad_(), w.to(cuda).requires_grad_()
    y = ops.upsample_dw(xg, wg, f)
    y.backward(gy.to(cuda))
    _close(y.detach().cpu(), ref.detach(), "forward", 1e-6)
    _close(xg.grad.cpu(), xd.grad, "grad_input", 1e-5)
    _close(wg.grad.cpu(), wd.grad, "grad_weight", 2e-5)


@pytest.mark.parametrize("B,C,H,W,f", [(2, 8, 5, 6, 2), (1, 64, 24, 80, 4)])
def test_depthwise_upsample_with_skip_sum(cuda, B, C, H, W, f):
    """IDAUp's `node(up(x) + skip)`: the sum formed inside the up-sampling kernel (DepthwiseUpsample.forward_add) against
    conv_transpose2d + add in fp64, all three gradients (the skip's is the incoming gradient itself)."""
    from dcd_amd.model.layers.conv import DepthwiseUpsample
    g = torch.Generator().manual_seed(C * f)
    up = DepthwiseUpsample(C, C, 2 * f, stride=f, padding=f // 2, output_padding=0, groups=C, bias=False)
    with torch.no_grad():
        up.weight.copy_(torch.randn(up.weight.shape, generator=g))
    x = torch.randn(B, C, H, W, generator=g)
    skip = torch.randn(B, C, H * f, W * f, generator=g)
    gy = torch.randn(B, C, H * f, W * f, generator=g)
    xd, sd, wd = x.double().requires_grad_(), skip.double().requires_grad_(), up.weight.detach().double().requires_grad_()
    ref = F.conv_transpose2d(xd, wd, stride=f, padding=f // 2, groups=C) + sd
    ref.backward(gy.double())
    up = up.to(cuda)
    xg, sg = x.to(cuda).requires_grad_(), skip.to(cuda).requires_grad_()
    y = up.forward_add(xg, sg)
    y.backward(gy.to(cuda))
    _close(y.detach().cpu(), ref.detach(), "forward", 1e-6)
    _close(xg.grad.cpu(), xd.grad, "grad_input", 1e-5)
    _close(sg.grad.cpu(), sd.grad, "grad_skip", 1e-7)
    _close(up.weight.grad.cpu(), wd.grad, "grad_weight", 2e-5)


@pytest.mark.parametrize("shape", [(2, 3, 4, 8), (1, 16, 96, 320), (2, 5, 6, 12)])
def test_maxpool2x2_matches_stock(cuda, shape):
    """csrc/upsample.hip max pooling (arg-max re-derived in the backward) against F.max_pool2d: values and gradient exact, ties
    (first maximum wins) and a NaN included."""
    from dcd_amd.model.layers.conv import MaxPool2x2
    g = torch.Generator().manual_seed(11)
    x = torch.randn(shape, generator=g)
    x[0, 0, 0:2, 0:2] = 1.5                                  # a four-way tie
    x[-1, -1, 2:4, 4:6] = torch.tensor([[0.1, float("nan")], [0.3, 0.2]])
    gy = torch.randn(shape[0], shape[1], shape[2] // 2, shape[3] // 2, generator=g)
    a = x.clone().to(cuda).requires_grad_()
    b = x.clone().to(cuda).requires_grad_()
    ya = MaxPool2x2(2, stride=2)(a)
    yb = F.max_pool2d(b, 2, 2)
    assert torch.equal(torch.nan_to_num(ya, nan=7.0), torch.nan_to_num(yb, nan=7.0))
    ya.backward(gy.to(cuda))
    yb.backward(gy.to(cuda))
    assert torch.equal(a.grad, b.grad)


def test_depthwise_upsample_module_dispatch(cuda):
    from dcd_amd.model.layers.conv import DepthwiseUpsample
    up = DepthwiseUpsample(16, 16, 4, stride=2, padding=1, output_padding=0, groups=16, bias=False).to(cuda)
    x = torch.randn(2, 16, 12, 40, device=cuda)
    ref = F.conv_transpose2d(x, up.weight, stride=2, padding=1, groups=16)
    _close(up(x).detach().cpu(), ref.detach().double().cpu(), "module forward", 1e-5)


def test_fan_out_sums_gradients_in_one_pass(cuda):
    from dcd_amd import ops
    x = torch.randn(2, 5, 7, 9, device=cuda, requires_grad=True)          # 630 elements: exercises the non-multiple-of-4 tail
    outs = ops.fan_out(x, 12)
    ws = [torch.randn_like(x) for _ in outs]
    sum((o * w).sum() for o, w in zip(outs, ws)).backward()
    ref = torch.stack(ws).double().sum(0)
    assert (x.grad.double() - ref).abs().max().item() < 1e-5
    from dcd_amd import _lib
    import ctypes
    base = torch.randn(3, 1003, device=cuda)
    srcs = [base[i, 1:1001] for i in range(3)]              # contiguous rows starting 4 bytes past a 16-byte boundary
    outp = torch.empty(1000, device=cuda)
    arr = (ctypes.c_void_p * 3)(*[t.data_ptr() for t in srcs])
    assert _lib.lib().dcd_sum_tensors(_lib.stream_of(outp), arr, 3, outp.data_ptr(), 1000) == 0
    assert torch.allclose(outp, srcs[0] + srcs[1] + srcs[2], atol=1e-6)
    y = torch.randn(3, 4, device=cuda, requires_grad=True)
    a, b_, c = ops.fan_out(y, 3)
    (a.sum() * 2 + c.sum()).backward()                                    # one consumer unused
    assert torch.allclose(y.grad, torch.full_like(y, 3.0))


def test_conv_bias_gradient(cuda):
    from dcd_amd.model.layers.conv import Conv2d
    conv = Conv2d(32, 27, 3, padding=1, bias=True).to(cuda)
    x = torch.randn(2, 32, 96, 352, device=cuda, requires_grad=True)
    gy = torch.randn(2, 27, 96, 352, device=cuda)
    conv(x).backward(gy)
    got = (conv.bias.grad.clone(), conv.weight.grad.clone(), x.grad.clone())
    conv.zero_grad(); x.grad = None
    F.conv2d(x, conv.weight, conv.bias, padding=1).backward(gy)
    _close(got[0].cpu(), conv.bias.grad.double().cpu(), "bias grad", 1e-5)
    _close(got[1].cpu(), conv.weight.grad.double().cpu(), "weight grad", 1e-5)
    _close(got[2].cpu(), x.grad.double().cpu(), "input grad", 1e-5)


@pytest.mark.parametrize("B,C,H,W", [(2, 64, 48, 160), (1, 128, 24, 80), (2, 256, 12, 40)])
def test_offset_conv_backward_on_our_kernels(cuda, B, C, H, W):
    """`conv_offset_mask` (Cin -> 27, 3x3, bias): forward (bias in the output transform), input gradient and weight gradient on
    the one-output-block variants of csrc/conv.hip; against fp64."""
    from dcd_amd.model.layers.conv import Conv2d
    g = torch.Generator().manual_seed(C + H)
    conv = Conv2d(C, 27, 3, padding=1, bias=True)
    conv.weight.data = torch.randn(27, C, 3, 3, generator=g) / (C * 9) ** 0.5
    conv.bias.data = torch.randn(27, generator=g)
    x = torch.randn(B, C, H, W, generator=g)
    gy = torch.randn(B, 27, H, W, generator=g)
    xd = x.double().requires_grad_()
    wd, bd = conv.weight.detach().double().requires_grad_(), conv.bias.detach().double().requires_grad_()
    ref = F.conv2d(xd, wd, bd, padding=1)
    ref.backward(gy.double())
    conv = conv.to(cuda)
    xg = x.to(cuda).requires_grad_()
    y = conv(xg)
    assert type(y.grad_fn).__name__ == "_ConvBiasBackward"
    y.backward(gy.to(cuda))
    _close(y.detach().cpu(), ref.detach(), "forward")
    _close(xg.grad.cpu(), xd.grad, "grad_input")
    _close(conv.weight.grad.cpu(), wd.grad, "grad_weight", 2e-5)
    _close(conv.bias.grad.cpu(), bd.grad, "grad_bias", 1e-5)


@pytest.mark.parametrize("B,Ci,k,H,W", [(2, 16, 3, 40, 320), (1, 3, 7, 48, 384), (2, 16, 3, 19, 260), (1, 3, 7, 21, 196)])
def test_stem_convolutions(cuda, B, Ci, k, H, W):
    """csrc/stem.hip (16x16x4 MFMA direct conv) against torch's conv2d in fp64: forward, input gradient (16->16) and weight
    gradient; the ragged cases have partial tiles in both directions."""
    from dcd_amd import ops
    g = torch.Generator().manual_seed(Ci * 100 + k)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(16, Ci, k, k, generator=g) / (Ci * k * k) ** 0.5
    gy = torch.randn(B, 16, H, W, generator=g)
    xd, wd = x.double().requires_grad_(Ci == 16), w.double().requires_grad_()
    ref = F.conv2d(xd, wd, padding=k // 2)
    ref.backward(gy.double())
    xg, wg = x.to(cuda).requires_grad_(Ci == 16), w.to(cuda).requires_grad_()
    y = ops.conv_stem(xg, wg)
    y.backward(gy.to(cuda))
    _close(y.detach().cpu(), ref.detach(), "forward", 1e-5)
    if Ci == 16:
        _close(xg.grad.cpu(), xd.grad, "grad_input", 1e-5)
    _close(wg.grad.cpu(), wd.grad, "grad_weight", 2e-5)


def test_stem_dispatch(cuda):
    from dcd_amd import ops
    from dcd_amd.model.layers.conv import Conv2d
    conv = Conv2d(16, 16, 3, padding=1, bias=False).to(cuda)
    x = torch.randn(1, 16, 128, 512, device=cuda, requires_grad=True)
    assert ops.conv_stem_supported(x, conv.weight, conv.stride, conv.padding, conv.dilation, conv.groups)
    y = conv(x)
    assert type(y.grad_fn).__name__ == "_ConvStemBackward"
    base = Conv2d(3, 16, 7, padding=3, bias=False).to(cuda)
    img = torch.randn(1, 3, 128, 512, device=cuda)
    assert type(base(img).grad_fn).__name__ == "_ConvStemBackward"
    assert type(base(img.requires_grad_()).grad_fn).__name__ != "_ConvStemBackward"      # no input gradient for the 7x7: stock op


def test_small_map_conv_runs_on_our_kernels(cuda):
    """24x80 / 12x40 maps (DLA levels 4 and 5): forward, input and weight gradient all on csrc/conv.hip."""
    from dcd_amd.model.layers.conv import Conv2d
    conv = Conv2d(64, 96, 3, padding=1, bias=False).to(cuda)
    x = torch.randn(2, 64, 24, 80, device=cuda, requires_grad=True)
    gy = torch.randn(2, 96, 24, 80, device=cuda)
    y = conv(x)
    assert type(y.grad_fn).__name__ == "_Conv3x3Backward"
    y.backward(gy)
    xd, wd = x.detach().double().cpu().requires_grad_(), conv.weight.detach().double().cpu().requires_grad_()
    ref = F.conv2d(xd, wd, padding=1)
    ref.backward(gy.double().cpu())
    _close(y.detach().cpu(), ref.detach(), "forward")
    _close(conv.weight.grad.cpu(), wd.grad, "grad_weight", 2e-5)
    _close(x.grad.cpu(), xd.grad, "grad_input")
    again = conv(x)
    assert torch.equal(again, y), "split contraction must be reproducible (partials summed in a fixed order)"


@pytest.mark.parametrize("C,K,H,W", [(64, 256, 96, 320), (128, 128, 48, 160), (256, 256, 24, 80), (512, 512, 12, 40)])
def test_conv3x3_full_size_against_stock_solver(cuda, C, K, H, W):
    """BASELINE size (bs 8): our three Winograd kernels against the stock fp32 solver on the same device tensors.  Both are
    fp32 algorithms of the same class, so the comparison is relative to the result's scale; plus linearity of the weight
    gradient in dY (a size-independent property: wrw(x, a*g1 + g2) = a*wrw(x, g1) + wrw(x, g2))."""
    from dcd_amd import ops
    g = torch.Generator(device=cuda).manual_seed(C + K)
    x = torch.randn(8, C, H, W, device=cuda, generator=g)
    w = torch.randn(K, C, 3, 3, device=cuda, generator=g) / (C * 9) ** 0.5
    gy = torch.randn(8, K, H, W, device=cuda, generator=g)
    y = ops._conv3x3_call(x, w, K, False)
    gx = ops._conv3x3_call(gy, w, C, True)
    gw = ops._conv3x3_wrw_call(x, gy, w.shape)
    ry = F.conv2d(x, w, padding=1)
    rgx, rgw, _ = torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, True, False])
    for a, r, what in ((y, ry, "forward"), (gx, rgx, "grad_input"), (gw, rgw, "grad_weight")):
        err = (a - r).abs().max().item()
        assert err <= 1e-4 * r.abs().max().item(), "%s: %.3e vs scale %.3e" % (what, err, r.abs().max().item())
    g2 = torch.randn(8, K, H, W, device=cuda, generator=g)
    lin = ops._conv3x3_wrw_call(x, 0.5 * gy + g2, w.shape)
    ref = 0.5 * gw + ops._conv3x3_wrw_call(x, g2, w.shape)
    assert (lin - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    again = ops._conv3x3_wrw_call(x, gy, w.shape)
    assert torch.equal(again, gw), "weight gradient must be bitwise reproducible (fixed-order partial sums)"


def test_stem_full_size_against_stock_solver(cuda):
    from dcd_amd import ops
    g = torch.Generator(device=cuda).manual_seed(5)
    for Ci, k in ((3, 7), (16, 3)):
        x = torch.randn(8, Ci, 384, 1280, device=cuda, generator=g)
        w = torch.randn(16, Ci, k, k, device=cuda, generator=g) / (Ci * k * k) ** 0.5
        gy = torch.randn(8, 16, 384, 1280, device=cuda, generator=g)
        xg, wg = x.clone().requires_grad_(Ci == 16), w.clone().requires_grad_()
        y = ops.conv_stem(xg, wg)
        y.backward(gy)
        xr, wr = x.clone().requires_grad_(Ci == 16), w.clone().requires_grad_()
        r = F.conv2d(xr, wr, padding=k // 2)
        r.backward(gy)
        pairs = [(y.detach(), r.detach(), "forward"), (wg.grad, wr.grad, "grad_weight")]
        if Ci == 16:
            pairs.append((xg.grad, xr.grad, "grad_input"))
        for a, b, what in pairs:
            err = (a - b).abs().max().item()
            assert err <= 1e-4 * b.abs().max().item(), "%dx%d %s: %.3e vs scale %.3e" % (k, k, what, err, b.abs().max().item())


def test_conv1x1_of_cat_matches_cat_then_conv(cuda):
    """ops.conv1x1_of_cat (DLA Root without the concatenation) against conv2d(cat(...)) in fp64: output, every input gradient,
    the weight gradient; one input that needs no gradient."""
    from dcd_amd import ops
    g = torch.Generator().manual_seed(31)
    B, H, W, O = 2, 12, 20, 48
    chans = (16, 32, 8)
    xs = [torch.randn(B, c, H, W, generator=g) for c in chans]
    w = torch.randn(O, sum(chans), 1, 1, generator=g) * 0.1
    gy = torch.randn(B, O, H, W, generator=g)
    xd = [x.double().requires_grad_(i != 2) for i, x in enumerate(xs)]
    wd = w.double().requires_grad_()
    ref = F.conv2d(torch.cat(xd, 1), wd)
    ref.backward(gy.double())
    xg = [x.to(cuda).requires_grad_(i != 2) for i, x in enumerate(xs)]
    wg = w.to(cuda).requires_grad_()
    y = ops.conv1x1_of_cat(xg, wg)
    y.backward(gy.to(cuda))
    _close(y.detach().cpu(), ref.detach(), "forward", 2e-6)
    for i in (0, 1):
        _close(xg[i].grad.cpu(), xd[i].grad, "grad_input %d" % i, 1e-5)
    assert xg[2].grad is None
    _close(wg.grad.cpu(), wd.grad, "grad_weight", 2e-5)


@pytest.mark.parametrize("B,C,K,H,W", CASES + [(2, 64, 27, 24, 64), (1, 128, 27, 12, 40)])
def test_conv3x3_split_bf16_form(cuda, monkeypatch, B, C, K, H, W):
    """`_ext.set_precision("bf16x3")`: forward and input gradient on wino_conv3x3_split (Winograd-domain products as
    hi*hi + hi*lo + lo*hi on the bf16 matrix cores) against conv2d in fp64 at 1e-4 of the output scale (north_star's bound is 1e-3;
    the split loses ~2^-16 per product), differs from the exact-fp32 kernel in the low bits (it is not the same kernel), with bias and
    with an accumulated residual; the weight gradient stays on the fp32 kernel."""
    from dcd_amd import _ext, ops
    monkeypatch.setattr(ops, "_CONV_SPLIT_MIN_MAP", 0)          # the product uses this kernel from 48 x 160 maps on; here on every size
    g = torch.Generator().manual_seed(C * 5 + K)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(K, C, 3, 3, generator=g) / (C * 9) ** 0.5
    bias = torch.randn(K, generator=g)
    res = torch.randn(B, K, H, W, generator=g)
    gy = torch.randn(B, K, H, W, generator=g)
    ref = F.conv2d(x.double(), w.double(), bias.double(), padding=1) + res.double()
    ref_gx = torch.nn.grad.conv2d_input(x.shape, w.double(), gy.double(), padding=1)
    xd, wd, bd, gd = x.to(cuda), w.to(cuda), bias.to(cuda), gy.to(cuda)
    y32 = ops._conv3x3_call(xd, wd, K, False, bd, residual=res.to(cuda).clone())
    _ext.set_precision("bf16x3")
    try:
        tf, tb = ops.conv3x3_transform_weights(wd)
        assert isinstance(tf, ops.SplitWeights) and isinstance(tb, ops.SplitWeights)
        y = ops._conv3x3_call(xd, wd, K, False, bd, residual=res.to(cuda).clone(), transformed=tf)
        gx = ops._conv3x3_call(gd, wd, C, True, transformed=tb)
        y_unprepared = ops._conv3x3_call(xd, wd, K, False, bd, residual=res.to(cuda).clone())
    finally:
        _ext.set_precision("f32")
    _close(y.cpu(), ref, "split forward", 1e-4)
    _close(gx.cpu(), ref_gx, "split grad_input", 1e-4)
    assert torch.equal(y, y_unprepared)
    assert not torch.equal(y, y32)
    assert (y - y32).abs().max().item() <= 5e-5 * y32.abs().max().item()


@pytest.mark.parametrize("B,C,K,H,W", CASES + [(2, 64, 27, 24, 64), (1, 128, 27, 12, 40), (8, 64, 64, 96, 320)])
def test_conv3x3_mixed_bf16_form(cuda, monkeypatch, B, C, K, H, W):
    """`_ext.precision_scope("bf16")` (MODEL.FP16): forward and input gradient on conv3x3_direct_bf16 and the weight gradient on
    conv3x3_wrw_direct_bf16 (Cout > 32; wino_wrw3x3_f32<1, true> below) -- ONE product of bf16-rounded operands on the bf16 matrix
    cores, fp32 accumulate -- against conv2d in fp64 (DCD_CONV_BF16_DIRECT=0 / DCD_CONV_WRW_DIRECT=0 run the Winograd-domain forms
    wino_conv3x3_split<., 1> / wino_wrw3x3_f32<2, true> through the same test).  Operand rounding is 2^-9 each (in the Winograd
    forms on TRANSFORMED values, sums of up to four inputs / nine weights): held to 1.5e-2 of the output scale, required to be above 1e-4 (it is not the fp32 kernel), with bias and residual; the
    autograd node remembers the scope for its backward (which runs outside it)."""
    from dcd_amd import _ext, ops
    g = torch.Generator().manual_seed(C * 7 + K)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(K, C, 3, 3, generator=g) / (C * 9) ** 0.5
    bias = torch.randn(K, generator=g)
    res = torch.randn(B, K, H, W, generator=g)
    gy = torch.randn(B, K, H, W, generator=g)
    ref = F.conv2d(x.double(), w.double(), bias.double(), padding=1) + res.double()
    ref_gx = torch.nn.grad.conv2d_input(x.shape, w.double(), gy.double(), padding=1)
    ref_gw = torch.nn.grad.conv2d_weight(x.double(), w.shape, gy.double(), padding=1)
    xd, wd, bd, gd = x.to(cuda), w.to(cuda), bias.to(cuda), gy.to(cuda)
    y32 = ops._conv3x3_call(xd, wd, K, False, bd, residual=res.to(cuda).clone())
    with _ext.precision_scope("bf16"):
        tf, tb = ops.conv3x3_transform_weights(wd)
        assert isinstance(tf, ops.SplitWeights) and tf.prec == ops.PREC_BF16
        y = ops._conv3x3_call(xd, wd, K, False, bd, residual=res.to(cuda).clone(), transformed=tf)
        gx = ops._conv3x3_call(gd, wd, C, True, transformed=tb)
        y_unprepared = ops._conv3x3_call(xd, wd, K, False, bd, residual=res.to(cuda).clone())
    gw = ops._conv3x3_wrw_call(xd, gd, w.shape, ops.PREC_BF16)
    gw32 = ops._conv3x3_wrw_call(xd, gd, w.shape)
    _close(y.cpu(), ref, "bf16 forward", 1.5e-2)
    _close(gx.cpu(), ref_gx, "bf16 grad_input", 1.5e-2)
    _close(gw.cpu(), ref_gw, "bf16 grad_weight", 1.5e-2)
    _close(gw32.cpu(), ref_gw, "fp32 grad_weight", 2e-5)
    assert torch.equal(y, y_unprepared)
    for a_, b_ in ((y, y32), (gw, gw32)):
        assert (a_ - b_).abs().max().item() > 1e-4 * b_.abs().max().item()
    if K >= 64 and C >= 64:
        # through the autograd node: scope at forward time only
        xg, wg = xd.clone().requires_grad_(True), wd.clone().requires_grad_(True)
        with _ext.precision_scope("bf16"):
            out = ops.conv3x3(xg, wg)
        assert _ext.get_precision() == "f32"
        out.backward(gd)
        _close(xg.grad.cpu(), ref_gx, "bf16 autograd grad_input", 1.5e-2)
        _close(wg.grad.cpu(), ref_gw, "bf16 autograd grad_weight", 1.5e-2)
        assert (wg.grad - gw32).abs().max().item() > 1e-4 * gw32.abs().max().item()     # the backward kept the forward's precision


@pytest.mark.parametrize("B,C,K,H,W", [(2, 64, 64, 16, 32), (2, 72, 80, 10, 36), (1, 512, 72, 12, 40), (1, 256, 64, 9, 32), (3, 27, 64, 14, 44),
                                       (1, 64, 27, 24, 64), (1, 16, 16, 96, 128), (2, 128, 256, 24, 80)])
def test_conv3x3_direct_bf16_is_the_exact_product_of_rounded_operands(cuda, B, C, K, H, W):
    """The direct one-product kernels through the C ABI (dcd_conv3x3_bf16_*, dcd_conv3x3_wrw with DCD_PREC_BF16): they round inputs
    and weights to bf16 ONCE and accumulate in fp32, so against conv2d in fp64 of the SAME bf16-rounded operands only the fp32
    accumulation is left -- 2e-5 of the output scale (the 1.5e-2 of the test above is the rounding itself).  Ragged channel counts
    (72, 27: the last chunk reads past the tensor and must get zeros), odd heights, partial regions, split contractions (small
    maps, one image), bias / residual; and the entry points' argument checks."""
    from dcd_amd import _lib, ops
    L = _lib.lib()
    g = torch.Generator().manual_seed(C * 11 + K)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(K, C, 3, 3, generator=g) / (C * 9) ** 0.5
    bias = torch.randn(K, generator=g)
    res = torch.randn(B, K, H, W, generator=g)
    gy = torch.randn(B, K, H, W, generator=g)
    r = lambda t: t.bfloat16().double()
    ref = F.conv2d(r(x), r(w), bias.double(), padding=1) + res.double()
    ref_gx = torch.nn.grad.conv2d_input(x.shape, r(w), r(gy), padding=1)
    ref_gw = torch.nn.grad.conv2d_weight(r(x), w.shape, r(gy), padding=1)
    xd, wd, bd, gd = x.to(cuda), w.to(cuda), bias.to(cuda), gy.to(cuda)
    st = _lib.stream_of(xd)
    tf = torch.empty(L.dcd_conv3x3_bf16_weights_bytes(C, K, 0) // 4, dtype=torch.int32, device=cuda)
    tb = torch.empty(L.dcd_conv3x3_bf16_weights_bytes(C, K, 1) // 4, dtype=torch.int32, device=cuda)
    assert L.dcd_conv3x3_bf16_transform_weights(st, wd.data_ptr(), C, K, tf.data_ptr(), tb.data_ptr()) == 0
    n = L.dcd_conv3x3_bf16_workspace_bytes(B, C, H, W, K)
    ws = torch.empty(max(n, 16), dtype=torch.uint8, device=cuda)
    y = res.to(cuda).clone()
    assert L.dcd_conv3x3_bf16_prepared(st, xd.data_ptr(), tf.data_ptr(), bd.data_ptr(), y.data_ptr(), y.data_ptr(), B, C, H, W, K, 0,
                                       ws.data_ptr(), n) == 0
    gx = torch.empty_like(xd)
    assert L.dcd_conv3x3_bf16_prepared(st, gd.data_ptr(), tb.data_ptr(), None, None, gx.data_ptr(), B, C, H, W, K, 1, ws.data_ptr(), n) == 0
    _close(y.cpu(), ref, "direct forward", 2e-5)
    _close(gx.cpu(), ref_gx, "direct grad_input", 2e-5)
    if W % 4 == 0 and H % 2 == 0:
        gw = ops._conv3x3_wrw_call(xd, gd, w.shape, ops.PREC_BF16)
        if K > 32:                                           # the direct weight gradient; <= 32 outputs run the Winograd-domain form
            _close(gw.cpu(), ref_gw, "direct grad_weight", 2e-5)
        else:
            _close(gw.cpu(), ref_gw, "grad_weight", 1.5e-2)
    # argument checks: W % 4, bias with backward_data, a null pointer, a workspace too small for a split contraction
    bad = torch.randn(1, 64, 8, 30, device=cuda)
    t64 = torch.empty(L.dcd_conv3x3_bf16_weights_bytes(64, 64, 0) // 4, dtype=torch.int32, device=cuda)
    assert L.dcd_conv3x3_bf16_prepared(st, bad.data_ptr(), t64.data_ptr(), None, None, bad.data_ptr(), 1, 64, 8, 30, 64, 0, ws.data_ptr(), n) == 1
    assert L.dcd_conv3x3_bf16_prepared(st, gd.data_ptr(), tb.data_ptr(), bd.data_ptr(), None, gx.data_ptr(), B, C, H, W, K, 1, ws.data_ptr(), n) == 1
    assert L.dcd_conv3x3_bf16_prepared(st, None, tb.data_ptr(), None, None, gx.data_ptr(), B, C, H, W, K, 1, ws.data_ptr(), n) == 1
    assert L.dcd_conv3x3_bf16_transform_weights(st, wd.data_ptr(), C, K, None, None) == 1
    if n > 16:                                               # at least one direction splits its contraction
        codes = {L.dcd_conv3x3_bf16_prepared(st, xd.data_ptr(), tf.data_ptr(), None, None, y.data_ptr(), B, C, H, W, K, 0, ws.data_ptr(), 16),
                 L.dcd_conv3x3_bf16_prepared(st, gd.data_ptr(), tb.data_ptr(), None, None, gx.data_ptr(), B, C, H, W, K, 1, ws.data_ptr(), 16)}
        assert 2 in codes and codes <= {0, 2}


@pytest.mark.parametrize("B,cs,O,H,W", [(2, (64, 64), 64, 24, 80), (1, (128, 128, 64, 128), 128, 12, 40), (2, (64,), 128, 16, 20),
                                        (1, (512, 512, 256), 512, 12, 40), (3, (16, 32), 48, 6, 10), (8, (64, 64), 64, 96, 320)])
def test_conv1x1_of_cat_in_the_bf16_scope(cuda, monkeypatch, B, cs, O, H, W):
    """ops.conv1x1_of_cat under `_ext.precision_scope("bf16")` (MODEL.FP16): forward, input gradients and weight gradient on
    csrc/conv1x1_bf16.inc -- the exact product of the bf16-rounded operands with fp32 accumulation: 2e-5 of the output scale against
    conv2d in fp64 of the SAME rounded operands (and within 1.5e-2 of the unrounded fp64 result); several inputs, a pixel count that
    is not a multiple of the 512-pixel workgroup tile or of the 64-pixel weight-gradient strip, 48 outputs (a partial 64-row slice);
    the backward keeps the forward's precision outside the scope."""
    from dcd_amd import _ext, ops
    monkeypatch.setattr(ops, "_PW_MIN_PIXELS", 0)              # the small launches too (the dispatch keeps them on the library)
    g = torch.Generator().manual_seed(sum(cs) + O)
    xs = [torch.randn(B, c, H, W, generator=g) for c in cs]
    C = sum(cs)
    w = torch.randn(O, C, 1, 1, generator=g) / C ** 0.5
    gy = torch.randn(B, O, H, W, generator=g)
    r = lambda t: t.bfloat16().double()
    xcat = torch.cat(xs, 1)
    ref = F.conv2d(r(xcat), r(w))
    ref_gx = torch.nn.grad.conv2d_input(xcat.shape, r(w), r(gy))
    ref_gw = torch.nn.grad.conv2d_weight(r(xcat), w.shape, r(gy))
    exact = F.conv2d(xcat.double(), w.double())
    xd = [x.to(cuda).requires_grad_(True) for x in xs]
    wd = w.to(cuda).requires_grad_(True)
    with _ext.precision_scope("bf16"):
        y = ops.conv1x1_of_cat(xd, wd)
    assert _ext.get_precision() == "f32"
    y.backward(gy.to(cuda))
    _close(y.detach().cpu(), ref, "1x1 forward", 2e-5)
    _close(y.detach().cpu(), exact, "1x1 forward vs unrounded", 1.5e-2)
    assert (y.detach().cpu().double() - exact).abs().max().item() > 1e-4 * exact.abs().max().item()      # it IS the bf16 form
    c0 = 0
    for x, c in zip(xd, cs):
        _close(x.grad.cpu(), ref_gx[:, c0:c0 + c], "1x1 grad_input", 2e-5)
        c0 += c
    _close(wd.grad.cpu(), ref_gw, "1x1 grad_weight", 2e-5)
    # the fp32 form of the same call (outside the scope) is untouched
    y32 = ops.conv1x1_of_cat([x.detach() for x in xd], wd.detach())
    _close(y32.cpu(), exact, "1x1 fp32 forward", 2e-5)


@pytest.mark.parametrize("B,cs,O,H,W", [(2, (64, 64), 64, 24, 80), (1, (128, 128, 64, 128), 128, 12, 40), (2, (64,), 128, 16, 20),
                                        (1, (512, 512, 256), 512, 12, 40), (3, (16, 32), 48, 6, 10), (2, (32,), 64, 96, 320),
                                        (8, (64, 64), 64, 96, 320),
                                        (2, (80, 16), 96, 10, 14),      # ragged everywhere: 96 = 1.5 blocks of 64 inputs, 140 pixels
                                        (1, (16,), 16, 2, 2)])          # one chunk, one group of four pixels
def test_conv1x1_of_cat_on_the_fp32_pointwise_kernels(cuda, monkeypatch, B, cs, O, H, W):
    """ops.conv1x1_of_cat in exact fp32 on csrc/conv1x1_f32.inc (round 6: forward, input gradients and weight gradient as own
    kernels instead of batched library GEMMs) against conv2d(cat(...)) in fp64 at 2e-5 of the output scale (the bar of the other
    fp32 convolution kernels): several inputs, a pixel count that is not a multiple of the 512-pixel workgroup tile or of the
    64-pixel weight-gradient strip, 48 outputs (a partial block of 64), an input that needs no gradient; the library path
    (DCD_CONV1X1_F32=0) gives the same numbers."""
    from dcd_amd import _lib, ops
    monkeypatch.setattr(ops, "_PW_F32_MIN_PIXELS", 0)          # the small launches too (the dispatch keeps them on the library)
    monkeypatch.setattr(ops, "_PW_F32_MAX_WEIGHTS", 1 << 30)   # ... and the wide ones
    g = torch.Generator().manual_seed(sum(cs) + O + 1)
    xs = [torch.randn(B, c, H, W, generator=g) for c in cs]
    C = sum(cs)
    w = torch.randn(O, C, 1, 1, generator=g) / C ** 0.5
    gy = torch.randn(B, O, H, W, generator=g)
    xcat = torch.cat(xs, 1).double()
    ref = F.conv2d(xcat, w.double())
    ref_gx = torch.nn.grad.conv2d_input(xcat.shape, w.double(), gy.double())
    ref_gw = torch.nn.grad.conv2d_weight(xcat, w.shape, gy.double())
    last = len(cs) - 1
    seen = []
    L = _lib.lib()
    for name in ("dcd_conv1x1_f32", "dcd_conv1x1_wrw_f32"):
        fn = getattr(L, name)
        def spy(*a, _fn=fn, _name=name):
            seen.append(_name)
            return _fn(*a)
        monkeypatch.setattr(L, name, spy, raising=False)
    xd = [x.to(cuda).requires_grad_(i != last or last == 0) for i, x in enumerate(xs)]
    wd = w.to(cuda).requires_grad_(True)
    y = ops.conv1x1_of_cat(xd, wd)
    y.backward(gy.to(cuda))
    assert "dcd_conv1x1_f32" in seen and "dcd_conv1x1_wrw_f32" in seen      # the own kernels ran, not the library
    _close(y.detach().cpu(), ref, "1x1 fp32 forward", 2e-5)
    c0 = 0
    for i, (x, c) in enumerate(zip(xd, cs)):
        if x.requires_grad:
            _close(x.grad.cpu(), ref_gx[:, c0:c0 + c], "1x1 fp32 grad_input %d" % i, 2e-5)
        else:
            assert x.grad is None
        c0 += c
    _close(wd.grad.cpu(), ref_gw, "1x1 fp32 grad_weight", 2e-5)
    monkeypatch.setattr(ops, "_PW_F32", False)
    y_lib = ops.conv1x1_of_cat([x.detach() for x in xd], wd.detach())
    _close(y_lib.cpu(), ref, "1x1 library forward", 2e-5)


def test_conv1x1_fp32_argument_checks(cuda):
    """dcd_conv1x1_f32 / dcd_conv1x1_wrw_f32 refuse what they cannot run: H W % 4, channel counts that are not multiples of 16, a
    null pointer, more than four inputs, a workspace that is too small."""
    import ctypes
    from dcd_amd import _lib
    L = _lib.lib()
    x = torch.randn(1, 16, 4, 8, device=cuda)
    w = torch.randn(16, 16, device=cuda)
    o = torch.empty(1, 16, 4, 8, device=cuda)
    st = _lib.stream_of(x)
    one = lambda ch, n=1: ((ctypes.c_void_p * n)(*[x.data_ptr()] * n), (ctypes.c_int * n)(*[ch] * n))
    p, c = one(16)
    assert L.dcd_conv1x1_f32(st, w.data_ptr(), 16, 0, 1, p, c, o.data_ptr(), 1, 16, 32) == 0
    assert L.dcd_conv1x1_f32(st, w.data_ptr(), 16, 0, 1, p, c, o.data_ptr(), 1, 16, 30) == 1
    p8, c8 = one(8)
    assert L.dcd_conv1x1_f32(st, w.data_ptr(), 16, 0, 1, p8, c8, o.data_ptr(), 1, 16, 32) == 1
    assert L.dcd_conv1x1_f32(st, None, 16, 0, 1, p, c, o.data_ptr(), 1, 16, 32) == 1
    p5, c5 = one(16, 5)
    assert L.dcd_conv1x1_f32(st, w.data_ptr(), 16, 0, 5, p5, c5, o.data_ptr(), 1, 16, 32) == 1
    n = L.dcd_conv1x1_wrw_f32_workspace_bytes(1, 16, 16, 32)
    ws = torch.empty(n, dtype=torch.uint8, device=cuda)
    gw = torch.empty(16, 16, device=cuda)
    assert L.dcd_conv1x1_wrw_f32(st, o.data_ptr(), x.data_ptr(), gw.data_ptr(), 16, 1, 16, 16, 32, ws.data_ptr(), n) == 0
    assert L.dcd_conv1x1_wrw_f32(st, o.data_ptr(), x.data_ptr(), gw.data_ptr(), 16, 1, 16, 16, 32, ws.data_ptr(), n - 1) == 2
    assert L.dcd_conv1x1_wrw_f32(st, o.data_ptr(), x.data_ptr(), gw.data_ptr(), 8, 1, 16, 16, 32, ws.data_ptr(), n) == 1


@pytest.mark.parametrize("B,C,K,H,W", [(2, 16, 32, 48, 80), (1, 32, 64, 24, 80), (2, 64, 128, 24, 80), (1, 128, 256, 24, 80),
                                       (1, 256, 512, 12, 40), (3, 16, 48, 8, 16), (8, 16, 32, 384, 1280), (8, 64, 128, 96, 320)])
def test_stride2_conv_native_fp32_kernels(cuda, monkeypatch, B, C, K, H, W):
    """csrc/conv_s2_f32.inc (round 6): the stride-2 / pad-1 3x3 convolutions of the DLA levels in exact fp32 -- forward, input gradient
    (both row phases, both forms: four and two pixels per lane) and weight gradient against conv2d in fp64 at 2e-5 of each
    result's scale: DLA's five channel pairs, 48 outputs (a partial block of 32), one-chunk layers, the full-size first two levels;
    the module dispatch takes them in exact fp32 and inside the split-bf16 scope (same results), not inside the bf16 scope, and the library refuses the shapes the kernels cannot run."""
    from dcd_amd import _ext, _lib, ops
    from dcd_amd.model.layers.conv import Conv2d
    monkeypatch.setattr(ops, "_S2_NATIVE_MIN_PIXELS", 0)
    g = torch.Generator().manual_seed(C * 3 + K)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(K, C, 3, 3, generator=g) / (C * 9) ** 0.5
    gy = torch.randn(B, K, H // 2, W // 2, generator=g)
    nb = min(B, 2)                                              # fp64 reference of the activations on two images (the full-size cases)
    ref = F.conv2d(x[:nb].double(), w.double(), None, 2, 1)
    ref_gx = torch.nn.grad.conv2d_input(x[:nb].shape, w.double(), gy[:nb].double(), stride=2, padding=1)
    ref_gw = torch.nn.grad.conv2d_weight(x.to(cuda).double(), w.shape, gy.to(cuda).double(), stride=2, padding=1).cpu()
    assert ops.conv3x3_stride2_native_supported(x.to(cuda), w.to(cuda))
    xd, wd = x.to(cuda).requires_grad_(True), w.to(cuda).requires_grad_(True)
    y = ops.conv3x3_stride2_native(xd, wd)
    y.backward(gy.to(cuda))
    _close(y.detach()[:nb].cpu(), ref, "native stride-2 forward", 2e-5)
    _close(xd.grad[:nb].cpu(), ref_gx, "native stride-2 grad_input", 2e-5)
    _close(wd.grad.cpu(), ref_gw, "native stride-2 grad_weight", 2e-5)
    conv = Conv2d(C, K, 3, stride=2, padding=1, bias=False).to(cuda)
    calls = []
    real = ops.conv3x3_stride2_native
    monkeypatch.setattr(ops, "conv3x3_stride2_native", lambda a, b: (calls.append(1), real(a, b))[1])
    out = conv(x.to(cuda))
    assert calls and (out - F.conv2d(x.to(cuda), conv.weight, None, 2, 1)).abs().max().item() <= 2e-5 * out.abs().max().item()
    del calls[:]
    with _ext.precision_scope("bf16x3"):                         # a scope permits reduced products, it does not oblige them: the exact kernels
        out3 = conv(x.to(cuda))                                  # are the faster ones for these layers (ops.conv3x3_stride2_native_supported)
    assert calls and torch.equal(out3, out), "inside the split-bf16 scope the same exact-fp32 kernels run"
    del calls[:]
    with _ext.precision_scope("bf16"):
        conv(x.to(cuda))
    assert not calls, "the bf16 scope keeps the space-to-depth form on the bf16 kernels"
    L = _lib.lib()
    st = _lib.stream_of(xd)
    bad = torch.randn(1, 16, 6, 12, device=cuda)                 # W % 8 != 0
    assert L.dcd_conv3x3_s2_f32(st, bad.data_ptr(), wd.data_ptr(), bad.data_ptr(), 1, 16, 6, 12, 16) == 1
    assert L.dcd_conv3x3_s2_f32(st, xd.data_ptr(), wd.data_ptr(), None, B, C, H, W, K) == 1
    assert L.dcd_conv3x3_s2_f32(st, xd.data_ptr(), wd.data_ptr(), y.data_ptr(), B, 24, H, W, K) == 1          # Cin % 16
    n = L.dcd_conv3x3_s2_f32_wrw_workspace_bytes(B, C, H, W, K)
    ws = torch.empty(n, dtype=torch.uint8, device=cuda)
    assert L.dcd_conv3x3_s2_f32_wrw(st, xd.data_ptr(), y.data_ptr(), wd.grad.data_ptr(), B, C, H, W, K, ws.data_ptr(), n - 1) == 2


@pytest.mark.parametrize("B,C,K,H,W", [(2, 16, 32, 48, 80), (1, 32, 64, 24, 80), (2, 64, 128, 24, 80), (1, 128, 256, 24, 80),
                                       (2, 256, 512, 6, 20), (1, 64, 128, 13, 38)])      # off H % 4 / W % 8: the zero-padded form
def test_stride2_conv_through_space_to_depth(cuda, monkeypatch, B, C, K, H, W):
    """ops.conv3x3_stride2: the stride-2 / pad-1 3x3 convolution of the DLA levels as a stride-1 convolution of the pixel-unshuffled
    input with the regrouped filter, on our Winograd kernels -- output and both gradients against conv2d in fp64: exact-fp32 form
    at 2e-5, one-product bf16 form (the mode that routes there by default) at 1.5e-2; the module dispatch takes it inside a bf16
    precision scope and keeps the stock solver in fp32."""
    from dcd_amd import _ext, ops
    from dcd_amd.model.layers.conv import Conv2d
    g = torch.Generator().manual_seed(C + K)
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(K, C, 3, 3, generator=g) / (C * 9) ** 0.5
    gy = torch.randn(B, K, (H + 1) // 2, (W + 1) // 2, generator=g)
    ref = F.conv2d(x.double(), w.double(), None, 2, 1)
    ref_gx = torch.nn.grad.conv2d_input(x.shape, w.double(), gy.double(), stride=2, padding=1)
    ref_gw = torch.nn.grad.conv2d_weight(x.double(), w.shape, gy.double(), stride=2, padding=1)
    for prec, tol in (("f32", 2e-5), ("bf16", 1.5e-2)):
        xd, wd = x.to(cuda).requires_grad_(True), w.to(cuda).requires_grad_(True)
        with _ext.precision_scope(prec):
            y = ops.conv3x3_stride2(xd, wd)
        y.backward(gy.to(cuda))
        _close(y.detach().cpu(), ref, "s2d %s forward" % prec, tol)
        _close(xd.grad.cpu(), ref_gx, "s2d %s grad_input" % prec, tol)
        _close(wd.grad.cpu(), ref_gw, "s2d %s grad_weight" % prec, tol)
    conv = Conv2d(C, K, 3, stride=2, padding=1, bias=False).to(cuda)
    xd = x.to(cuda)
    calls = []
    real = ops.conv3x3_stride2
    monkeypatch.setattr(ops, "conv3x3_stride2", lambda a, b: (calls.append(1), real(a, b))[1])
    conv(xd)
    assert not calls, "exact fp32 keeps the stock solver for the stride-2 layers"
    aligned = H % 4 == 0 and W % 8 == 0 and (H // 2) * (W // 2) >= ops._CONV_MIN_MAP
    with _ext.precision_scope("bf16"):
        out = conv(xd)
    assert bool(calls) == aligned                        # the bf16 scope takes the own kernels for the shapes that pay
    assert (out - F.conv2d(xd, conv.weight, None, 2, 1)).abs().max().item() <= 1.5e-2 * out.abs().max().item()
    # a process that captures whole-step graphs (ops.stride2_on_own_kernels, round 6): EVERY size in exact fp32 too, and a stride-2
    # layer that still reached the stock solver inside a capture would raise instead of recording a possible memset node
    del calls[:]
    monkeypatch.setattr(ops, "_S2D_MODE", "1")
    out = conv(xd)
    assert calls and (out - F.conv2d(xd, conv.weight, None, 2, 1)).abs().max().item() <= 2e-5 * out.abs().max().item()
    narrow = Conv2d(8, 16, 3, stride=2, padding=1, bias=False).to(cuda)      # fewer than 16 input channels: not taken
    xs = torch.randn(1, 8, 16, 32, device=cuda)
    narrow(xs)                                                                # eager: stock op, fine
    stream = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    stream.wait_stream(torch.cuda.current_stream())
    with pytest.raises(RuntimeError, match="inside a stream capture"):
        with torch.cuda.graph(graph, stream=stream):
            narrow(xs)
