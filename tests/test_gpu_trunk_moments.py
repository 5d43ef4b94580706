"""GPU: the shifted-view GEMM behind dcd_amd/model/head/trunk_moments.py (dcd_sgemm_shifted, 64-row tiles of sgemm_f32.inc)
against the same sums written with torch slices in float64 on the host, forward and backward, and the assembled moments against
the patch-matrix form."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(2, 64, 24, 40), (3, 16, 9, 14), (1, 80, 12, 21)])
def test_shift_correlations_match_host_float64(cuda, shape):
    from dcd_amd.model.head import trunk_moments as TM
    torch.manual_seed(3)
    B, C, H, W = shape
    x = torch.randn(B, C, H, W)
    outs = []
    for dev, dt in ((cuda, torch.float32), (torch.device("cpu"), torch.float64)):
        xi = x.to(dev, dt).requires_grad_(True)
        pos = torch.randint(0, H * W, (B, 37), generator=torch.Generator().manual_seed(9))
        pos[:, 0], pos[:, 1], pos[:, 2] = 0, H * W - 1, W - 1                    # corners: the zero padding is read
        R, total, ring, P = TM._ShiftCorr.apply(xi, pos.to(dev))
        assert ring.shape == (B, 9 * C, 2 * (W + 2) + 2 * H)                    # patches of the ring around the image (border correction)
        g = torch.Generator().manual_seed(5)
        wR = torch.randn(R.shape, generator=g, dtype=torch.float64).to(dev)
        wt = torch.randn(total.shape, generator=g, dtype=torch.float64).to(dev)
        wr = torch.randn(ring.shape, generator=g).to(dev, dt)
        wP = torch.randn(P.shape, generator=g).to(dev, dt)
        loss = (R * wR).sum() + (total * wt).sum() + (ring * wr).sum().double() + (P * wP).sum().double()
        loss.backward()
        outs.append((R.detach().cpu().double(), total.detach().cpu().double(), xi.grad.detach().cpu().double(), P.detach().cpu().double(),
                     ring.detach().cpu().double()))
    (R0, t0, g0, P0, r0), (R1, t1, g1, P1, r1) = outs
    assert (r0 - r1).abs().max().item() <= 1e-6 * max(r1.abs().max().item(), 1.0)
    # the ring's patches, independently: pixel (-1, -1) sees only x[0, 0] (at tap (2, 2)), pixel (H, W) only x[H-1, W-1] (tap (0, 0))
    corner = r1.view(B, C, 9, -1)
    assert torch.allclose(corner[:, :, 8, 0], x[:, :, 0, 0].double(), atol=1e-6) and float(corner[:, :, :8, 0].abs().max()) == 0.0
    assert torch.allclose(corner[:, :, 0, 2 * (W + 2) - 1], x[:, :, H - 1, W - 1].double(), atol=1e-6)
    assert (P0 - P1).abs().max().item() <= 1e-6 * max(P1.abs().max().item(), 1.0)     # gathered patches (fp32 copy of the input)
    n = B * H * W
    assert (R0 - R1).abs().max().item() <= 2e-6 * n ** 0.5 * max(R1.abs().max().item() / n ** 0.5, 1.0)
    assert (t0 - t1).abs().max().item() <= 1e-5 * n ** 0.5
    assert (g0 - g1).abs().max().item() <= 2e-5 * g1.abs().max().item()


def test_moments_equal_patch_matrix_form(cuda, monkeypatch):
    """S1 and G from the autocorrelation form == from the explicit patch matrix (the round's first form), on the device."""
    from dcd_amd.model.head import trunk_moments as TM
    torch.manual_seed(1)
    x = torch.randn(2, 64, 24, 40, device=cuda)
    monkeypatch.setenv("DCD_TRUNK_GRAM", "shift")          # ("auto" picks by size: this input would take the patch form)
    S1a, Ga, _ = TM.patch_moments(x)
    monkeypatch.setenv("DCD_TRUNK_GRAM", "bmm")
    S1b, Gb, _ = TM.patch_moments(x)
    assert (S1a - S1b).abs().max().item() <= 1e-3
    assert (Ga - Gb).abs().max().item() <= 2e-5 * Gb.abs().max().item()


@pytest.mark.parametrize("shift,bound", [(0.0, 2e-6), (1.74, 1e-3), (17.4, 2e-2)])
def test_batch_variance_error_with_a_dc_component(cuda, monkeypatch, shift, bound):
    """w^T G w / n - mean^2 is a cancellation form and G is accumulated in fp32 pieces of ~2000 products (advisor, round 2): the
    relative error of the batch variance grows LINEARLY with the input's mean / std.  Measured at 8 x 64 x 96 x 320
    (tools/probes/trunk_dc_error.py): 9e-8 at 0.7 (what a ReLU output has), 1.3e-4 at 1.7, 3.8e-4 at 3.7, 1.1e-3 at 10, 3.7e-3 at 30.
    This test pins that curve from above on a smaller map, against the dense convolution's statistics in float64."""
    from torch.nn import functional as F
    from dcd_amd.model.head import trunk_moments as TM
    torch.manual_seed(0)
    B, C, H, W, O = 2, 64, 48, 80, 64
    w = (torch.randn(O, C, 3, 3) / (C * 9) ** 0.5).to(cuda)
    x = torch.relu(torch.randn(B, C, H, W, device=cuda)) + shift
    y = F.conv2d(x.double(), w.double(), padding=1)
    var_ref = y.var(dim=(0, 2, 3), unbiased=False)
    monkeypatch.setenv("DCD_TRUNK_GRAM", "shift")
    S1, G, _ = TM.patch_moments(x)
    Wd = w.reshape(O, -1).double()
    n = B * H * W
    mean = (Wd @ S1) / n
    var = ((Wd @ G) * Wd).sum(-1) / n - mean * mean
    rel = ((var - var_ref).abs() / var_ref).max().item()
    assert rel <= bound, (shift, rel)


def test_fused_scale_shift_node_equals_tensor_operation_form(cuda, monkeypatch):
    """`_TrunkScaleShift` (one fp64 product + the finalisation kernels of csrc/norm.hip, hand-written backward) against the same
    statistics written as tensor operations with autograd's backward: trunk outputs at the positions, gradients of the input, of
    every trunk's convolution weight and BatchNorm weight / bias, and the running estimates."""
    from torch import nn
    from dcd_amd.model.head import trunk_moments as TM
    from dcd_amd.model.layers.norm import BatchNorm2d
    torch.manual_seed(4)
    B, C, H, W, O, T = 2, 64, 24, 40, 32, 3
    monkeypatch.setenv("DCD_TRUNK_GRAM", "shift")
    x0 = torch.relu(torch.randn(B, C, H, W, device=cuda))
    centers = torch.randint(0, H * W, (B, 7), device=cuda)
    wts = [torch.randn(B, 7, O, device=cuda) for _ in range(T)]
    res = []
    for fused in (True, False):
        monkeypatch.setattr(TM, "_FUSED_STATS", fused)
        torch.manual_seed(9)
        trunks = nn.ModuleList([nn.Sequential(nn.Conv2d(C, O, 3, padding=1, bias=False), BatchNorm2d(O, fuse_relu=True), nn.Identity())
                                for _ in range(T)]).to(cuda).train()
        for t in trunks:
            nn.init.uniform_(t[1].weight, 0.5, 1.5)
            nn.init.uniform_(t[1].bias, -0.5, 0.5)
        x = x0.clone().requires_grad_()
        outs = TM.trunks_at(x, trunks, centers)
        sum((o * w).sum() for o, w in zip(outs, wts)).backward()
        res.append(([o.detach().double() for o in outs], x.grad.double(), [p.grad.double() for p in trunks.parameters()],
                    [t[1].running_mean.double() for t in trunks], [t[1].running_var.double() for t in trunks]))
    (o0, gx0, gp0, rm0, rv0), (o1, gx1, gp1, rm1, rv1) = res

    def close(a, b, tol, what):
        err = (a - b).abs().max().item()
        assert err <= tol * max(b.abs().max().item(), 1e-6), (what, err, b.abs().max().item())
    for a, b in zip(o0, o1):
        close(a, b, 1e-5, "outputs")
    close(gx0, gx1, 1e-4, "grad_input")
    for a, b in zip(gp0, gp1):
        close(a, b, 1e-4, "parameter gradient")
    for a, b in zip(rm0 + rv0, rm1 + rv1):
        close(a, b, 1e-6, "running estimate")
