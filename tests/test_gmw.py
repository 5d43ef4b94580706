"""GMW train step (dcd_amd/gmw) against tests/golden/gmw.npz, which was produced by running the
reference's own GMW code (tests/golden/make_golden_gmw.py): identical initialisation from the same seed, losses, transport
plan, regression weights and per-parameter gradients of one step on 2 objects x 2628 edges.
CPU test: edge depths from the oracle restatement of compute_z; GPU test: from the HIP solver kernel, model on the device."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_golden_gmw import inputs  # noqa: E402  (seeded input builder shared with the generator; pure numpy)


def _run(device, compute_z):
    from dcd_amd.gmw import GMW, gmw_losses
    fx = np.load(os.path.join(HERE, "golden", "gmw.npz"))
    torch.manual_seed(0)
    model = GMW().train()
    names = [n for n, _ in model.named_parameters()]
    assert names == list(fx["param_names"])                    # checkpoints are interchangeable
    sums = np.array([float(p.detach().double().sum()) for _, p in model.named_parameters()])
    asums = np.array([float(p.detach().double().abs().sum()) for _, p in model.named_parameters()])
    assert np.array_equal(sums, fx["param_sums"]) and np.array_equal(asums, fx["param_abs_sums"])      # same init stream
    model = model.to(device)
    k2, k3, rot, loc = (torch.from_numpy(a).to(device) for a in inputs())
    loss, cls, reg, z = gmw_losses(model, k2, k3, rot, loc, 0.1, 1.0, compute_z=compute_z)
    loss.backward()
    assert abs(float(loss) - float(fx["loss"])) <= 2e-5 * abs(float(fx["loss"]))
    assert abs(float(cls) - float(fx["cls_loss"])) <= 2e-6
    assert abs(float(reg) - float(fx["reg_loss"])) <= 2e-5 * abs(float(fx["reg_loss"]))
    assert np.allclose(z.detach().cpu().numpy(), fx["pred_depth"], rtol=2e-5)
    with torch.no_grad():
        reg_weights, P = model(k2, k3, rot)
    P = P.cpu()
    assert np.allclose(reg_weights.cpu().numpy(), fx["reg_weights"], rtol=2e-4)
    assert np.allclose(P.diagonal(dim1=-2, dim2=-1).numpy(), fx["P_diag"], rtol=1e-3, atol=1e-9)
    assert np.allclose(P[:, :64, :64].numpy(), fx["P_block"], rtol=1e-3, atol=1e-9)
    assert np.allclose(P.sum(-1).numpy(), fx["P_row_sums"], rtol=1e-4) and np.allclose(P.sum(-2).numpy(), fx["P_col_sums"], rtol=1e-4)
    assert np.allclose(P.sum(-1).numpy(), 1.0 / 2628, rtol=1e-3)                                   # a transport plan
    gn = np.array([float(p.grad.double().norm()) for _, p in model.named_parameters()])
    gs = np.array([float(p.grad.double().sum()) for _, p in model.named_parameters()])
    scale = fx["grad_norms"].max()
    # biases in front of a context normalisation have a mathematically zero gradient (round-off noise in both codes)
    assert np.all(np.abs(gn - fx["grad_norms"]) <= 1e-3 * fx["grad_norms"] + 1e-4 * scale), np.abs(gn - fx["grad_norms"]).max()
    # a second, sign-sensitive statistic: |sum(g) - sum(g_ref)| <= sqrt(numel) ||g - g_ref||, so 1e-3 relative accuracy of the
    # gradient bounds it by 1e-3 sqrt(numel) ||g_ref||  (the Cholesky-based backward differs at the 1e-4 level between LAPACK
    # on the host and rocSOLVER on the device)
    numel = np.array([p.numel() for _, p in model.named_parameters()], dtype=np.float64)
    assert np.all(np.abs(gs - fx["grad_sums"]) <= 1e-3 * np.sqrt(numel) * fx["grad_norms"] + 1e-4 * scale)
    return model, (k2, k3, rot, loc)


def test_gmw_step_matches_reference_fixture_cpu():
    from oracle import torch_ops
    _run(torch.device("cpu"), torch_ops.compute_z)


def test_transport_layer_gradient_is_the_derivative_of_its_fixed_point():
    """Finite differences through the Sinkhorn fixed point (fp64, small problem) against the declarative backward."""
    from dcd_amd.gmw.optimal_transport import RegularisedTransport
    torch.manual_seed(1)
    M = torch.rand(2, 6, 6, dtype=torch.float64).requires_grad_()
    r = torch.full((2, 6), 1 / 6, dtype=torch.float64)
    layer = RegularisedTransport(10.0, 1e-14, 2000)
    W = torch.rand(2, 6, 6, dtype=torch.float64)
    (layer(M, r, r) * W).sum().backward()
    eps = 1e-6
    for idx in [(0, 0, 0), (1, 3, 2), (0, 5, 4)]:
        Mp, Mm = M.detach().clone(), M.detach().clone()
        Mp[idx] += eps
        Mm[idx] -= eps
        fd = ((layer(Mp, r, r) * W).sum() - (layer(Mm, r, r) * W).sum()) / (2 * eps)
        assert abs(float(fd) - float(M.grad[idx])) <= 1e-5 * max(abs(float(fd)), 1e-3)


@pytest.mark.gpu
def test_gmw_step_matches_reference_fixture_gpu(cuda):
    from dcd_amd import ops
    model, (k2, k3, rot, loc) = _run(cuda, ops.compute_z)
    from dcd_amd.gmw import gmw_train_step
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.999))
    before = float(gmw_train_step(model, opt, k2, k3, rot, loc, 0.1, 1.0)[0])
    for _ in range(3):
        last = float(gmw_train_step(model, opt, k2, k3, rot, loc, 0.1, 1.0)[0])
    assert np.isfinite(last) and last < before                       # the step optimises what it reports


def test_transport_backward_equals_generic_declarative_formula():
    """The re-associated backward (one Cholesky solve) against Lemma 4.4 of Gould et al. evaluated literally on small
    problems: DP(M) = H^-1 A^T (A H^-1 A^T)^-1 A H^-1 - H^-1 with the (m + n - 1) x mn constraint matrix A written out."""
    from dcd_amd.gmw.optimal_transport import RegularisedTransportFn as T
    torch.manual_seed(2)
    for dtype, tol, (m, n) in ((torch.float64, 1e-9, (7, 5)), (torch.float64, 1e-9, (6, 6)), (torch.float32, 5e-4, (12, 12))):
        M = torch.rand(2, m, n, dtype=dtype)
        r = torch.full((2, m), 1.0 / m, dtype=dtype)
        c = torch.full((2, n), 1.0 / n, dtype=dtype)
        P = T.sinkhorn(M, r, c, 10.0, 1e-12, 1000)
        v = torch.randn(2, m * n, dtype=dtype)
        got = T.gradient(P, 10.0, v)
        A = torch.zeros(m + n - 1, m * n, dtype=torch.float64)
        for i in range(1, m):                       # row constraints, first row dropped
            A[i - 1, i * n:(i + 1) * n] = 1
        for j in range(n):                          # column constraints
            A[m - 1 + j, j::n] = 1
        for k in range(2):
            Hinv = torch.diag(10.0 * P[k].double().flatten())
            DP = Hinv @ A.T @ torch.linalg.inv(A @ Hinv @ A.T) @ A @ Hinv - Hinv
            ref = v[k].double() @ DP
            assert (got[k].double() - ref).abs().max().item() <= tol * ref.abs().max().item()


@pytest.mark.gpu
def test_context_norm_kernel_against_stock_formula(cuda):
    """csrc/heads.hip context normalisation (forward + backward) against the reference's formula in fp64."""
    from dcd_amd import ops
    g = torch.Generator().manual_seed(3)
    for B, C, K in ((2, 128, 2628), (3, 5, 37), (1, 7, 4096), (2, 3, 100), (1, 2, 4100)):      # row-in-registers kernel: K % 4 == 0, K <= 4096
        x = torch.randn(B, C, K, generator=g) * 2 + 0.5
        gy = torch.randn(B, C, K, generator=g)
        xd = x.double().requires_grad_()
        m = torch.mean(xd, 2, keepdim=True)
        v = torch.var(xd, 2, keepdim=True)
        ref = (xd - m) * (1.0 / torch.sqrt(v + 1e-3))
        ref.backward(gy.double())
        xg = x.to(cuda).requires_grad_()
        y = ops.context_norm(xg, 1e-3)
        y.backward(gy.to(cuda))
        assert (y.detach().cpu().double() - ref.detach()).abs().max().item() <= 2e-6 * ref.abs().max().item()
        assert (xg.grad.cpu().double() - xd.grad).abs().max().item() <= 1e-5 * xd.grad.abs().max().item()


def test_gmw_val_step_rescales_the_location_along_its_ray():
    from oracle import torch_ops
    from dcd_amd.gmw import GMW, gmw_val_step
    torch.manual_seed(0)
    model = GMW().eval()
    k2, k3, rot, loc = (torch.from_numpy(a) for a in inputs())
    dim = torch.tensor([[1.5, 1.6, 3.9], [1.4, 1.7, 4.1]])
    fx = np.load(os.path.join(HERE, "golden", "gmw.npz"))
    loss, cls, reg, z, ploc = gmw_val_step(model, k2, k3, rot, loc, dim, 0.1, 1.0, compute_z=torch_ops.compute_z)
    assert abs(float(loss) - float(fx["loss"])) <= 2e-5 * abs(float(fx["loss"]))       # no train/eval difference: no BN, no dropout
    assert torch.allclose(ploc[:, 2], z, rtol=1e-6)                                       # the new depth is the predicted one
    centre_raw = loc.clone(); centre_raw[:, 1] -= dim[:, 0] / 2
    centre_new = ploc.clone(); centre_new[:, 1] -= dim[:, 0] / 2
    assert torch.allclose(centre_new / centre_new[:, 2:3], centre_raw / centre_raw[:, 2:3], rtol=1e-5, atol=1e-6)   # same ray
