"""GPU: batched SPD solve (csrc/spd.hip: blocked Cholesky on the fp32 matrix pipe + blocked substitutions) against
torch.linalg.solve in float64 on the host."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def spd(b, n, seed, cond_shift):
    g = torch.Generator().manual_seed(seed)
    A = torch.randn(b, n, n, generator=g, dtype=torch.float64)
    S = A @ A.transpose(1, 2) / n + cond_shift * torch.eye(n, dtype=torch.float64)
    return S, torch.randn(b, n, generator=g, dtype=torch.float64)


@pytest.mark.parametrize("case", [(2, 64, 0.5), (3, 128, 0.5), (2, 132, 0.1), (2, 300, 0.05), (1, 1024, 0.05), (8, 2628, 0.02)])
def test_spd_solve_matches_float64(cuda, case):
    from dcd_amd import ops
    b, n, shift = case
    S, r = spd(b, n, 7, shift)
    ref = torch.linalg.solve(S, r.unsqueeze(-1)).squeeze(-1)
    y = ops.spd_solve(S.float().to(cuda).contiguous(), r.float().to(cuda)).cpu().double()
    # fp32 factorisation: error ~ cond(S) * 2^-24 relative to the solution's scale
    cond = float(torch.linalg.cond(S[0]))
    assert (y - ref).abs().max().item() <= 4e-7 * cond * ref.abs().max().item() + 1e-6, (cond, (y - ref).abs().max().item())


def test_transport_schur_system(cuda):
    """The system the transport layer's backward solves (Schur complement of a transport plan), against float64."""
    from dcd_amd import ops
    from dcd_amd.gmw.optimal_transport import RegularisedTransportFn as T
    torch.manual_seed(0)
    b, n = 2, 516
    M = torch.rand(b, n, n, dtype=torch.float64)
    r = torch.full((b, n), 1.0 / n, dtype=torch.float64)
    P = T.sinkhorn(M, r, r, 10.0, 1e-12, 500)
    lamP = 10.0 * P
    G = lamP[:, 1:, :]
    S = -G.transpose(1, 2) @ (G.sum(-1).reciprocal().unsqueeze(-1) * G)
    S.diagonal(dim1=-2, dim2=-1).add_(lamP.sum(-2))
    rhs = torch.randn(b, n, dtype=torch.float64)
    ref = torch.linalg.solve(S, rhs.unsqueeze(-1)).squeeze(-1)
    y = ops.spd_solve(S.float().to(cuda).contiguous(), rhs.float().to(cuda)).cpu().double()
    assert (y - ref).abs().max().item() <= 2e-4 * ref.abs().max().item()


def test_not_positive_definite_yields_nan(cuda):
    """A matrix that is not positive definite must not produce a finite-looking solution (advisor r2): the kernel poisons the
    factor at the first non-positive pivot, y is NaN (torch.linalg.cholesky would raise), and the train step's non-finite
    guard then skips the update.  The well-conditioned matrix in the same batch is unaffected."""
    from dcd_amd import ops
    S, r = spd(2, 132, 3, 0.5)
    S[1, 40, 40] = -5.0
    y = ops.spd_solve(S.float().to(cuda).contiguous(), r.float().to(cuda)).cpu()
    ref0 = torch.linalg.solve(S[0], r[0])
    assert torch.isfinite(y[0]).all() and (y[0].double() - ref0).abs().max() <= 1e-4 * ref0.abs().max()
    assert torch.isnan(y[1]).any()
