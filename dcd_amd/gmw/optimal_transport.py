"""Entropy-regularised optimal transport layer (GMW/lib/optimal_transport.py, Campbell, Liu & Gould 2020).

forward: Sinkhorn iterations on K = exp(-lambda min(M, 5)) until every u of the batch moved < tolerance (max 100);
backward: the declarative-node vector-Jacobian product of optimal_transport.py:77-128 (Gould et al. 2019, Lemma 4.4):
one n x n Cholesky factorisation per object; the products are associated so that a single back-substitution replaces the
explicit inverse and the two dense blocks the reference builds (see `gradient`).  Only the uniform-marginal case the GMW model
uses (r, c > 0) is implemented.
"""
import os

import torch


class RegularisedTransportFn(torch.autograd.Function):
    @staticmethod
    def sinkhorn(M, r, c, lmbda=10.0, tolerance=1e-9, max_iterations=100, max_distance=5.0):
        """Cuturi (2013) Algorithm 1 on the Gibbs kernel exp(-lmbda min(M, max_distance)): alternate scalings until no entry of
        the row scaling of ANY object in the batch moved by more than `tolerance` (optimal_transport.py:52-75)."""
        gibbs = torch.exp(-lmbda * M.clamp_max(max_distance))
        gibbs_t = gibbs.transpose(-2, -1)
        row_marg, col_marg = r.unsqueeze(-1), c.unsqueeze(-1)
        scale_r, previous = row_marg.clone(), torch.ones_like(row_marg)
        for _ in range(max_iterations):
            if torch.all(torch.isclose(scale_r, previous, atol=tolerance, rtol=0.0)):
                break
            previous = scale_r
            scale_r = row_marg / gibbs.matmul(col_marg / gibbs_t.matmul(scale_r))
        scale_c = col_marg / gibbs_t.matmul(scale_r)
        return (scale_r * gibbs) * scale_c.transpose(-2, -1)

    @staticmethod
    def gradient(P, lmbda, v):
        """Vector-Jacobian product DJ(M) = DJ(P) DP(M) for v = DJ(P) flattened to (b, m n)  (optimal_transport.py:77-128).

        The transport plan minimises f(M, P) = <M, P> + <P, log P - 1> / lmbda subject to the m + n - 1 independent marginal
        constraints A vec(P) = const (first row constraint dropped).  With H^-1 = diag(lmbda vec(P)) the declarative-node
        result is DP(M) = H^-1 A^T (A H^-1 A^T)^-1 A H^-1 - H^-1.  A H^-1 A^T = [[diag(rows), G], [G^T, diag(cols)]] with
        G = lmbda P without its first row, rows = rowsum(G) (m-1), cols = colsum(lmbda P) (n); its Schur complement is
            S = diag(cols) - G^T diag(1/rows) G                                   (n x n, symmetric positive definite).
        The reference forms S^-1 and two dense (m-1) x n / (m-1) x (m-1) blocks of (A H^-1 A^T)^-1 (126 GFLOP per object
        at 2628 edges) before multiplying the row vector through; everything the product needs is
            w = v H^-1 (as an m x n field),  a = rowsum(w)[1:] / rows,  y = (colsum(w) - a G) S^-1,
            multipliers: columns -> y,  rows -> a - (y G^T) / rows,
        i.e. S (36 GFLOP), one factorisation and one right-hand side.  tests/test_gmw.py checks this against the generic
        formula with an explicit A on small problems, and against finite differences of the fixed point."""
        with torch.no_grad():
            b, m, n = P.size()
            lamP = lmbda * P                                     # H^-1 as an m x n field
            w = v.reshape(b, m, n) * lamP
            G = lamP[:, 1:, :]
            # row / column sums of the (b, m, n) fields as products with a vector of ones: ATen's reduction over the
            # second-to-last axis of a 221 MB tensor takes 1.6 ms (0.14 TB/s), the batched matrix-vector product 26 us
            ones_n, ones_m = lamP.new_ones(n, 1), lamP.new_ones(1, m)
            inv_rows = G.matmul(ones_n).squeeze(-1).reciprocal()              # b x (m-1)
            cols = ones_m.matmul(lamP).squeeze(-2)                            # b x n
            a = (w.matmul(ones_n).squeeze(-1)[:, 1:] * inv_rows).unsqueeze(-2)   # b x 1 x (m-1)
            rhs = ones_m.matmul(w) - a.matmul(G)                              # b x 1 x n
            hip = (P.is_cuda and P.dtype == torch.float32 and n % 4 == 0 and lamP.is_contiguous()
                   and os.environ.get("DCD_GMW_SOLVER", "hip") == "hip")
            if hip:
                # y = rhs S^-1 on our own kernels (csrc/spd.hip): S's lower triangle straight into the solver's buffer (MFMA GEMM,
                # tiles above the diagonal skipped), blocked Cholesky with 128-wide panels whose panel / trailing updates are the
                # same GEMM, the right-hand side carried through the factorisation, blocked backward substitution.
                # The stock route on this ROCm build: MAGMA potrf 8.7 ms for eight 2628^2 systems, then no usable solve --
                # batched potrs with one right-hand side faults (hipErrorLaunchFailure from torch.cholesky_solve on
                # (b, 2628, 1)), two single-column trsm take 8.2 ms, potri 15.3 ms, inverting the factor against the identity
                # 5.0 ms (tools/micro/chol_time.py; kept below as DCD_GMW_SOLVER=stock).
                from dcd_amd import ops
                aug = ops.spd_buffer(b, n, P.device)
                ops.schur_lower(G, inv_rows, cols, aug)
                aug[:, n] = rhs.squeeze(-2)
                y = ops.spd_solve_inplace(aug).unsqueeze(-2)
            else:
                S = -G.transpose(-2, -1).matmul(inv_rows.unsqueeze(-1) * G)
                S.diagonal(dim1=-2, dim2=-1).add_(cols)
                L = torch.linalg.cholesky(S)
                if P.is_cuda:
                    eye = torch.eye(n, dtype=P.dtype, device=P.device).expand(b, n, n)
                    Linv = torch.linalg.solve_triangular(L, eye, upper=False)
                    y = rhs.matmul(Linv.transpose(-2, -1)).matmul(Linv)
                else:
                    y = torch.cholesky_solve(rhs.transpose(-2, -1), L).transpose(-2, -1)
            row_mult = a - y.matmul(G.transpose(-2, -1)) * inv_rows.unsqueeze(-2)     # b x 1 x (m-1)
            # A^T [row multipliers; column multipliers] as an m x n field: entry (i, j) = row_mult[i-1] (i >= 1) + y[j]
            field = y.expand(b, m, n).clone()
            field[:, 1:, :] += row_mult.transpose(-2, -1)
            return (field * lamP - w).flatten(start_dim=-2)

    @staticmethod
    def forward(ctx, M, r, c, lmbda, tolerance, max_iterations):
        P = RegularisedTransportFn.sinkhorn(M.detach(), r.detach(), c.detach(), lmbda, tolerance, max_iterations)
        ctx.lmbda = lmbda
        ctx.save_for_backward(P)
        return P.clone()

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        (P,) = ctx.saved_tensors
        grad_input = None
        if ctx.needs_input_grad[0]:
            grad_input = RegularisedTransportFn.gradient(P, ctx.lmbda, grad_output.flatten(start_dim=-2)).reshape(P.size())
        return grad_input, None, None, None, None, None


class RegularisedTransport(torch.nn.Module):
    def __init__(self, lmbda=10.0, tolerance=1e-9, max_iterations=100):
        super().__init__()
        self.lmbda, self.tolerance, self.max_iterations = lmbda, tolerance, max_iterations

    def forward(self, M, r, c):
        if not (bool((r > 0).all()) and bool((c > 0).all())):
            raise NotImplementedError("zero prior probabilities (optimal_transport.py:186-216) are not used by GMW and not built")
        return RegularisedTransportFn.apply(M, r, c, self.lmbda, self.tolerance, self.max_iterations)
