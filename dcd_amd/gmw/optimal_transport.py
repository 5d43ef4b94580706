"""Entropy-regularised optimal transport layer (GMW/lib/optimal_transport.py, Campbell, Liu & Gould 2020).

forward: Sinkhorn iterations on K = exp(-lambda min(M, 5)) until every u of the batch moved < tolerance (max 100);
backward: the declarative-node vector-Jacobian product of optimal_transport.py:77-128 (Gould et al. 2019, Lemma 4.4):
one n x n Cholesky factorisation per object; the products are associated so that a single back-substitution replaces the
explicit inverse and the two (m-1) x n x n products of the reference (see `gradient`), and `torch.cholesky` is spelled
`torch.linalg.cholesky`.  Only the uniform-marginal case the GMW model uses (r, c > 0) is implemented.
"""
import torch


class RegularisedTransportFn(torch.autograd.Function):
    @staticmethod
    def sinkhorn(M, r, c, lmbda=10.0, tolerance=1e-9, max_iterations=100, max_distance=5.0):
        K = (-lmbda * M.clamp_max(max_distance)).exp()
        Kt = K.transpose(-2, -1)
        r = r.unsqueeze(-1)
        c = c.unsqueeze(-1)
        u = r.clone()
        u_prev = torch.ones_like(u)
        for _ in range(max_iterations):
            if torch.all(torch.isclose(u, u_prev, atol=tolerance, rtol=0.0)):
                break
            u_prev = u
            u = r / K.matmul(c / Kt.matmul(u))
        v = c / Kt.matmul(u)
        return (u * K) * v.transpose(-2, -1)

    @staticmethod
    def gradient(P, lmbda, v, explicit_inverse=False):
        """DJ(M) = DJ(P) DP(M), v = DJ(P) flattened to (b, m n).

        With H^-1 = diag(lmbda vec(P)), B = lmbda P[1:], D1 = diag(1 / rowsum(lmbda P)[1:]), D2 = diag(colsum(lmbda P)) and
        S = D2 - B^T D1 B, the reference builds S^-1, R = -D1 B S^-1 and Q = D1 - R B^T D1 explicitly (two more
        (m-1) x n x n products and a triangular inverse per object: 126 GFLOP for 2628 edges) and then multiplies the row
        vectors u1, u2 through them.  The same products associate the other way round into ONE solve:
            y = (u2 - (u1 D1) B) S^-1,      u4 = u1 R + u2 S^-1 = y,      u3 = u1 Q + u2 R^T = u1 D1 - (y B^T) D1,
        i.e. form S (36 GFLOP), factor it (6 GFLOP) and back-substitute one right-hand side (on the GPU: inverse of the
        triangular factor + two mat-vecs, see below).  `explicit_inverse=True` keeps the reference's order of operations (the CPU
        test compares the two)."""
        with torch.no_grad():
            b, m, n = P.size()
            B = lmbda * P
            hinv = B.flatten(start_dim=-2)
            d1inv = B.sum(-1)[:, 1:].reciprocal()
            d2 = B.sum(-2)
            B = B[:, 1:, :]
            S = -B.transpose(-2, -1).matmul(d1inv.unsqueeze(-1) * B)
            S.diagonal(dim1=-2, dim2=-1).add_(d2)
            L = torch.linalg.cholesky(S)
            vHinv = v * hinv
            blocks = vHinv.reshape((-1, m, n))
            u1 = blocks.sum(-1)[:, 1:].unsqueeze(-2)
            u2 = blocks.sum(-2).unsqueeze(-2)
            if explicit_inverse:
                Sinv = torch.cholesky_inverse(L)
                R = -B.matmul(Sinv) * d1inv.unsqueeze(-1)
                Q = -R.matmul(B.transpose(-2, -1) * d1inv.unsqueeze(-2))
                Q.diagonal(dim1=-2, dim2=-1).add_(d1inv)
                u3 = u1.matmul(Q) + u2.matmul(R.transpose(-2, -1))
                u4 = u1.matmul(R) + u2.matmul(Sinv)
            else:
                a = u1 * d1inv.unsqueeze(-2)                                            # b x 1 x (m-1)
                rhs = u2 - a.matmul(B)                                                  # b x 1 x n
                if P.is_cuda:
                    # y = rhs S^-1 = (rhs L^-T) L^-1.  On this ROCm build batched potrs with one right-hand side faults
                    # (hipErrorLaunchFailure from torch.cholesky_solve on (b, 2628, 1)), two single-column trsm take 8.2 ms
                    # and potri 15.3 ms at b = 8; inverting the triangular factor against the identity takes 5.0 ms
                    # (tools/micro/chol_time.py), followed by two mat-vecs
                    eye = torch.eye(n, dtype=P.dtype, device=P.device).expand(b, n, n)
                    Linv = torch.linalg.solve_triangular(L, eye, upper=False)
                    u4 = rhs.matmul(Linv.transpose(-2, -1)).matmul(Linv)                # y: b x 1 x n
                else:
                    u4 = torch.cholesky_solve(rhs.transpose(-2, -1), L).transpose(-2, -1)
                u3 = a - u4.matmul(B.transpose(-2, -1)) * d1inv.unsqueeze(-2)
            u5 = u3.expand(-1, n, -1).transpose(-2, -1) + u4.expand(-1, m - 1, -1)
            uHinv = torch.cat((u4, u5), dim=-2).flatten(start_dim=-2) * hinv
            return uHinv - vHinv

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
