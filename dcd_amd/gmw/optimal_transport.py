"""Entropy-regularised optimal transport layer (GMW/lib/optimal_transport.py, Campbell, Liu & Gould 2020).

forward: Sinkhorn iterations on K = exp(-lambda min(M, 5)) until every u of the batch moved < tolerance (max 100);
backward: the declarative-node vector-Jacobian product of optimal_transport.py:77-128 (Gould et al. 2019, Lemma 4.4):
one n x n Cholesky factorisation + inverse per object.  Same arithmetic in the same order; the per-object Python loop over
`cholesky_inverse` (reference :113-114, "currently cannot handle batches") is one batched call, and `torch.cholesky` is
spelled `torch.linalg.cholesky`.  Only the uniform-marginal case the GMW model uses (r, c > 0) is implemented.
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
    def gradient(P, lmbda, v):
        """DJ(M) = DJ(P) DP(M), v = DJ(P) flattened to (b, m n)."""
        with torch.no_grad():
            b, m, n = P.size()
            B = lmbda * P
            hinv = B.flatten(start_dim=-2)
            d1inv = B.sum(-1)[:, 1:].reciprocal()
            d2 = B.sum(-2)
            B = B[:, 1:, :]
            S = -B.transpose(-2, -1).matmul(d1inv.unsqueeze(-1) * B)
            S.diagonal(dim1=-2, dim2=-1).add_(d2)
            Sinv = torch.cholesky_inverse(torch.linalg.cholesky(S))
            R = -B.matmul(Sinv) * d1inv.unsqueeze(-1)
            Q = -R.matmul(B.transpose(-2, -1) * d1inv.unsqueeze(-2))
            Q.diagonal(dim1=-2, dim2=-1).add_(d1inv)
            vHinv = v * hinv
            blocks = vHinv.reshape((-1, m, n))
            u1 = blocks.sum(-1)[:, 1:].unsqueeze(-2)
            u2 = blocks.sum(-2).unsqueeze(-2)
            u3 = u1.matmul(Q) + u2.matmul(R.transpose(-2, -1))
            u4 = u1.matmul(R) + u2.matmul(Sinv)
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
