import sys, time, torch
dev = torch.device("cuda:0")
n, B = 2628, int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.manual_seed(0)
A = torch.randn(B, n, n, device=dev)
S = A @ A.transpose(-2, -1) / n + torch.eye(n, device=dev)
def timed(fn, it=3):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / it * 1e3
for lib in ("default", "cusolver", "magma"):
    try:
        torch.backends.cuda.preferred_linalg_library(lib)
        L = torch.linalg.cholesky(S)
        print(lib, "cholesky %.2f ms" % timed(lambda: torch.linalg.cholesky(S)), " cholesky_inverse %.2f ms" % timed(lambda: torch.cholesky_inverse(L)),
              " inv via solve_triangular %.2f ms" % timed(lambda: torch.linalg.solve_triangular(L, torch.eye(n, device=dev).expand(B, n, n), upper=False)))
    except Exception as e:
        print(lib, "failed:", str(e)[:100])
torch.backends.cuda.preferred_linalg_library("default")
L = torch.linalg.cholesky(S)
rhs = torch.randn(B, n, 1, device=dev)
def two_trsm():
    z = torch.linalg.solve_triangular(L, rhs, upper=False)
    return torch.linalg.solve_triangular(L.transpose(-2, -1), z, upper=True)
y = two_trsm()
torch.cuda.synchronize()
ref = torch.linalg.solve(S.double(), rhs.double()).float()
print("two single-rhs trsm %.2f ms, rel err %.2e" % (timed(two_trsm), float((y - ref).abs().max() / ref.abs().max())))
