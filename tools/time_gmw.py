"""GMW train step on the GPU (dcd_amd/gmw): objects/s and where the time goes.  python tools/time_gmw.py [B=8] [iters=5]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import torch
from make_golden_gmw import inputs
from dcd_amd import ops
from dcd_amd.gmw import GMW, gmw_train_step
from dcd_amd.gmw.optimal_transport import RegularisedTransportFn

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = GMW().to(dev).train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.999))
k2, k3, rot, loc = (torch.from_numpy(a).to(dev) for a in inputs(seed=11, B=B))


def timed(fn, n=iters):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


step = timed(lambda: gmw_train_step(model, opt, k2, k3, rot, loc, 0.1, 1.0))
print("B=%d: train step %.1f ms  (%.1f objects/s)" % (B, step, B / step * 1e3))
with torch.no_grad():
    f4, f6 = model.edge_expand(k2), model.edge_expand(k3)
    print("  compute_z (HIP solver)        %.2f ms" % timed(lambda: ops.compute_z(k2, k3, rot)))
    print("  extractors (2 x 37 Conv1d)    %.2f ms" % timed(lambda: (model.FeatureExtractor4d(f4.transpose(-2, -1)), model.FeatureExtractor6d(f6.transpose(-2, -1)))))
    a = torch.nn.functional.normalize(model.FeatureExtractor4d(f4.transpose(-2, -1)).transpose(-2, -1), dim=-1)
    b = torch.nn.functional.normalize(model.FeatureExtractor6d(f6.transpose(-2, -1)).transpose(-2, -1), dim=-1)
    from dcd_amd.gmw import pairwise_l2_dist
    print("  pairwise distances            %.2f ms" % timed(lambda: pairwise_l2_dist(a, b)))
    M = pairwise_l2_dist(a, b)
    r = M.new_ones((B, M.shape[1])) / M.shape[1]
    print("  Sinkhorn forward              %.2f ms" % timed(lambda: RegularisedTransportFn.sinkhorn(M, r, r, 10.0, 1e-9, 100)))
    P = RegularisedTransportFn.sinkhorn(M, r, r, 10.0, 1e-9, 100)
    g = torch.randn(B, M.shape[1] * M.shape[2], device=dev)
    print("  transport backward (Cholesky) %.2f ms" % timed(lambda: RegularisedTransportFn.gradient(P, 10.0, g)))
print("  peak memory %.2f GB" % (torch.cuda.max_memory_allocated() / 2 ** 30))
