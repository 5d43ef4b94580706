"""Kernel time of dcd_edge_depth_forward (run under `rocprofv3 --kernel-trace --stats`): N objects x 73 keypoints, with the
training top-1500 ordering and without (topk = 0: the pair depths only)."""
import sys

import torch

from dcd_amd import _lib

dev = torch.device("cuda")
L = _lib.lib()
K = 73
for N in (80, 640):
    g = torch.Generator().manual_seed(N)
    kps = torch.randn(N, K, 2, generator=g).to(dev)
    k3 = torch.randn(N, K, 3, generator=g).to(dev)
    rot = torch.randn(N, generator=g).to(dev)
    P = torch.randn(N, 3, 4, generator=g).abs().add(1).to(dev)
    for topk in (1500, 0):
        M = topk if topk else K * (K - 1) // 2
        depth = torch.empty(N, M, device=dev)
        idx = torch.empty(N, M, dtype=torch.int32, device=dev)
        mask = torch.empty(N, M, device=dev)
        st = _lib.stream_of(kps)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for rep in range(2):
            e0.record()
            for _ in range(20):
                _lib.check(L.dcd_edge_depth_forward(st, kps.data_ptr(), k3.data_ptr(), rot.data_ptr(), P.data_ptr(), None, N, K, topk,
                                                    2.0, 80.0, 0, 1, depth.data_ptr(), idx.data_ptr(), mask.data_ptr()), "edge")
            e1.record()
            torch.cuda.synchronize()
        print("N %4d topk %4d: %.1f us per launch (back to back)" % (N, topk, e0.elapsed_time(e1) * 50))
