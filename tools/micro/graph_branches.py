"""Does a captured HIP graph run independent branches concurrently on this stack?  N chains of tiny element-wise kernels,
captured on one stream vs. forked over side streams; replay time per variant (tools/micro: evidence for the loss graph)."""
import sys
import time
import torch

dev = torch.device("cuda:0")
NCH, LEN = 8, 100
xs = [torch.randn(64, device=dev) for _ in range(NCH)]


def chain(x):
    for _ in range(LEN):
        x = x * 1.0001 + 0.5
    return x


def capture(nstreams):
    side = [torch.cuda.Stream() for _ in range(nstreams)]
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream()
        outs = []
        if nstreams == 0:
            outs = [chain(x) for x in xs]
        else:
            for i, x in enumerate(xs):
                s = side[i % nstreams]
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    outs.append(chain(x))
            for s in side:
                cur.wait_stream(s)
        total = torch.stack([o.sum() for o in outs]).sum()
    return g, total


for ns in (0, 2, 4, 8):
    g, total = capture(ns)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print("streams %d: %.3f ms per replay (%d kernels, %.2f us each), total %.6f" % (ns, dt * 1e3, NCH * LEN * 2, dt * 1e6 / (NCH * LEN * 2), total.item()))
