"""Does a multi-block torch reduction survive HIP-graph replay?  (round 2: the graphed loss returned 0x / 2x / 1e5x sums of a
(320, 1500) tensor from step 2 on while every input was bit-identical to the eager evaluation.)"""
import torch
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
x = torch.randn(320, 1500, device=dev)
w = torch.randn(4096, 4096, device=dev)
def body(x):
    y = (w @ w).sum() * 0                      # a heavy kernel in front, like the model's
    v = (x.abs() * 2.0)
    return v.sum() + y, v.sum(dim=1).sum() + y, (v > 1).float().sum()
# warmup
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): body(x)
torch.cuda.current_stream().wait_stream(s)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    out = body(x)
bad = [0, 0, 0]
for it in range(300):
    x.copy_(torch.randn(320, 1500, generator=g).to(dev))
    graph.replay()
    torch.cuda.synchronize()
    got = [float(o) for o in out]
    ref = [float(o) for o in body(x)]
    for i in range(3):
        if abs(got[i] - ref[i]) > 1e-3 * abs(ref[i]):
            bad[i] += 1
            if bad[i] <= 3: print("iter", it, "output", i, "graph", got[i], "eager", ref[i])
print("mismatches over 300 replays: flat sum %d, two-stage sum %d, count %d" % tuple(bad))
