"""Weights after k graphed steps against k eager steps (bf16x3: deterministic forward / backward), in units of the learning rate."""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from dcd_amd.engine import trainer

dev = torch.device("cuda:0")
K = [1, 2, 3, 5, 10, 20, 30, 40, 50]
res = {}
for mode in ("eager", "graph", "eager2"):
    args = argparse.Namespace(batch=8, objects=6, precision=os.environ.get("PREC", "bf16x3"), scaling="weak", amp=False)
    cfg, model, optimizer, images, targets = bench.build_everything(args, dev, 1, 0)[:5]
    clip = cfg.SOLVER.GRAD_NORM_CLIP
    lr = float(optimizer.param_groups[0]["lr"])
    step = trainer.GraphedTrainStep(model, optimizer, clip) if mode == "graph" else None
    snaps = {}
    for it in range(1, max(K) + 1):
        ld, _ = step(images, targets) if step else trainer.train_step(model, optimizer, images, targets, clip)
        if it in K:
            torch.cuda.synchronize()
            snaps[it] = ({n: p.detach().clone() for n, p in model.named_parameters()}, float(sum(float(v) for v in ld.values())))
    res[mode] = snaps
    del step, model, optimizer
    torch.cuda.empty_cache()
print("lr %.2e" % lr)
for k in K:
    for other in ("graph", "eager2"):
        a, b = res["eager"][k][0], res[other][k][0]
        rows = sorted(((float((a[n] - b[n]).abs().max()) / lr, n) for n in a if not n.endswith("conv.bias")), reverse=True)
        print("after %2d steps  eager vs %-6s loss %.5f / %.5f  max |dw|/lr: %.2e (%s)  median %.1e" % (
            k, other, res["eager"][k][1], res[other][k][1], rows[0][0], rows[0][1][-40:], rows[len(rows) // 2][0]), flush=True)
