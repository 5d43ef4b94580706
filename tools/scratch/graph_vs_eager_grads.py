"""Gradients of the graphed step against the eager step, same weights (every learning rate zero), same batch, bf16x3 (the mode
whose eager gradients repeat to 5e-6): per parameter max |g_graph - g_eager| / max |g_eager|."""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from dcd_amd.engine import trainer

dev = torch.device("cuda:0")
res = {}
for mode in ("eager", "graph"):
    args = argparse.Namespace(batch=8, objects=6, precision=os.environ.get("PREC", "bf16x3"), scaling="weak", amp=False)
    cfg, model, optimizer, images, targets = bench.build_everything(args, dev, 1, 0)[:5]
    for g in optimizer.param_groups:
        g["lr"].fill_(0.0)
        g["weight_decay"] = 0.0
    clip = cfg.SOLVER.GRAD_NORM_CLIP
    step = trainer.GraphedTrainStep(model, optimizer, clip) if mode == "graph" else None
    for it in range(3):
        ld, _ = step(images, targets) if step else trainer.train_step(model, optimizer, images, targets, clip)
    torch.cuda.synchronize()
    res[mode] = ({n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}, float(sum(float(v) for v in ld.values())))
    del step, model, optimizer
    torch.cuda.empty_cache()
ge, gg = res["eager"][0], res["graph"][0]
print("loss eager %.9f graph %.9f" % (res["eager"][1], res["graph"][1]))
rows = sorted(((float((gg[n] - g).abs().max() / g.abs().max().clamp_min(1e-20)), n, float(g.abs().max())) for n, g in ge.items() if n in gg), reverse=True)
print("missing in graph:", [n for n in ge if n not in gg][:5], " extra:", [n for n in gg if n not in ge][:5])
for v, n, s in rows[:25]:
    print("  %.3e  %-66s |g| %.2e" % (v, n, s))
print("  median %.2e" % rows[len(rows) // 2][0])
