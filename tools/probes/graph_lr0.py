"""Graphed step with every learning rate at zero: weights, inputs and therefore losses and gradients must repeat exactly from
replay to replay.  Prints the loss and the relative change of every parameter gradient against replay 1."""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from dcd_amd.engine import trainer

N = int(os.environ.get("N", "60"))
dev = torch.device("cuda:0")
prec = os.environ.get("PREC", "bf16x3")
args = argparse.Namespace(batch=8, objects=6, precision=prec, scaling="weak", amp=False)
cfg, model, optimizer, images, targets = bench.build_everything(args, dev, 1, 0)[:5]
for g in optimizer.param_groups:
    g["lr"].fill_(0.0) if torch.is_tensor(g["lr"]) else None
    g["weight_decay"] = 0.0
clip = cfg.SOLVER.GRAD_NORM_CLIP
step = trainer.GraphedTrainStep(model, optimizer, clip) if os.environ.get("MODE", "graph") == "graph" else None
ref = None
named = dict(model.named_parameters())
w0 = {n: p.detach().clone() for n, p in named.items()}
for it in range(N):
    ld, log = step(images, targets) if step else trainer.train_step(model, optimizer, images, targets, clip)
    torch.cuda.synchronize()
    total = float(sum(float(v) for v in ld.values()))
    grads = {n: p.grad.detach().clone() for n, p in named.items() if p.grad is not None}
    if ref is None:
        ref = grads
        print("replay %2d loss %.9f (reference)" % (it, total), flush=True)
        continue
    worst = sorted(((float((g - ref[n]).abs().max() / ref[n].abs().max().clamp_min(1e-20)), n) for n, g in grads.items() if not n.endswith("conv.bias")), reverse=True)[:3]
    dw = max(float((p.detach() - w0[n]).abs().max()) for n, p in named.items())
    if it < 6 or it % 6 == 0:
        print("replay %2d loss %.9f  max|dw| %.1e  worst grad changes: %s" % (it, total, dw, "  ".join("%.1e %s" % (v, n[-46:]) for v, n in worst)), flush=True)
