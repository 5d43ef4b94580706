"""Gradients at identical weights three ways in one process: instance A eager, instance B eager, instance A graphed (lr = 0)."""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from dcd_amd.engine import trainer

dev = torch.device("cuda:0")
def build():
    args = argparse.Namespace(batch=8, objects=6, precision=os.environ.get("PREC", "bf16x3"), scaling="weak", amp=False)
    r = bench.build_everything(args, dev, 1, 0)[:5]
    for g in r[2].param_groups:
        g["lr"].fill_(0.0)
        g["weight_decay"] = 0.0
    return r
def grads(m):
    torch.cuda.synchronize()
    return {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
def worst(a, b):
    rows = sorted(((float((a[n] - b[n]).abs().max() / b[n].abs().max().clamp_min(1e-20)), n) for n in b if not n.endswith("conv.bias") and float(b[n].abs().max()) > 1e-7), reverse=True)
    return "  ".join("%.1e %s" % (v, n[-40:]) for v, n in rows[:3])
cfg, A, optA, images, targets = build()
_, B, optB, _, _ = build()
clip = cfg.SOLVER.GRAD_NORM_CLIP
order = os.environ.get("ORDER", "A,B,G,A,B")
out = {}
step = None
for k, what in enumerate(order.split(",")):
    if what == "A":
        trainer.train_step(A, optA, images, targets, clip); out["A%d" % k] = grads(A)
    elif what == "B":
        trainer.train_step(B, optB, images, targets, clip); out["B%d" % k] = grads(B)
    else:
        step = step or trainer.GraphedTrainStep(A, optA, clip)
        step(images, targets); out["G%d" % k] = grads(A)
keys = list(out)
for i in range(1, len(keys)):
    print("%s vs %s: %s" % (keys[i], keys[0], worst(out[keys[i]], out[keys[0]])), flush=True)
