"""Capture the graphed step on instance A at w0 (lr = 0); move the weights of A and of an eager instance B by the same random
amounts; A's next replay against B's eager trainer.train_step: gradients must agree."""
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
    return "  ".join("%.1e %s" % (v, n[-40:]) for v, n in rows[:4]) + "  | median %.1e" % rows[len(rows) // 2][0]
cfg, A, optA, images, targets = build()
_, B, optB, _, _ = build()
clip = cfg.SOLVER.GRAD_NORM_CLIP
step = trainer.GraphedTrainStep(A, optA, clip)
step(images, targets); step(images, targets)
for which in os.environ.get("WHICH", "all").split(","):
    gen = torch.Generator(device=dev).manual_seed(1)
    with torch.no_grad():
        for (n, p), (_, q) in zip(A.named_parameters(), B.named_parameters()):
            d = torch.randn(p.shape, generator=gen, device=dev) * p.abs().mean() * 0.02
            if which == "all" or which in n:
                p.add_(d); q.add_(d)
    step(images, targets)
    gA = grads(A)
    trainer.train_step(B, optB, images, targets, clip)
    print("perturbed %-28s graph(A) vs eager(B): %s" % (which, worst(gA, grads(B))), flush=True)
