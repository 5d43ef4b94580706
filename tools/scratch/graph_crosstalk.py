"""Graph replays with zero learning rate must repeat.  Between two replays run an EAGER piece of work on a second model instance
(X = none | fwd | fwdbwd | headonly) and see whether the next replay's gradients change."""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from dcd_amd.engine import trainer

dev = torch.device("cuda:0")
def build():
    args = argparse.Namespace(batch=8, objects=6, precision=os.environ.get("PREC", "bf16x3"), scaling="weak", amp=False)
    return bench.build_everything(args, dev, 1, 0)[:5]
cfg, model, optimizer, images, targets = build()
_, twin, _, _, _ = build()
for g in optimizer.param_groups:
    g["lr"].fill_(0.0)
    g["weight_decay"] = 0.0
step = trainer.GraphedTrainStep(model, optimizer, cfg.SOLVER.GRAD_NORM_CLIP)
def grads():
    torch.cuda.synchronize()
    return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
def worst(a, b):
    rows = sorted(((float((a[n] - b[n]).abs().max() / b[n].abs().max().clamp_min(1e-20)), n) for n in b if not n.endswith("conv.bias") and float(b[n].abs().max()) > 1e-7), reverse=True)
    return "  ".join("%.1e %s" % (v, n[-40:]) for v, n in rows[:3])
step(images, targets); step(images, targets)
g2 = grads()
step(images, targets)
print("X = none      :", worst(grads(), g2), flush=True)
for X in ("fwd", "fwdbwd", "fwdbwd"):
    twin.zero_grad(set_to_none=True)
    if X == "fwd":
        with torch.no_grad():
            twin(images, targets)
    else:
        ld_t, _ = twin(images, targets)
        sum(ld_t.values()).backward()
    torch.cuda.synchronize()
    step(images, targets)
    print("X = %-10s:" % X, worst(grads(), g2), flush=True)
step(images, targets)
print("X = none again:", worst(grads(), g2), flush=True)
