"""Capture the step at weights w0 (learning rates zero), then move every weight by a small random amount -- the same in an eager
twin -- and replay once: the graph's gradients must equal the twin's.  A copy of anything made at capture time shows up here."""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from dcd_amd.engine import trainer

dev = torch.device("cuda:0")
prec = os.environ.get("PREC", "bf16x3")
def build():
    args = argparse.Namespace(batch=8, objects=6, precision=prec, scaling="weak", amp=False)
    return bench.build_everything(args, dev, 1, 0)[:5]
cfg, model, optimizer, images, targets = build()
_, twin, _, _, _ = build()
for g in optimizer.param_groups:
    g["lr"].fill_(0.0)
    g["weight_decay"] = 0.0
clip = cfg.SOLVER.GRAD_NORM_CLIP
step = trainer.GraphedTrainStep(model, optimizer, clip)
step(images, targets)
step(images, targets)
gen = torch.Generator(device=dev).manual_seed(1)
scale_w = float(os.environ.get("EPS", "0.02"))
with torch.no_grad():
    for p in model.parameters():
        p.add_(torch.randn(p.shape, generator=gen, device=dev) * p.abs().mean() * scale_w)
torch.cuda.synchronize()
twin.load_state_dict(model.state_dict())
twin.zero_grad(set_to_none=True)
ld_t, _ = twin(images, targets)
sum(ld_t.values()).backward()
tg = {n: p.grad.detach().clone() for n, p in twin.named_parameters() if p.grad is not None}
tn = torch.linalg.vector_norm(torch.stack([g.norm() for g in tg.values()]))
sc = float((clip / (tn + 1e-6)).clamp(max=1.0))
ld, _ = step(images, targets)
gg = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
print("loss graph %.6f twin %.6f" % (float(sum(float(v) for v in ld.values())), float(sum(float(v) for v in ld_t.values()))))
rows = sorted(((float((gg[n] - tg[n] * sc).abs().max() / (tg[n] * sc).abs().max().clamp_min(1e-20)), n) for n in tg
               if n in gg and not n.endswith("conv.bias") and float(tg[n].abs().max()) > 1e-7), reverse=True)
for v, n in rows[:12]:
    print("  %.2e  %s" % (v, n))
print("  median %.1e   trunk1 max %.1e" % (rows[len(rows) // 2][0], max([v for v, n in rows if "reg_features.1." in n] + [0.0])))
