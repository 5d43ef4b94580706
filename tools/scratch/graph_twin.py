"""Graphed training (lr > 0) with an eager TWIN: every 5 steps the twin takes the graph model's present weights, computes its
gradients eagerly on the same batch, and the next replay's gradients (computed from the same weights) are compared with them."""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from dcd_amd.engine import trainer

N = int(os.environ.get("N", "61"))
dev = torch.device("cuda:0")
prec = os.environ.get("PREC", "bf16x3")
def build():
    args = argparse.Namespace(batch=8, objects=6, precision=prec, scaling="weak", amp=False)
    return bench.build_everything(args, dev, 1, 0)[:5]
cfg, model, optimizer, images, targets = build()
_, twin, _, _, _ = build()
clip = cfg.SOLVER.GRAD_NORM_CLIP
step = trainer.GraphedTrainStep(model, optimizer, clip)
for it in range(N):
    check = it % 5 == 0 and it > 0
    if check:
        torch.cuda.synchronize()
        twin.load_state_dict(model.state_dict())
        twin.zero_grad(set_to_none=True)
        ld_t, _ = twin(images, targets)
        sum(ld_t.values()).backward()
        tg = {n: p.grad.detach().clone() for n, p in twin.named_parameters() if p.grad is not None}
        tn = torch.linalg.vector_norm(torch.stack([g.norm() for g in tg.values()]))
        scale = float((clip / (tn + 1e-6)).clamp(max=1.0))
    ld, _ = step(images, targets)
    total = float(sum(float(v) for v in ld.values()))
    if check:
        gg = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        rows = sorted(((float((gg[n] - tg[n] * scale).abs().max() / (tg[n] * scale).abs().max().clamp_min(1e-20)), n) for n in tg
                       if n in gg and not n.endswith("conv.bias") and float(tg[n].abs().max()) > 1e-6), reverse=True)
        if os.environ.get("COMPACT"):
            print("step %2d worst %.1e %s | trunk1 %.1e" % (it, rows[0][0], rows[0][1][-40:], max([v for v, n in rows if "reg_features.1." in n] + [0.0])), flush=True)
            continue
        print("step %2d loss graph %.5f twin %.5f | worst grad mismatch: %s | median %.1e" % (
            it, total, float(sum(float(v) for v in ld_t.values())), "  ".join("%.1e %s" % (v, n[-44:]) for v, n in rows[:4]), rows[len(rows) // 2][0]), flush=True)
