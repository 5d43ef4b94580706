"""Graphed TRAINING on instance A; every 5 steps an eager instance B (learning rate zero) takes A's parameters and buffers by
in-place copies and runs trainer.train_step: its (clipped) gradients against the gradients of A's next replay."""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from dcd_amd.engine import trainer

N = int(os.environ.get("N", "61"))
dev = torch.device("cuda:0")
def build(zero_lr):
    args = argparse.Namespace(batch=8, objects=6, precision=os.environ.get("PREC", "bf16x3"), scaling="weak", amp=False)
    r = bench.build_everything(args, dev, 1, 0)[:5]
    if zero_lr:
        for g in r[2].param_groups:
            g["lr"].fill_(0.0)
            g["weight_decay"] = 0.0
    return r
cfg, A, optA, images, targets = build(False)
_, B, optB, _, _ = build(True)
clip = cfg.SOLVER.GRAD_NORM_CLIP
step = trainer.GraphedTrainStep(A, optA, clip)
for it in range(N):
    check = it % 5 == 0 and it > 0
    if check:
        torch.cuda.synchronize()
        with torch.no_grad():
            for p, q in zip(A.parameters(), B.parameters()):
                q.copy_(p)
            for p, q in zip(A.buffers(), B.buffers()):
                q.copy_(p)
        ldB, _ = trainer.train_step(B, optB, images, targets, clip)
        torch.cuda.synchronize()
        gB = {n: p.grad.detach().clone() for n, p in B.named_parameters() if p.grad is not None}
        lB = float(sum(float(v) for v in ldB.values()))
    ld, _ = step(images, targets)
    total = float(sum(float(v) for v in ld.values()))
    if check:
        gA = {n: p.grad.detach().clone() for n, p in A.named_parameters() if p.grad is not None}
        rows = sorted(((float((gA[n] - gB[n]).abs().max() / gB[n].abs().max().clamp_min(1e-20)), n) for n in gB
                       if n in gA and not n.endswith("conv.bias") and float(gB[n].abs().max()) > 1e-7), reverse=True)
        import dcd_amd
        taps = getattr(dcd_amd, "_DEBUG_TAPS", {})
        ha, hb = id(A.heads.predictor), id(B.heads.predictor)
        for nm in ("at_extra", "at_centres", "edge", "reg_pois"):
            if (ha, nm) in taps and (hb, nm) in taps:
                va, vb, ga, gb = taps[(ha, nm)][0], taps[(hb, nm)][0], taps[(ha, nm)][1], taps[(hb, nm)][1]
                print("      %-10s value diff %.1e (max %.2e) | grad: graph max %.3e eager max %.3e diff %.1e" % (
                    nm, float((va - vb).abs().max()), float(vb.abs().max()), float(ga.abs().max()), float(gb.abs().max()), float((ga - gb).abs().max())))
        print("step %2d loss graph %.5f eager-twin %.5f | worst: %s | median %.1e" % (
            it, total, lB, "  ".join("%.1e %s" % (v, n[-40:]) for v, n in rows[:3]), rows[len(rows) // 2][0]), flush=True)
