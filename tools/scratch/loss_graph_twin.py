"""The nested HIP graph of the LOSS section (eager steps replay it every step: detector_loss._call_graphed) against the same loss
evaluated op by op, during eager training: instance A trains with the loss graph, instance B (learning rate zero, loss graph off)
takes A's weights every five steps; A's next step's gradients against B's."""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from dcd_amd.engine import trainer

N = int(os.environ.get("N", "41"))
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
for m in B.modules():
    if hasattr(getattr(m, "loss_evaluator", None), "use_graph"):
        m.loss_evaluator.use_graph = False
clip = cfg.SOLVER.GRAD_NORM_CLIP
for it in range(N):
    check = it % 5 == 0 and it > 0
    if check:
        with torch.no_grad():
            for p, q in zip(A.parameters(), B.parameters()):
                q.copy_(p)
            for p, q in zip(A.buffers(), B.buffers()):
                q.copy_(p)
        ldB, _ = trainer.train_step(B, optB, images, targets, clip)
        torch.cuda.synchronize()
        gB = {n: p.grad.detach().clone() for n, p in B.named_parameters() if p.grad is not None}
    w_before = None
    ld, _ = trainer.train_step(A, optA, images, targets, clip)
    if check:
        torch.cuda.synchronize()
        gA = {n: p.grad.detach().clone() for n, p in A.named_parameters() if p.grad is not None}
        rows = sorted(((float((gA[n] - gB[n]).abs().max() / gB[n].abs().max().clamp_min(1e-20)), n) for n in gB
                       if n in gA and not n.endswith("conv.bias") and float(gB[n].abs().max()) > 1e-7), reverse=True)
        print("step %2d loss graphed-loss %.6f plain %.6f | worst: %s | median %.1e" % (
            it, float(sum(float(v) for v in ld.values())), float(sum(float(v) for v in ldB.values())),
            "  ".join("%.1e %s" % (v, n[-36:]) for v, n in rows[:3]), rows[len(rows) // 2][0]), flush=True)
