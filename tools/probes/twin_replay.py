"""After twin_catch.py: the gradients of the DCN offset convolutions from the caught weights, eagerly, under the DCD_DCN_HANDOVER
mode of the environment (argv: tag), twice (repeatability); saved to /tmp/twin_grads_<tag>.pt."""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from dcd_amd.engine import trainer
dev = torch.device("cuda:0")
tag = sys.argv[1]
c = torch.load("/tmp/twin_catch.pt")
args = argparse.Namespace(batch=8, objects=6, precision="bf16x3", scaling="weak", amp=False)
cfg, B, optB, images, targets = bench.build_everything(args, dev, 1, 0)[:5]
for g in optB.param_groups:
    g["lr"].fill_(0.0)
    g["weight_decay"] = 0.0
runs = []
for r in range(3):
    B.load_state_dict(c["state"])
    trainer.train_step(B, optB, images, targets, cfg.SOLVER.GRAD_NORM_CLIP)
    torch.cuda.synchronize()
    runs.append({n: p.grad.detach().cpu().clone() for n, p in B.named_parameters() if p.grad is not None})
torch.save(runs, "/tmp/twin_grads_%s.pt" % tag)
n = c["name"]
def rel(a, b):
    return float((a - b).abs().max() / b.abs().max())
print(tag, "caught tensor", n, "| run1 vs run0 %.2e, run2 vs run0 %.2e | vs caught twin %.2e | vs caught graph %.2e" % (
    rel(runs[1][n], runs[0][n]), rel(runs[2][n], runs[0][n]), rel(runs[0][n], c["twin"][n]), rel(runs[0][n], c["graph"][n])))
