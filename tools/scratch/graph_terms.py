"""Graphed step on one batch: the 13 loss terms every few steps (which one leaves the eager trajectory?)."""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from dcd_amd.engine import trainer

N = int(os.environ.get("N", "70"))
dev = torch.device("cuda:0")
mode = os.environ.get("MODE", "graph")
args = argparse.Namespace(batch=8, objects=6, precision="f32", scaling="weak", amp=False)
cfg, model, optimizer, images, targets = bench.build_everything(args, dev, 1, 0)[:5]
clip = cfg.SOLVER.GRAD_NORM_CLIP
step = trainer.GraphedTrainStep(model, optimizer, clip) if mode == "graph" else None
names = None
for it in range(N):
    ld, log = step(images, targets) if step else trainer.train_step(model, optimizer, images, targets, clip)
    if it % 5 == 0 or 36 <= it <= 50:
        d = {k: float(v) for k, v in ld.items()}
        if names is None:
            names = list(d)
            print("terms:", " ".join(n.replace("_loss", "") for n in names))
        wn = sum(float(p.detach().float().pow(2).sum()) for p in model.parameters()) ** 0.5
        off = max(float(p.detach().abs().max()) for n, p in model.named_parameters() if "conv_offset_mask.weight" in n)
        print("%3d total %.3f | %s | |w| %.3f max|w_off| %.4f" % (it, sum(d.values()), " ".join("%.3f" % d[n] for n in names), wn, off), flush=True)
