"""Does a replayed step see the CURRENT weights?  Every K replays: an eager forward on the model's present weights, then the next
replay's in-graph loss (computed from the same weights before its update): the 13 terms must agree."""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from dcd_amd.engine import trainer

N = int(os.environ.get("N", "61"))
dev = torch.device("cuda:0")
args = argparse.Namespace(batch=8, objects=6, precision=os.environ.get("PREC", "bf16x3"), scaling="weak", amp=False)
cfg, model, optimizer, images, targets = bench.build_everything(args, dev, 1, 0)[:5]
clip = cfg.SOLVER.GRAD_NORM_CLIP
step = trainer.GraphedTrainStep(model, optimizer, clip)
names = None
for it in range(N):
    if it % 10 == 0 and it > 0:
        with torch.no_grad():
            ld_e, _ = model(images, targets)
        e = {k: float(v) for k, v in ld_e.items()}
    ld, _ = step(images, targets)
    g = {k: float(v) for k, v in ld.items()}
    if it % 10 == 0 and it > 0:
        names = names or list(g)
        print("replay %2d  eager-forward total %.6f  in-graph total %.6f | per-term |diff|: %s" % (
            it, sum(e.values()), sum(g.values()), " ".join("%s %.1e" % (n.replace("_loss", ""), abs(e[n] - g[n])) for n in names)), flush=True)
