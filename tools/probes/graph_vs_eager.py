"""Eager vs graphed train step on the same cycle of four synthetic batches: the loss of every step."""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from dcd_amd.data.synthetic import make_batch
from dcd_amd.engine import trainer

N = int(os.environ.get("N", "60"))
NB = int(os.environ.get("NB", "4"))
dev = torch.device("cuda:0")
res = {}
MODES = os.environ.get("MODES", "eager,graph").split(",")
for mode in MODES:
    args = argparse.Namespace(batch=8, objects=6, precision="f32", scaling="weak", amp=False)
    cfg, model, optimizer, images, targets = bench.build_everything(args, dev, 1, 0)[:5]
    clip = cfg.SOLVER.GRAD_NORM_CLIP
    batches = [make_batch(8, seed=200 + i, n_objects=6, device=dev) for i in range(NB)]
    rec = int(os.environ.get("RECAP", "0")) or None
    step = trainer.GraphedTrainStep(model, optimizer, clip, recapture_every=rec) if mode == "graph" else None
    hist = []
    for it in range(N):
        im, tg = batches[it % NB]
        ld, _ = step(im, tg) if step else trainer.train_step(model, optimizer, im, tg, clip)
        total = getattr(ld, "total", None)
        hist.append(float(total if total is not None else sum(ld.values())))
    res[mode] = hist
    del step, model, optimizer
    torch.cuda.empty_cache()
for it in range(N):
    if it % 5 == 0:
        print("step %3d  %s" % (it, "  ".join("%s %.4f" % (m, res[m][it]) for m in MODES)), flush=True)
