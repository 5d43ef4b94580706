"""Overfit one synthetic batch for N steps in both precisions: the summed loss must fall and stay finite (a sanity run of the whole
train step -- kernels, fused loss, optimizer -- not a benchmark)."""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from dcd_amd import _ext
from dcd_amd.engine import trainer

N = int(os.environ.get("N", "120"))
for prec in ("f32", "bf16x3", "amp"):                       # amp: MODEL.FP16 (the bf16 precision scope), round 5
    args = argparse.Namespace(batch=4, objects=6, precision="f32" if prec == "amp" else prec, scaling="weak", amp=prec == "amp")
    dev = torch.device("cuda:0")
    cfg, model, optimizer, images, targets = bench.build_everything(args, dev, 1, 0)[:5]
    clip = cfg.SOLVER.GRAD_NORM_CLIP
    hist = []
    for it in range(N):
        loss_dict, log = trainer.train_step(model, optimizer, images, targets, clip)
        if it % 20 == 0 or it == N - 1 or it < 4:
            total = sum(float(v) for v in loss_dict.values())
            hist.append((it, total))
            assert total == total and abs(total) != float("inf"), (prec, it, total)
    print(prec, " ".join("%d:%.3f" % h for h in hist))
    assert hist[-1][1] < 0.7 * hist[0][1], (prec, hist)
_ext.set_precision("f32")
print("ok")
