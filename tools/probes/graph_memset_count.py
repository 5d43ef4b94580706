"""Count hipMemsetAsync calls issued while the train step is being captured (torch.profiler sees the runtime API calls of the capture)."""
import argparse, os, sys, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from dcd_amd.engine import trainer

dev = torch.device("cuda:0")
args = argparse.Namespace(batch=int(os.environ.get("B", "8")), objects=6, precision=os.environ.get("PREC", "f32"), scaling="weak", amp=False)
cfg, model, optimizer, images, targets = bench.build_everything(args, dev, 1, 0)[:5]
step = trainer.GraphedTrainStep(model, optimizer, cfg.SOLVER.GRAD_NORM_CLIP)
# warm everything up eagerly first so that the profiled region is the capture's warm-up + record only
trainer.train_step(model, optimizer, images, targets, cfg.SOLVER.GRAD_NORM_CLIP)
torch.cuda.synchronize()
orig = step._record_graph
counts = {}
def record(st_images, st_targets):
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        out = orig(st_images, st_targets)
    c = collections.Counter(e.name for e in prof.events() if "emset" in e.name or "emcpy" in e.name)
    counts.update(c)
    return out
step._record_graph = record
step(images, targets)
torch.cuda.synchronize()
print("memset / memcpy events during the capture:", dict(counts))
