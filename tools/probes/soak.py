"""Soak: N steps of the bs-8 step on changing synthetic batches -- eager fp32, graphed fp32, eager mixed precision: losses finite,
memory flat (allocator high-water mark after step 20 vs the end), no launch failures."""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from dcd_amd.data.synthetic import make_batch
from dcd_amd.engine import trainer

N = int(os.environ.get("N", "300"))
dev = torch.device("cuda:0")
for mode in os.environ.get("MODES", "eager,graph,amp,amp_graph").split(","):
    args = argparse.Namespace(batch=8, objects=6, precision="f32", scaling="weak", amp=mode.startswith("amp"))
    cfg, model, optimizer, images, targets = bench.build_everything(args, dev, 1, 0)[:5]
    clip = cfg.SOLVER.GRAD_NORM_CLIP
    batches = [make_batch(8, seed=200 + i, n_objects=6, device=dev) for i in range(4)]
    step = trainer.GraphedTrainStep(model, optimizer, clip) if mode.endswith("graph") else None
    hist, mem20 = [], None
    for it in range(N):
        im, tg = batches[it % 4]
        ld, _ = step(im, tg) if step else trainer.train_step(model, optimizer, im, tg, clip)
        if it % 50 == 0 or it == N - 1:
            total = getattr(ld, "total", None)
            v = float(total if total is not None else sum(ld.values()))
            assert v == v and abs(v) < 1e5, (mode, it, v)
            hist.append((it, v))
        if it == 20:
            torch.cuda.synchronize(); mem20 = torch.cuda.max_memory_allocated()
    torch.cuda.synchronize()
    mem = torch.cuda.max_memory_allocated()
    print("%-6s %s | max allocated after 20 steps %.2f GB, at the end %.2f GB" % (mode, " ".join("%d:%.3f" % h for h in hist), mem20 / 2**30, mem / 2**30), flush=True)
    assert mem <= 1.02 * mem20, (mem20, mem)
    del step, model, optimizer
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
print("ok")
