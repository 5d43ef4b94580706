"""One-rank data-parallel soak (SyncBN + gradient all-reduce over a world-size-1 RCCL group): the graphed distributed step and the
eager DDP step against the plain eager step, 150 steps over four batches."""
import argparse, os, sys
os.environ["DCD_FORCE_DDP"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import torch.distributed as dist
import bench
from dcd_amd.data.synthetic import make_batch
from dcd_amd.engine import trainer

N = int(os.environ.get("N", "151"))
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
for mode in ("ddp_graph", "ddp_eager", "plain"):
    os.environ["DCD_FORCE_DDP"] = "0" if mode == "plain" else "1"
    os.environ["DCD_STEP_GRAPH"] = "1" if mode == "ddp_graph" else "0"
    args = argparse.Namespace(batch=8, objects=6, precision="f32", scaling="weak", amp=False)
    cfg, model, optimizer, images, targets, per_rank, use_graph, data_parallel = bench.build_everything(args, dev, 1, 0)
    clip = cfg.SOLVER.GRAD_NORM_CLIP
    batches = [make_batch(8, seed=200 + i, n_objects=6, device=dev) for i in range(4)]
    step = trainer.GraphedTrainStep(model, optimizer, clip, distributed=True) if mode == "ddp_graph" else None
    hist = []
    for it in range(N):
        im, tg = batches[it % 4]
        ld, _ = step(im, tg) if step else trainer.train_step(model, optimizer, im, tg, clip)
        if it % 30 == 0:
            total = getattr(ld, "total", None)
            hist.append((it, float(total if total is not None else sum(ld.values()))))
    print("%-10s use_graph=%s data_parallel=%s  %s" % (mode, use_graph, data_parallel, " ".join("%d:%.3f" % h for h in hist)), flush=True)
    del step, model, optimizer
    torch.cuda.empty_cache()
dist.destroy_process_group()
