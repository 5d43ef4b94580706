"""Host-side cost of one train step: how long the CPU needs to ENQUEUE a step (no sync until the end) against the
step's wall time, and a cProfile of the enqueue.  At 1 image per GPU (the 8-GPU strong-scaling point) the step is
launch-bound, so this is the profile that matters there.

    python tools/cpu_profile_step.py --batch 1 [--steps 10] [--top 45]       (DCD_FORCE_DDP=1 for the SyncBN/DDP path)
"""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--objects", type=int, default=6)
    ap.add_argument("--top", type=int, default=45)
    ap.add_argument("--precision", default="f32")
    ap.add_argument("--scaling", default="weak")
    args = ap.parse_args()

    import torch
    import bench
    from dcd_amd.engine.trainer import train_step

    force_ddp = os.environ.get("DCD_FORCE_DDP", "0") == "1"
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    if force_ddp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=device)
    cfg, model, optimizer, images, targets, per_rank = bench.build_everything(args, device, 1, 0)[:6]
    clip = cfg.SOLVER.GRAD_NORM_CLIP
    for _ in range(args.warmup):
        train_step(model, optimizer, images, targets, clip)
    torch.cuda.synchronize()

    t0 = time.perf_counter()
    for _ in range(args.steps):
        train_step(model, optimizer, images, targets, clip)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("batch %d  ddp %d: enqueue %.2f ms/step, wall %.2f ms/step" % (
        per_rank, int(force_ddp), 1e3 * t_enq / args.steps, 1e3 * t_all / args.steps))

    pr = cProfile.Profile()
    pr.enable()
    for _ in range(args.steps):
        train_step(model, optimizer, images, targets, clip)
    pr.disable()
    torch.cuda.synchronize()
    for key in ("tottime", "cumtime"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(args.top)
        print("==== by %s (totals over %d steps) ====" % (key, args.steps))
        print("\n".join(s.getvalue().splitlines()[4:]))
    if force_ddp:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
