"""Where the HOST spends a train step (no device sync inside the step): forward / loss / backward / clip+optimizer enqueue
times, and the wall time.  python tools/cpu_phase_times.py [--batch 8]   (DCD_LOSS_GRAPH=0 to compare)"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--objects", type=int, default=6)
    ap.add_argument("--precision", default="f32")
    ap.add_argument("--scaling", default="weak")
    args = ap.parse_args()
    import torch
    import bench
    from dcd_amd.engine import trainer
    device = torch.device("cuda", 0)
    cfg, model, optimizer, images, targets, per_rank = bench.build_everything(args, device, 1, 0)[:6]
    clip = cfg.SOLVER.GRAD_NORM_CLIP
    for _ in range(4):
        trainer.train_step(model, optimizer, images, targets, clip)
    torch.cuda.synchronize()
    acc = [0.0] * 5
    t_all = time.perf_counter()
    for _ in range(args.steps):
        t0 = time.perf_counter()
        from dcd_amd.structures.image_list import to_image_list
        features = model.backbone(to_image_list(images).tensors)
        t1 = time.perf_counter()
        loss_dict, _ = model.heads(features, targets)
        t2 = time.perf_counter()
        losses = sum(loss_dict.values())
        optimizer.zero_grad(set_to_none=True)
        losses.backward()
        t3 = time.perf_counter()
        trainer.clip_grad_norm(trainer._parameters_of(model), clip)
        optimizer.step()
        t4 = time.perf_counter()
        for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
            acc[i] += d
    t_enq = time.perf_counter() - t_all
    torch.cuda.synchronize()
    wall = time.perf_counter() - t_all
    n = args.steps
    print("batch %d graph=%s: host ms/step backbone %.2f  heads+loss %.2f  backward %.2f  clip+adam %.2f | enqueue %.2f  wall %.2f" % (
        per_rank, os.environ.get("DCD_LOSS_GRAPH", "1"), 1e3 * acc[0] / n, 1e3 * acc[1] / n, 1e3 * acc[2] / n, 1e3 * acc[3] / n,
        1e3 * t_enq / n, 1e3 * wall / n))


if __name__ == "__main__":
    main()
