"""Whole-step HIP graph (engine.trainer.GraphedTrainStep) against eager train_step at full size: same seed, same batch,
N steps each; prints the total loss per step for both and the largest relative difference of the summed loss and of the
parameters after the last step.   python tools/check_step_graph.py [--batch 8] [--steps 8]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(args, graphed):
    import torch
    import bench
    from dcd_amd.engine import trainer
    device = torch.device("cuda", 0)
    cfg, model, optimizer, images, targets, per_rank = bench.build_everything(args, device, 1, 0)[:6]
    clip = cfg.SOLVER.GRAD_NORM_CLIP
    step = trainer.GraphedTrainStep(model, optimizer, clip) if graphed else None
    losses = []
    for _ in range(args.steps):
        if graphed:
            ld, _ = step(images, targets)
        else:
            ld, _ = trainer.train_step(model, optimizer, images, targets, clip)
        total = getattr(ld, "total", None)
        losses.append(float(total if total is not None else sum(ld.values())))
    torch.cuda.synchronize()
    params = torch.cat([p.detach().flatten().double() for p in model.parameters()]).cpu()
    return losses, params


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--objects", type=int, default=6)
    ap.add_argument("--precision", default="f32")
    ap.add_argument("--scaling", default="weak")
    ap.add_argument("--amp", action="store_true")
    args = ap.parse_args()
    le, pe = run(args, False)
    lg, pg = run(args, True)
    le2, pe2 = run(args, False)          # eager run-to-run noise (fp32 atomics) as the yardstick
    for i, (a, b, c) in enumerate(zip(le, lg, le2)):
        print("step %d  eager %.6f  graph %.6f  eager(2nd run) %.6f" % (i, a, b, c))
    d_g = (pe - pg).norm().item() / pe.norm().item()
    d_e = (pe - pe2).norm().item() / pe.norm().item()
    print("relative parameter distance after %d steps: graph vs eager %.3e, eager vs eager %.3e" % (args.steps, d_g, d_e))


if __name__ == "__main__":
    main()
