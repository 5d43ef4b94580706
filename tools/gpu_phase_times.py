"""GPU time of the phases of a train step (events on the launch stream; no synchronisation inside the step):
backbone forward | heads + loss forward | backward | clip + AdamW.   python tools/gpu_phase_times.py [--batch 8]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--objects", type=int, default=6)
    ap.add_argument("--precision", default="f32")
    ap.add_argument("--scaling", default="weak")
    ap.add_argument("--amp", action="store_true")
    ap.add_argument("--presleep-ms", type=float, default=0.0,
                    help="spin the GPU this long before every step, so the host runs ahead: phases become pure GPU time")
    args = ap.parse_args()
    import torch
    import bench
    from dcd_amd.engine import trainer
    from dcd_amd.structures.image_list import to_image_list
    device = torch.device("cuda", 0)
    cfg, model, optimizer, images, targets, per_rank = bench.build_everything(args, device, 1, 0)[:6]
    clip = cfg.SOLVER.GRAD_NORM_CLIP
    for _ in range(4):
        trainer.train_step(model, optimizer, images, targets, clip)
    torch.cuda.synchronize()
    marks, fines = [], []
    for _ in range(args.steps):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        if args.presleep_ms > 0:
            torch.cuda._sleep(int(args.presleep_ms * 2.4e6))
        ev[0].record()
        features = model.backbone(to_image_list(images).tensors)
        ev[1].record()
        fine = [torch.cuda.Event(enable_timing=True) for _ in range(3)]     # after predictor | loss bwd done | heads bwd done
        features.register_hook(lambda g, e=fine[2]: e.record())
        preds_ = model.heads.predictor(features, targets)
        fine[0].record()
        first = [True]

        def after_loss_bwd(g, e=fine[1]):
            if first[0]:
                first[0] = False
                e.record()
        for v in preds_.values():
            if torch.is_tensor(v) and v.requires_grad:
                v.register_hook(after_loss_bwd)
        loss_dict, _ = model.heads.loss_evaluator(preds_, targets)
        fines.append(fine)
        total = getattr(loss_dict, "total", None)
        losses = total if total is not None else sum(loss_dict.values())
        ev[2].record()
        optimizer.zero_grad(set_to_none=True)
        losses.backward()
        ev[3].record()
        trainer.guard_nonfinite_step(optimizer, trainer.clip_grad_norm(trainer._parameters_of(model), clip))
        optimizer.step()
        ev[4].record()
        marks.append(ev)
    torch.cuda.synchronize()
    # the loss section alone (forward graph, backward graph) on fixed predictions
    heads = model.heads
    features = model.backbone(to_image_list(images).tensors)
    predictions = heads.predictor(features, targets)
    preds = {k: (v.detach().requires_grad_(True) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in predictions.items()}
    lf = lb = 0.0
    for it in range(args.steps + 2):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        ld, _ = heads.loss_evaluator(preds, targets)
        total = getattr(ld, "total", None)
        total = total if total is not None else sum(ld.values())
        e[1].record()
        total.backward()
        e[2].record()
        torch.cuda.synchronize()
        if it >= 2:
            lf += e[0].elapsed_time(e[1]) / args.steps
            lb += e[1].elapsed_time(e[2]) / args.steps
    print("loss section alone: forward %.2f ms, backward %.2f ms (GPU, events around the calls; host-bound if eager)" % (lf, lb))
    n = len(marks)
    pf = sum(m[1].elapsed_time(f[0]) for m, f in zip(marks, fines)) / n
    lfw = sum(f[0].elapsed_time(m[2]) for m, f in zip(marks, fines)) / n
    lbw = sum(m[2].elapsed_time(f[1]) for m, f in zip(marks, fines)) / n
    hbw = sum(f[1].elapsed_time(f[2]) for m, f in zip(marks, fines)) / n
    bbw = sum(f[2].elapsed_time(m[3]) for m, f in zip(marks, fines)) / n
    print("  predictor fwd %.2f | loss fwd %.2f | loss bwd (first gradient out) %.2f | predictor bwd %.2f | backbone bwd %.2f" % (
        pf, lfw, lbw, hbw, bbw))
    acc = [sum(e[i].elapsed_time(e[i + 1]) for e in marks) / n for i in range(4)]
    print("batch %d: GPU ms/step  backbone fwd %.2f | heads+loss fwd %.2f | backward %.2f | clip+adam %.2f | sum %.2f" % (
        per_rank, acc[0], acc[1], acc[2], acc[3], sum(acc)))


if __name__ == "__main__":
    main()
