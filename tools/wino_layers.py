"""Per-shape timing of the Winograd 3x3 convolutions of one train step: records the shapes `ops.conv3x3` is called with, then
times forward / backward-data / weight gradient of every distinct shape alone (events, 20 runs) and prints the matrix-pipe
share: direct-convolution flops / 2.25 (the multiplies Winograd actually issues) over 157.3 TFLOP/s.
python tools/wino_layers.py [--batch 8]"""
import argparse, collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--objects", type=int, default=6)
    ap.add_argument("--precision", default="f32")
    ap.add_argument("--scaling", default="weak")
    ap.add_argument("--amp", action="store_true")
    args = ap.parse_args()
    import torch
    import bench
    from dcd_amd import ops
    from dcd_amd.engine import trainer
    device = torch.device("cuda", 0)
    cfg, model, optimizer, images, targets = bench.build_everything(args, device, 1, 0)[:5]
    seen = collections.Counter()
    orig = ops._Conv3x3.forward

    def spy(ctx, x, weight):
        seen[(x.shape[0], weight.shape[1], weight.shape[0], x.shape[2], x.shape[3], bool(ctx.needs_input_grad[0]))] += 1
        return orig(ctx, x, weight)
    ops._Conv3x3.forward = staticmethod(spy)
    trainer.train_step(model, optimizer, images, targets, cfg.SOLVER.GRAD_NORM_CLIP)
    ops._Conv3x3.forward = staticmethod(orig)
    torch.cuda.synchronize()

    def timed(fn, n=20):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    tot = [0.0, 0.0, 0.0]
    print("%-30s %3s %8s %6s %8s %6s %8s %6s" % ("B Ci Co H W", "n", "fwd ms", "pipe%", "bwdD ms", "pipe%", "wrw ms", "pipe%"))
    for (B, Ci, Co, H, W, need_gx), n in sorted(seen.items(), key=lambda kv: -kv[0][3] * kv[0][4]):
        x = torch.randn(B, Ci, H, W, device=device)
        w = torch.randn(Co, Ci, 3, 3, device=device) * 0.05
        gy = torch.randn(B, Co, H, W, device=device)
        tf, tb = ops.conv3x3_transform_weights(w)
        flops = 2.0 * B * H * W * Ci * Co * 9
        pipe = lambda ms: 100.0 * flops / 2.25 / (ms * 1e-3) / 157.3e12
        f = timed(lambda: ops._conv3x3_call(x, w, Co, False, transformed=tf))
        d = timed(lambda: ops._conv3x3_call(gy, w, Ci, True, transformed=tb)) if need_gx else 0.0
        g = timed(lambda: ops._conv3x3_wrw_call(x, gy, w.shape))
        tot[0] += n * f; tot[1] += n * d; tot[2] += n * g
        print("%-30s %3d %8.3f %6.1f %8.3f %6.1f %8.3f %6.1f" % ("%d %d %d %d %d" % (B, Ci, Co, H, W), n, f, pipe(f),
                                                                  d, pipe(d) if d else 0.0, g, pipe(g)))
    print("per step: forward %.2f ms, backward-data %.2f ms, weight gradient %.2f ms" % tuple(tot))


if __name__ == "__main__":
    main()
