"""Which conv_offset_mask initialisation gives which sampling offsets in the benchmarked model (bench.py --offset-std-2px): mean |offset|,
far fraction and step time after a few train steps, per std.   python tools/offset_std_probe.py 0.033 0.06 0.1"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dcd_amd import _ext
from dcd_amd.engine.trainer import train_step

dev = torch.device("cuda:0")
for std in [float(v) for v in sys.argv[1:]] or [0.01, 0.033, 0.06, 0.1]:
    args = argparse.Namespace(batch=8, objects=6, precision="f32", scaling="strong", amp=False, offset_std=std)
    cfg, model, opt, images, targets = bench.build_everything(args, dev, 1, 0)[:5]
    clip = cfg.SOLVER.GRAD_NORM_CLIP
    row = []
    for phase in range(3):
        with bench.OffsetStats(torch, _ext) as st, torch.no_grad():
            model(images, targets)
        s = st.summary()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            train_step(model, opt, images, targets, clip)
        torch.cuda.synchronize()
        row.append("after %2d steps: mean %.3f px, far %.4f, then %.1f ms/step" % (5 * phase, s["mean_abs_px"], s["far_fraction"], (time.perf_counter() - t0) * 200))
    print("std %.3f | " % std + " | ".join(row), flush=True)
    del model, opt
