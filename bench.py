#!/usr/bin/env python3
"""DGDE train-step benchmark on MI355X:  images/sec at global batch 8, 384x1280, fp32 (BASELINE.json configs[1]).

  python bench.py --gpus N --steps K --warmup W
      N > 1: one rank per GPU, either launched by torch.distributed.run (RANK/WORLD_SIZE in the environment) or, when
      started bare, by bench.py itself (it spawns torch.distributed.run as a child before touching the GPU).

A step = forward + 13-term loss + backward + gradient all-reduce (RCCL, N>1) + grad clip + AdamW, on a synthetic
KITTI-shaped batch that is resident in HBM before the timed region.  The metric's batch is GLOBAL: with N > 1 the 8 images
are split over the ranks as the reference does (IMS_PER_BATCH // world, DGDE/data/build.py:63-67: one image per rank at
N = 8, north_star's partition), value = 8 images / max-over-ranks step time, "scaling": "strong" -- the default since
round 3.  The same run then also times 8 images PER rank (BASELINE.json configs[2] is that at N = 4: bs 32 on 4 GPUs) and
reports it under the extra key "weak_scaling"; `--scaling weak` makes that the headline instead (metric string says so).
At <= 2 images per rank the step is launch-bound and is replayed from ONE HIP graph that contains the SyncBN and gradient
collectives (engine.trainer.GraphedTrainStep(distributed=True)); DCD_STEP_GRAPH=0 keeps the eager DDP step.

One JSON line on rank 0.  Besides the driver's contract it carries
  roofline     -- DCNv2 forward+backward of the 16 DLA-34 DCN layers: GEMM flops (computed from the
                  layer list below) / time of those calls measured with events on the launch stream, inside the timed
                  steps; peak = the fp32 matrix peak (the op is MFMA-bound: 196 FLOP/B).  `roofline_hbm` repeats it with
                  the algorithmic bytes against 8 TB/s (north_star's yardstick); `traffic` = HBM bytes from the PMC pass.
  cpu_baseline -- the same train step on the host cores with the CPU oracle (oracle/, "port") on a bounded sample.
  split_bf16x3 -- (one GPU, default run only) the same job timed again with `_ext.set_precision("bf16x3")`: the DCNv2 products and the
                  3x3 convolutions' forward / input gradient in split-bf16 (hi*hi + hi*lo + lo*hi on the bf16 matrix cores, fp32
                  accumulate; <= 1e-4 of the output scale against the fp32 oracle, north_star's bound is 1e-3).  `value` is always
                  the exact-fp32 run.
"""
import argparse
import json
import os

import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# (Cin, Cout, H, W, count) of the 16 DCN layers at 384x1280
DCN_LAYERS = [(512, 256, 12, 40, 1), (256, 256, 24, 80, 1), (256, 128, 24, 80, 2), (128, 128, 48, 160, 2),
              (128, 64, 48, 160, 4), (64, 64, 96, 320, 5), (256, 64, 24, 80, 1)]
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec
MFMA_PEAK_TFLOPS = {"f32": 157.3, "bf16x3": 2500.0 / 3.0, "bf16": 2500.0}


def dcn_algorithmic(batch):
    """(bytes fwd+bwd, GEMM flops fwd+bwd) of the 16 layers: every operand read once, every result written once,
    no column buffer."""
    by = fl = 0
    for cin, cout, h, w, n in DCN_LAYERS:
        hw = h * w
        fwd = 4 * (batch * hw * (cin + 27 + cout) + 9 * cin * cout + cout)
        bwd = 4 * (batch * hw * (cin + 27 + cout) + 9 * cin * cout) + 4 * (batch * hw * (cin + 27) + 9 * cin * cout + cout)
        by += n * (fwd + bwd)
        fl += n * 3 * 2 * batch * cout * 9 * cin * hw
    return by, fl


class DcnTimer:
    """Wraps dcd_amd._ext.dcn_v2_forward/backward with event pairs recorded on the current (= launch) stream."""

    def __init__(self, torch, ext):
        self.torch, self.ext = torch, ext
        self.pairs = []
        self.enabled = False
        self._f, self._b = ext.dcn_v2_forward, ext.dcn_v2_backward
        ext.dcn_v2_forward = self._wrap(self._f, "fwd")
        ext.dcn_v2_backward = self._wrap(self._b, "bwd")

    def _wrap(self, fn, tag):
        def timed(*a, **k):
            if not self.enabled:
                return fn(*a, **k)
            e0, e1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **k)
            e1.record()
            x, w = a[0], a[1]                                   # input (B, Cin, H, W), weight (Cout, Cin, kh, kw)
            self.pairs.append((e0, e1, tag, (x.shape[1], w.shape[0], x.shape[2], x.shape[3])))
            return out
        return timed

    def total_ms(self):
        return sum(p[0].elapsed_time(p[1]) for p in self.pairs)

    def per_layer(self, steps, batch):
        """{"CinxCout@HxW": {"n": layers, "fwd_ms": per layer, "bwd_ms": per layer, "tflops": GEMM rate of fwd+bwd}}"""
        acc = {}
        for e0, e1, tag, geom in self.pairs:
            d = acc.setdefault(geom, {"fwd": 0.0, "bwd": 0.0, "calls": 0})
            d[tag] += e0.elapsed_time(e1)
            d["calls"] += tag == "fwd"
        out = {}
        for (cin, cout, h, w), d in acc.items():
            n = max(d["calls"] // max(steps, 1), 1)
            f, b = d["fwd"] / steps / n, d["bwd"] / steps / n
            fl = 3 * 2 * batch * cout * 9 * cin * h * w
            out["%dx%d@%dx%d" % (cin, cout, h, w)] = {"n": n, "fwd_ms": round(f, 4), "bwd_ms": round(b, 4),
                                                     "tflops": round(fl / 1e9 / (f + b), 2) if f + b > 0 else None}
        return out


class OffsetStats:
    """What sampling offsets the DCN number was measured at: one extra eager forward, outside the timed region,
    with the offset argument of every `dcn_v2_forward` call reduced on the device -- mean |d|, max |d| and the fraction of samples
    with a coordinate displaced by >= 3 px (what the one-pass backward's LDS window does not hold: `far`)."""

    def __init__(self, torch, ext):
        self.torch, self.ext = torch, ext
        self.rows = []
        self._f = ext.dcn_v2_forward

    def __enter__(self):
        def hooked(*a, **k):
            off = a[3].detach()
            n = off.shape[1] // 2
            ab = off.abs()
            pair = ab.view(off.shape[0], n, 2, -1).amax(dim=2)                     # per sample: the larger of |dy|, |dx|
            self.rows.append(self.torch.stack([ab.sum(), ab.amax(), (pair >= 3.0).sum().float(),
                                               self.torch.tensor(float(ab.numel()), device=off.device),
                                               self.torch.tensor(float(pair.numel()), device=off.device)]))
            return self._f(*a, **k)
        self.ext.dcn_v2_forward = hooked
        return self

    def __exit__(self, *exc):
        self.ext.dcn_v2_forward = self._f

    def summary(self):
        if not self.rows:
            return None
        t = self.torch.stack(self.rows).double().cpu()
        return {"mean_abs_px": round(float(t[:, 0].sum() / t[:, 3].sum()), 4), "max_abs_px": round(float(t[:, 1].max()), 3),
                "far_fraction": float(t[:, 2].sum() / t[:, 4].sum()), "far_threshold_px": 3.0, "calls": len(self.rows),
                "source": "offset tensors of the 16 DCN calls of one eager forward of the benchmarked model and batch "
                          "(conv_offset_mask ~ N(0, std^2), SURVEY 8d)"}


def op_level_dcn(torch, ext, batch, sigma, prec, iters=3):
    """SURVEY 8(d)'s op-level DCN bench: the 16 layers through the C ABI (dcd_amd._ext), input randn, offsets sigma * randn px
    (DGDE/model/backbone/DCNv2/DCN/testcuda.py:74 uses 2), mask sigmoid(randn), weight ~ 1 / sqrt(9 Cin).  (fwd ms, bwd ms)
    over the 16 layers, event-timed on the launch stream."""
    dev = torch.device("cuda", torch.cuda.current_device())
    a = (3, 3, 1, 1, 1, 1, 1, 1, 1)
    g = torch.Generator(device=dev).manual_seed(1234)
    tf = tb = 0.0
    for cin, cout, h, w, n in DCN_LAYERS:
        x = torch.randn(batch, cin, h, w, device=dev, generator=g)
        off = torch.randn(batch, 18, h, w, device=dev, generator=g) * sigma
        m = torch.sigmoid(torch.randn(batch, 9, h, w, device=dev, generator=g))
        wt = torch.randn(cout, cin, 3, 3, device=dev, generator=g) / (cin * 9) ** 0.5
        b = torch.zeros(cout, device=dev)
        gy = torch.randn(batch, cout, h, w, device=dev, generator=g)
        for _ in range(2):
            ext.dcn_v2_forward(x, wt, b, off, m, *a, precision=prec)
            ext.dcn_v2_backward(x, wt, b, off, m, gy, *a, precision=prec)
            # the backward's host-side hand-over policy reads the far count its layer's PREVIOUS call reported (dcn_v2.hip): let
            # the warm-up calls finish, as a train step's calls of one layer are a whole step apart
            torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        for _ in range(iters):
            ext.dcn_v2_forward(x, wt, b, off, m, *a, precision=prec)
        e[1].record()
        for _ in range(iters):
            ext.dcn_v2_backward(x, wt, b, off, m, gy, *a, precision=prec)
        e[2].record()
        torch.cuda.synchronize()
        tf += n * e[0].elapsed_time(e[1]) / iters
        tb += n * e[1].elapsed_time(e[2]) / iters
    return tf, tb


def op_level_lines(torch, ext, batch, prec):
    """`roofline_op_2px` (+ the same layers at 0.5 px, the ratio the review asks for)."""
    by, fl = dcn_algorithmic(batch)
    out = {}
    for key, sigma in (("roofline_op_0p5px", 0.5), ("roofline_op_1px", 1.0), ("roofline_op_2px", 2.0)):
        tf, tb = op_level_dcn(torch, ext, batch, sigma, prec)
        t = tf + tb
        out[key] = {"bound": "mfma", "kernel": "DCNv2 fwd+bwd, 16 layers through the C ABI, batch %d, offsets %g * randn px" % (batch, sigma),
                    "fwd_ms": round(tf, 3), "bwd_ms": round(tb, 3), "ms": round(t, 3), "achieved": fl / 1e9 / t, "peak": MFMA_PEAK_TFLOPS[prec],
                    "unit": "TFLOP/s", "frac": fl / 1e9 / t / MFMA_PEAK_TFLOPS[prec], "hbm_frac": by / 1e6 / t / HBM_PEAK_GBS,
                    "offset_sigma_px": sigma}
    for key in ("roofline_op_1px", "roofline_op_2px"):
        out[key]["ratio_to_0p5px"] = round(out[key]["ms"] / out["roofline_op_0p5px"]["ms"], 3)
    return out


def build_everything(args, device, world, local_rank):
    import torch
    from dcd_amd import _ext
    from dcd_amd.config import get_cfg
    from dcd_amd.data.synthetic import make_batch
    from dcd_amd.engine.trainer import build_optimizer, init_like_trained, prepare_data_parallel, wrap_distributed
    from dcd_amd.model.detector import KeypointDetector

    _ext.set_precision(args.precision)
    force_ddp = os.environ.get("DCD_FORCE_DDP", "0") == "1"
    in_w, in_h = (int(v) for v in getattr(args, "input", "1280x384").split("x"))
    cfg = get_cfg(opts=["MODEL.PRETRAIN", False, "MODEL.DEVICE", str(device), "MODEL.USE_SYNC_BN", world > 1 or force_ddp,
                        "MODEL.FP16", bool(args.amp), "INPUT.WIDTH_TRAIN", in_w, "INPUT.HEIGHT_TRAIN", in_h])
    torch.manual_seed(0)
    model = KeypointDetector(cfg)
    init_like_trained(model, std=getattr(args, "offset_std", 0.01), seed=0)
    model = model.to(device).train()
    optimizer = build_optimizer(model, cfg)
    per_rank = args.batch if args.scaling == "weak" else max(args.batch // world, 1)
    # Whole-step HIP graph: at <= 4 images per rank the step is launch-bound (round 5: bs 2 19.0 -> 16.2 ms, bs 4 24.2 -> 23.3; bs 8
    # 38.0 eager against 38.4 graphed: the GPU is the bound there and the eager host runs ahead);
    # under data parallelism at EVERY batch -- the 136 SyncBN all-reduces issued from Python cost the eager step 6-11 ms
    # (DDP + SyncBN forced on one GPU: bs 8 52.3 -> 46.5 ms, bs 4 38.4 -> 30.1, bs 1 33.9 -> 17.2), inside the graph they are
    # stream work between kernels.  DCD_STEP_GRAPH=1 / 0 forces it on / off.
    graph_env = os.environ.get("DCD_STEP_GRAPH")
    data_parallel = world > 1 or force_ddp
    use_graph = graph_env == "1" or (graph_env is None and (per_rank <= 4 or data_parallel))
    if data_parallel and use_graph:
        model = prepare_data_parallel(model, cfg)            # bare module: the graphed step reduces the gradients itself
    else:
        model = wrap_distributed(model, cfg, local_rank)
    rank = int(os.environ.get("RANK", 0))
    images, targets = make_batch(per_rank, seed=100 + rank, n_objects=args.objects, input_size=(in_w, in_h), device=device)
    return cfg, model, optimizer, images, targets, per_rank, use_graph, data_parallel


def run_gpu(args):
    import torch
    import torch.distributed as dist
    from dcd_amd import _ext
    from dcd_amd.engine.trainer import train_step

    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 or "RANK" in os.environ or os.environ.get("DCD_FORCE_DDP", "0") == "1":   # launched by torch.distributed.run
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")               # (or a one-rank group for DCD_FORCE_DDP=1)
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=device)       # "nccl" is RCCL on ROCm
    # MIOpen: its exhaustive find costs minutes on a fresh box, so by default its heuristics (immediate mode) pick the
    # conv kernels.  DCD_MIOPEN_FIND=1 turns the search on (measured in round 1: no gain).
    torch.backends.cudnn.benchmark = os.environ.get("DCD_MIOPEN_FIND", "0") == "1"

    cfg, model, optimizer, images, targets, per_rank, use_graph, data_parallel = build_everything(args, device, world, local_rank)
    timer = DcnTimer(torch, _ext)
    clip = cfg.SOLVER.GRAD_NORM_CLIP

    # Whole step replayed from ONE HIP graph (engine.trainer.GraphedTrainStep): see build_everything for when.  At bs 8 on one
    # GPU without collectives the GPU is the limit either way (49.7 vs 48.7 ms, round 2), so the default single-GPU line stays
    # eager.  Data parallel: the graph holds the SyncBN + gradient all-reduces; if capturing fails on ANY rank all ranks agree
    # (one eager all-reduce) to fall back to the eager DDP step.
    force_ddp = os.environ.get("DCD_FORCE_DDP", "0") == "1"
    step_launch = "eager"
    if use_graph:
        from dcd_amd.engine.trainer import GraphedTrainStep, wrap_distributed
        graphed = GraphedTrainStep(model, optimizer, clip, distributed=data_parallel, recapture_every=args.recapture_every)
        # capture WITHOUT replaying, then one eager all-reduce of the outcome, then -- only if every rank captured -- replays.
        # (A rank that replayed before the vote would sit in the graph's SyncBN all-reduces while a failed peer issues the
        # 1-element flag all-reduce: mismatched collectives, a hang.  The capture itself issues none: its warm-up steps are
        # common to all ranks and captured collectives are recorded, not run.)
        ok = graphed.capture(images, targets)
        if not ok:
            sys.stderr.write("[bench] whole-step graph unavailable on rank %d (%r)\n" % (rank, graphed.capture_error))
        ok = graphed.agree(ok)
        if not ok:
            sys.stderr.write("[bench] rank %d: all ranks take the eager step\n" % rank)
        if ok:
            step_launch = "one HIP graph per step" + (" (SyncBN + gradient all-reduce inside)" if data_parallel else "")

            if args.recapture_every:
                def step():
                    graphed(images, targets)                 # __call__: re-captures every N replays (all ranks count the same calls)
            else:
                def step():
                    graphed.replay(images, targets)
        else:
            use_graph = False
            if data_parallel:
                model = wrap_distributed(model, cfg, local_rank)
    if not use_graph:
        def step():
            train_step(model, optimizer, images, targets, clip)
    trace = (lambda m: (sys.stderr.write("[bench] %s\n" % m), sys.stderr.flush())) if os.environ.get("DCD_BENCH_TRACE") else (lambda m: None)
    for i in range(args.warmup):
        step()
        if i == 0:
            torch.cuda.synchronize()
            trace("first warm-up step done (graph captured and replayed once)" if use_graph else "first warm-up step done")
    trace("warm-up done")

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    timer.enabled = not use_graph
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    trace("timed region done")
    timer.enabled = False
    dcn_source = "event pairs around every DCN call inside the timed steps"
    if use_graph:
        # kernels inside a graph replay cannot be bracketed by events: the DCN time comes from eager steps of the same model
        # and batch, run right after the timed region (same process, same clocks)
        # (one untimed eager step first: allocator blocks of the eager step's sizes, DCN hand-over reports).  Under data
        # parallelism the eager step is host-bound (136 SyncBN collectives issued from Python), so the launches of one DCN call are
        # not queued back to back and the gaps between them fall between the call's two events: the figure is an upper bound there
        # (16-18 ms where the single-process eager step measures 13.5-14).
        train_step(model, optimizer, images, targets, clip)
        torch.cuda.synchronize()
        timer.enabled = True
        for _ in range(args.dcn_steps):
            train_step(model, optimizer, images, targets, clip)
        torch.cuda.synchronize()
        timer.enabled = False
        dcn_source = ("event pairs around every DCN call in %d eager steps run right after the timed (graph-replayed) region%s"
                      % (args.dcn_steps, "; host-bound under data parallelism: includes launch gaps, an upper bound" if data_parallel else ""))
    offsets = None
    if rank == 0 and not (isinstance(model, torch.nn.parallel.DistributedDataParallel) or data_parallel):
        # one eager FORWARD of the same model and batch, outside the timed region (no collectives involved: single-process runs only)
        with OffsetStats(torch, _ext) as st, torch.no_grad():
            model(images, targets)
        offsets = st.summary()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    global_batch = per_rank * world
    size_hw = "x".join(reversed(getattr(args, "input", "1280x384").split("x")))      # "384x1280" (H x W, as BASELINE.json writes it)
    # MODEL.FP16: a precision scope around backbone and predictor -- every DCN / 3x3 contraction as ONE product of bf16-rounded
    # operands (DCD_PREC_BF16), fp32 accumulate and storage
    prec = "bf16" if args.amp else args.precision
    # the fused DCN op has 196 FLOP per algorithmic byte: above the fp32 matrix ridge (157.3 TF / 8 TB/s = 19.7), below the bf16
    # one (2 500 / 8 = 312) -- in mixed precision HBM is the roofline that bounds it
    hbm_bound = prec == "bf16"
    dcn_count = args.dcn_steps if use_graph else args.steps
    dcn_ms = timer.total_ms() / max(dcn_count, 1)                       # per step, this rank's share of the batch
    by, fl = dcn_algorithmic(per_rank)
    traffic, mfma_busy, pmc_source, pmc_commit = load_pmc(per_rank, prec)
    out = None
    if rank == 0:
        out = {
            "metric": "images/sec DGDE train step (bs=%d, %s)" % (global_batch, size_hw), "value": global_batch * args.steps / elapsed,
            "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": {"f32": "f32", "bf16x3": "bf16x3", "bf16": "bf16"}[prec], "data": "synthetic",
            "config": {"workload": "DGDE train bs=%d %s on %dxMI355X, synthetic KITTI %s + random kpts_ann "
                                   "(DGDE.yaml, DLA-34+DCNv2, %d objects/image)" % (
                                       global_batch, "bf16 matrix operands, fp32 accumulate / storage (MODEL.FP16)" if args.amp
                                       else "fp32", world, size_hw, args.objects),
                       "global_batch": global_batch, "per_gpu_batch": per_rank, "input": size_hw,
                       "parallelism": "dp%d" % world, "dcn_precision": prec, "step_launch": step_launch,
                       "sync_bn": bool(data_parallel)},
            # The fused op has 196 FLOP per algorithmic byte (ridge of the part: 157.3 TF / 8 TB/s = 19.7), so the matrix pipe
            # is the bound that applies; the HBM view north_star also asks for is kept beside it.
            "roofline": {"bound": "hbm" if hbm_bound else "mfma", "kernel": "DCNv2 fwd+bwd, 16 layers, batch %d per GPU" % per_rank,
                         "achieved": ((by / 1e9 if hbm_bound else fl / 1e12) / (dcn_ms / 1e3)) if dcn_ms > 0 else None,
                         "peak": HBM_PEAK_GBS if hbm_bound else MFMA_PEAK_TFLOPS[prec], "unit": "GB/s" if hbm_bound else "TFLOP/s",
                         "frac": (((by / 1e9) / (dcn_ms / 1e3)) / HBM_PEAK_GBS if hbm_bound
                                  else (fl / 1e12 / (dcn_ms / 1e3)) / MFMA_PEAK_TFLOPS[prec]) if dcn_ms > 0 else None,
                         "mfma_frac": (fl / 1e12 / (dcn_ms / 1e3)) / MFMA_PEAK_TFLOPS[prec] if dcn_ms > 0 else None,
                         "traffic": traffic, "mfma_busy_pmc": mfma_busy, "pmc_source": pmc_source, "pmc_commit": pmc_commit,
                         # the ATTAINABLE bound beside the spec bound, from the same counter passes: per DCN kernel
                         # floor = matrix-pipe busy time + (vector instructions x 2.44 cycles at full occupancy) -- the f32-input MFMA
                         # shares the vector lanes, so the two ADD -- never below HBM bytes / 8 TB/s; composite_frac = sum of the floors
                         # over the sum of the kernels' times under the counters (tools/pmc_kernels.py)
                         "composite_frac": PMC_EXTRA.get("composite_frac"), "floor_ms": PMC_EXTRA.get("floor_ms"),
                         "kernel_ms_under_counters": PMC_EXTRA.get("kernel_ms"),
                         # what the counters say bounds these kernels (the algorithmic bound is "mfma": 196 FLOP/B): on gfx950 the
                         # f32-input MFMA executes on the vector ALUs' FP32 lanes -- one MFMA and one VALU instruction never
                         # overlap (tools/micro/mfma_valu_overlap.hip, profiles/r03_mfma_valu_overlap.txt) -- so the measured bound
                         # is the SUM of matrix and vector issue cycles, not the matrix pipe alone
                         "bound_measured": ("valu / lds issue (bf16 MFMA overlaps; sampling arithmetic is per (sample, channel))" if hbm_bound
                                            else "valu+mfma issue (f32 MFMA shares the VALU lanes)"), "flops": fl, "algorithmic_bytes": by, "ms_per_step": dcn_ms,
                         "calls_per_step": len(timer.pairs) // max(dcn_count, 1), "source": dcn_source,
                         "offsets": offsets,
                         "layers": timer.per_layer(max(dcn_count, 1), per_rank)},
            "roofline_hbm": {"bound": "hbm", "achieved": by / 1e9 / (dcn_ms / 1e3) if dcn_ms > 0 else None, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": (by / 1e9 / (dcn_ms / 1e3)) / HBM_PEAK_GBS if dcn_ms > 0 else None,
                             "algorithmic_bytes": by},
        }
    if out is not None and world == 1 and not args.amp and not args.no_op_line:
        out.update(op_level_lines(torch, _ext, per_rank, args.precision))
    # The same job with the DCN contractions in split-bf16 (DCD_PREC_BF16X3: hi*hi + hi*lo + lo*hi on the bf16 matrix cores, fp32
    # accumulate, ~2^-16 relative per product against north_star's 1e-3 bound) as an EXTRA object: `value` above stays the exact
    # fp32 run.  One GPU, eager step only (a captured graph has the fp32 kernels baked in).
    if out is not None and world == 1 and not use_graph and not args.amp and args.precision == "f32" and not args.no_split_line:
        # ... and in mixed precision (DCD_PREC_BF16, what MODEL.FP16 / --amp selects: ONE product of bf16-rounded operands, fp32
        # accumulate and storage): BASELINE config 3's arithmetic on this box, next to the fp32 number it is compared with
        for key, pname in (("split_bf16x3", "bf16x3"), ("mixed_bf16", "bf16")):
            _ext.set_precision(pname)
            s_steps = args.steps
            for _ in range(max(args.warmup, 2)):
                step()
            torch.cuda.synchronize()
            timer2 = DcnTimer(torch, _ext)
            timer2.enabled = True
            t0 = time.perf_counter()
            for _ in range(s_steps):
                step()
            torch.cuda.synchronize()
            s_el = time.perf_counter() - t0
            timer2.enabled = False
            _ext.set_precision("f32")
            s_dcn = timer2.total_ms() / s_steps
            out[key] = {"value": global_batch * s_steps / s_el, "unit": "images/s", "ms_per_step": 1e3 * s_el / s_steps,
                        "ratio_to_fp32": (global_batch * s_steps / s_el) / out["value"],
                        "steps": s_steps, "dcn_precision": pname, "dcn_ms_per_step": s_dcn,
                        "dcn_tflops": fl / 1e12 / (s_dcn / 1e3) if s_dcn > 0 else None,
                        "dcn_frac_of_fp32_mfma_peak": (fl / 1e12 / (s_dcn / 1e3)) / MFMA_PEAK_TFLOPS["f32"] if s_dcn > 0 else None,
                        # against its OWN peak: three (one) bf16 products per fp32 one on the 2.5 PF matrix cores
                        "dcn_frac_of_own_mfma_peak": (fl / 1e12 / (s_dcn / 1e3)) / MFMA_PEAK_TFLOPS[pname] if s_dcn > 0 else None,
                        "dcn_hbm_frac": (by / 1e9 / (s_dcn / 1e3)) / HBM_PEAK_GBS if s_dcn > 0 else None,
                        "note": "same model, data and step as `value`; the DCN and 3x3-convolution weight contractions change precision"}
    # The same train step at LARGE sampling offsets.  The headline's offsets are sub-pixel by construction
    # (conv_offset_mask ~ N(0, 0.01^2): `roofline.offsets`), the regime the one-pass DCN backward is built for; the reference's own
    # op test samples at 2 * randn px (DGDE/model/backbone/DCNv2/DCN/testcuda.py:73).  A second model, same architecture, batch and
    # seed with conv_offset_mask ~ N(0, std^2), std chosen so that mean |offset| is ~1.6 px (= E|2 randn|): what the step costs when
    # the offsets have grown.  One GPU, eager step, fp32.
    if (out is not None and world == 1 and not use_graph and not args.amp and args.precision == "f32" and not args.no_split_line
            and args.offset_std_2px > 0):
        import argparse as _ap
        a2 = _ap.Namespace(**vars(args))
        a2.offset_std = args.offset_std_2px
        _, model2, opt2, images2, targets2 = build_everything(a2, device, world, local_rank)[:5]
        for _ in range(max(args.warmup, 3)):
            train_step(model2, opt2, images2, targets2, clip)
        torch.cuda.synchronize()
        timer3 = DcnTimer(torch, _ext)
        timer3.enabled = True
        s_steps = args.steps
        t0 = time.perf_counter()
        for _ in range(s_steps):
            train_step(model2, opt2, images2, targets2, clip)
        torch.cuda.synchronize()
        s_el = time.perf_counter() - t0
        timer3.enabled = False
        with OffsetStats(torch, _ext) as st2, torch.no_grad():
            model2(images2, targets2)
        s_dcn = timer3.total_ms() / s_steps
        out["step_2px"] = {"value": global_batch * s_steps / s_el, "unit": "images/s", "ms_per_step": 1e3 * s_el / s_steps,
                           "ratio_to_headline": (global_batch * s_steps / s_el) / out["value"], "steps": s_steps,
                           "offset_weight_std": args.offset_std_2px, "offsets": st2.summary(), "dcn_ms_per_step": s_dcn,
                           "dcn_frac_of_fp32_mfma_peak": (fl / 1e12 / (s_dcn / 1e3)) / MFMA_PEAK_TFLOPS["f32"] if s_dcn > 0 else None,
                           "layers": timer3.per_layer(s_steps, per_rank),
                           "note": "same train step, model, batch and seed as `value` with conv_offset_mask ~ N(0, std^2) instead of "
                                   "N(0, 0.01^2): sampling offsets of the size the reference's op test uses"}
        del model2, opt2
    # N > 1, north_star's split as the headline: the weak-scaling number of the same job (8 images per rank, eager DDP step with
    # bucketed all-reduce overlapped with the backward) as an extra key -- what BASELINE.json configs[2] is at N = 4
    test_weak = force_ddp and os.environ.get("DCD_TEST_WEAK") == "1"        # one-GPU rehearsal of this branch (twice the batch)
    if (world > 1 or test_weak) and args.scaling == "strong" and not args.no_weak:
        from dcd_amd.data.synthetic import make_batch
        from dcd_amd.engine.trainer import wrap_distributed
        w_batch = args.batch * (2 if test_weak else 1)
        w_images, w_targets = make_batch(w_batch, seed=200 + rank, n_objects=args.objects, device=device)
        if use_graph:                                          # same graphed data-parallel step, captured for this batch size
            def w_step():
                graphed(w_images, w_targets)
        else:
            if not isinstance(model, torch.nn.parallel.DistributedDataParallel):
                model = wrap_distributed(model, cfg, local_rank)

            def w_step():
                train_step(model, optimizer, w_images, w_targets, clip)
        w_steps = max(2, min(args.steps, 5))
        for _ in range(2):
            w_step()
        fence()
        t0 = time.perf_counter()
        for _ in range(w_steps):
            w_step()
        fence()
        w_el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
        dist.all_reduce(w_el, op=dist.ReduceOp.MAX)
        if out is not None:
            out["weak_scaling"] = {"value": w_batch * world * w_steps / float(w_el.item()), "unit": "images/s", "scaling": "weak",
                                   "global_batch": w_batch * world, "per_gpu_batch": w_batch, "steps": w_steps,
                                   "ms_per_step": 1e3 * float(w_el.item()) / w_steps, "step_launch": step_launch}
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    return out


PMC_EXTRA = {}          # load_pmc: the counter file's attainable-bound figures (tools/pmc_kernels.py: composite_frac, floor_ms)


def load_pmc(per_rank, prec="f32"):
    """Counter evidence for the DCN kernels, from separate `rocprofv3 --pmc` passes over THIS command (tools/pmc_kernels.py ->
    profiles/dcn_pmc_r05.json, falling back to the earlier rounds' files): HBM bytes per step (FETCH_SIZE / WRITE_SIZE, corrected as
    MI355X_MICROARCH.md prescribes) and the time-weighted matrix-pipe busy fraction, plus the commit the passes were made from.
    (None, None, "absent", None) when the passes have not been made for this batch / precision (the committed passes are the
    fp32 step's; the mixed-precision kernels' counters are profiles/amp_pmc_r05.json, together with the Winograd kernels')."""
    if prec == "bf16" and per_rank == 8:
        # the mixed-precision step's passes cover the DCN, Winograd and GEMM kernels together: take the DCN kernels' rows
        try:
            with open(os.path.join(ROOT, "profiles", "amp_pmc_r05.json")) as f:
                d = json.load(f)
            steps = d["summary"]["steps_profiled"]
            rows = [v for k, v in d["kernels"].items() if k.startswith("dcn_") or k.startswith("sgemm_bf16")]
            by = sum(v["hbm_bytes_per_dispatch"] * v["dispatches"] for v in rows) / steps
            wt = sum(v["avg_us_profiled"] * v["dispatches"] for v in rows)
            busy = sum(v["mfma_busy"] * v["avg_us_profiled"] * v["dispatches"] for v in rows) / wt
            return int(by), float(busy), "profiles/amp_pmc_r05.json (dcn_* and sgemm_bf16* rows)", d.get("commit", "unknown")
        except (OSError, ValueError, KeyError, ZeroDivisionError):
            return None, None, "absent for this precision", None
    if prec != "f32":
        return None, None, "absent for this precision", None
    for name in ("dcn_pmc_r06.json", "dcn_pmc_r05.json", "dcn_pmc_r04.json", "dcn_pmc_r03.json", "dcn_pmc_r02.json"):
        path = os.path.join(ROOT, "profiles", name)
        try:
            with open(path) as f:
                d = json.load(f)
            if "--batch" in d.get("command", "") or per_rank != 8:      # the committed passes are the default bs-8 single-GPU step
                return None, None, "absent for this batch", None
            s_ = d["summary"]
            PMC_EXTRA["composite_frac"] = s_.get("composite_frac_time_weighted")
            PMC_EXTRA["floor_ms"] = s_.get("floor_ms_per_step")
            PMC_EXTRA["kernel_ms"] = s_.get("kernel_ms_per_step")
            return int(s_["hbm_bytes_per_step"]), float(s_["mfma_busy_time_weighted"]), "profiles/" + name, d.get("commit", "unknown")
        except (OSError, ValueError, KeyError):
            continue
    return None, None, "absent", None


def cpu_baseline_child(args):
    """Runs in a child process: the same train step on the host cores with the oracle patched in (kind "port")."""
    import torch
    from dcd_amd import ops
    from dcd_amd.config import get_cfg
    from dcd_amd.data.synthetic import make_batch
    from dcd_amd.engine.trainer import build_optimizer, init_like_trained, train_step
    from dcd_amd.model.backbone.DCNv2 import dcn_v2
    from dcd_amd.model.detector import KeypointDetector
    from oracle import dcn_oracle, torch_ops
    for name in ("pairs_kpts_depth", "compute_z", "focal_loss", "giou_loss", "nms_hm", "select_topk",
                 "select_point_of_interest", "iou_3d"):
        setattr(ops, name, getattr(torch_ops, name))
    dcn_v2._backend = dcn_oracle
    cfg = get_cfg(opts=["MODEL.PRETRAIN", False, "MODEL.DEVICE", "cpu", "MODEL.USE_SYNC_BN", False])
    torch.manual_seed(0)
    model = KeypointDetector(cfg).train()
    init_like_trained(model)
    opt = build_optimizer(model, cfg)
    images, targets = make_batch(args.cpu_batch, seed=100, n_objects=args.objects)
    # warm-up step(s) first, then timed steps.  A CPU step takes tens of seconds, so the number of timed
    # steps is bounded by a time budget (>= 1, <= 3): the sample says how many were run.
    t0 = time.perf_counter()
    train_step(model, opt, images, targets, cfg.SOLVER.GRAD_NORM_CLIP)
    warm = time.perf_counter() - t0
    n_timed = max(1, min(3, int(args.cpu_budget / max(warm, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(n_timed):
        train_step(model, opt, images, targets, cfg.SOLVER.GRAD_NORM_CLIP)
    dt = (time.perf_counter() - t0) / n_timed
    print(json.dumps({"value": args.cpu_batch / dt, "unit": "images/s", "cores": torch.get_num_threads(),
                      "kind": "port", "host_cpus": os.cpu_count(), "batch": args.cpu_batch, "warmup_steps": 1, "timed_steps": n_timed,
                      "s_per_step": dt, "warmup_s": warm,
                      "sample": "DGDE train step (fwd+loss+bwd+AdamW) at bs=%d, 384x1280, %d objects/image, oracle DCNv2 (C, OpenMP) + "
                                "PyTorch CPU convs: 1 warm-up step (%.1f s) + %d timed step(s) of %.1f s.  Batch %d only: BASELINE.md "
                                "section 3 also names bs 8 (the metric's batch), ~4x this step's time, outside the bench's budget "
                                "for the baseline" % (args.cpu_batch, args.objects, warm, n_timed, dt, args.cpu_batch)}))


def cpu_serial_dcn_child(args):
    """Variant A of SURVEY section 8(d): the reference's CPU DCN is single-threaded scalar loops (cpu/dcn_v2_im2col_cpu.cpp) --
    our C restatement of them on ONE core (OMP_NUM_THREADS=1 set by the parent).  Bounded sample: one forward+backward of each
    of the 7 distinct DCN geometries at one image, scaled by how often the geometry occurs (16 layers)."""
    import torch
    from oracle import dcn_oracle
    a = (3, 3, 1, 1, 1, 1, 1, 1, 1)
    g = torch.Generator().manual_seed(0)
    total = 0.0
    for cin, cout, h, w, n in DCN_LAYERS:
        x = torch.randn(1, cin, h, w, generator=g)
        off = torch.randn(1, 18, h, w, generator=g) * 0.5
        m = torch.rand(1, 9, h, w, generator=g)
        wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
        b = torch.zeros(cout)
        gy = torch.randn(1, cout, h, w, generator=g)
        t0 = time.perf_counter()
        dcn_oracle.dcn_v2_forward(x, wt, b, off, m, *a)
        dcn_oracle.dcn_v2_backward(x, wt, b, off, m, gy, *a)
        total += n * (time.perf_counter() - t0)
    print(json.dumps({"dcn_s_per_image": total, "cores": 1, "kind": "port",
                      "sample": "DCNv2 forward+backward, serial C loops on one core: one call per distinct geometry (7) at 1 image, "
                                "scaled by the layer counts (16 layers)"}))


def cpu_baseline(args):
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", "--cpu-batch", str(args.cpu_batch),
           "--cpu-budget", str(args.cpu_budget), "--objects", str(args.objects), "--workload", args.workload]
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    try:
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=args.cpu_timeout, env=env)
        line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
        out = json.loads(line)
    except Exception as e:  # a missing baseline must not kill the GPU number
        return {"value": None, "unit": "objects/s" if args.workload == "gmw" else "images/s", "cores": None, "kind": "port",
                "sample": "failed: %r" % (e,)}
    if args.workload == "dgde":          # variant A (serial, like the reference's own CPU loops) beside variant B (all cores)
        try:
            res = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-serial-dcn-child"], capture_output=True, text=True,
                                 timeout=args.cpu_timeout, env=dict(env, OMP_NUM_THREADS="1"))
            out["serial_dcn"] = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
        except Exception as e:
            out["serial_dcn"] = {"dcn_s_per_image": None, "sample": "failed: %r" % (e,)}
    return out


# ------------------------------------------------------------------------------------------------
# Secondary workload (BASELINE config 5): the GMW train step.  `--workload gmw`; never the default.
# ------------------------------------------------------------------------------------------------
GMW_EDGES = 2628


def _gmw_inputs(B, seed, device=None):
    """Seeded GMW batch (K-normalised 2-D keypoints of projected object-frame 3-D keypoints + 2e-3 noise, yaw, location)."""
    import numpy as np
    import torch
    rng = np.random.default_rng(seed)
    dims = np.array([3.9, 1.5, 1.6], dtype=np.float32)
    k3 = ((rng.random((B, 73, 3)) - 0.5) * dims).astype(np.float32)
    rot = (rng.random((B, 1)) * 2 * np.pi - np.pi).astype(np.float32)
    loc = np.stack([(rng.random(B) - 0.5) * 10, np.full(B, 1.65), 8 + rng.random(B) * 40], 1).astype(np.float32)
    c, s = np.cos(rot[:, 0]), np.sin(rot[:, 0])
    zc = -k3[:, :, 0] * s[:, None] + k3[:, :, 2] * c[:, None] + loc[:, None, 2]
    k2 = np.stack([(k3[:, :, 0] * c[:, None] + k3[:, :, 2] * s[:, None] + loc[:, None, 0]) / zc, (k3[:, :, 1] + loc[:, None, 1]) / zc], -1)
    k2 = (k2 + rng.standard_normal(k2.shape) * 2e-3).astype(np.float32)
    out = tuple(torch.from_numpy(a) for a in (k2, k3, rot, loc))
    return tuple(t.to(device) for t in out) if device is not None else out


def run_gmw(args):
    import torch
    import torch.distributed as dist
    from dcd_amd.gmw import GMW, gmw_train_step
    from dcd_amd.gmw.optimal_transport import RegularisedTransportFn
    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1 or "RANK" in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=device)
    per_rank = args.batch if args.scaling == "weak" else max(args.batch // world, 1)
    torch.manual_seed(0)
    model = GMW().to(device).train()
    if world > 1:
        model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank])
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.999))           # GMW/main.py:255-259
    batch = _gmw_inputs(per_rank, 100 + rank, device)
    # the transport layer's backward, timed with events around its own entry point (the step's dominant piece)
    pairs, orig = [], RegularisedTransportFn.gradient
    timing = {"on": False}

    def timed_gradient(*a, **k):
        if not timing["on"]:
            return orig(*a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig(*a, **k)
        e1.record()
        pairs.append((e0, e1))
        return out
    RegularisedTransportFn.gradient = staticmethod(timed_gradient)
    for _ in range(args.warmup):
        gmw_train_step(model, opt, *batch, 0.1, 1.0)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    fence()
    timing["on"] = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        gmw_train_step(model, opt, *batch, 0.1, 1.0)
    fence()
    elapsed = time.perf_counter() - t0
    timing["on"] = False
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    tb_ms = sum(a.elapsed_time(b) for a, b in pairs) / max(args.steps, 1)
    if os.environ.get("DCD_BENCH_TRACE"):
        print("[gmw] elapsed %.2f ms per step, argv %s" % (1e3 * elapsed / args.steps, sys.argv[1:]), file=sys.stderr)
    n = GMW_EDGES
    flops = per_rank * (2.0 * n * (n - 1) * n + n ** 3 / 3.0 + n ** 3 / 3.0)       # S = D2 - B^T D1 B, potrf, inverse of the factor
    out = None
    if rank == 0:
        out = {"metric": "objects/sec GMW train step (73 keypoints, 2628 edges)", "value": per_rank * world * args.steps / elapsed,
               "unit": "objects/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "config": {"workload": "GMW train step, %d objects (%d per GPU) x 2628 edges, cls 0.1 + reg 1.0 (GMW/main.py:313-315)" % (
                   per_rank * world, per_rank), "global_batch": per_rank * world, "per_gpu_batch": per_rank, "parallelism": "dp%d" % world},
               "roofline": {"bound": "mfma", "kernel": "transport-layer backward (Schur complement on the MFMA GEMM, blocked Cholesky + substitutions: csrc/spd.hip)",
                            "achieved": flops / 1e12 / (tb_ms / 1e3) if tb_ms > 0 else None, "peak": MFMA_PEAK_TFLOPS["f32"],
                            "unit": "TFLOP/s", "frac": (flops / 1e12 / (tb_ms / 1e3)) / MFMA_PEAK_TFLOPS["f32"] if tb_ms > 0 else None,
                            "traffic": None, "flops": flops, "ms_per_step": tb_ms}}
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    return out


def gmw_cpu_baseline_child(args):
    """The same GMW step on the host cores (our mirror with the oracle's compute_z; kind "port"), 2 objects."""
    import torch
    from dcd_amd.gmw import GMW, gmw_train_step
    from oracle import torch_ops
    torch.manual_seed(0)
    model = GMW().train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.999))
    batch = _gmw_inputs(2, 100)
    gmw_train_step(model, opt, *batch, 0.1, 1.0, compute_z=torch_ops.compute_z)
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        gmw_train_step(model, opt, *batch, 0.1, 1.0, compute_z=torch_ops.compute_z)
    dt = (time.perf_counter() - t0) / n
    print(json.dumps({"value": 2 / dt, "unit": "objects/s", "cores": torch.get_num_threads(), "kind": "port", "host_cpus": os.cpu_count(),
                      "sample": "%d GMW train steps of 2 objects x 2628 edges on the host (PyTorch CPU ops, LAPACK Cholesky); %.1f s each" % (n, dt)}))


# ------------------------------------------------------------------------------------------------
# Secondary workload, BASELINE config 4: the `--generate_for_GMW` pass at bs 16.  `--workload gen`.
# One step = one batch of 16 images through BOTH halves of the pass:
#   (1) the training forward + loss-path decode under no_grad, BatchNorm frozen (DGDE/engine/trainer.py:62-67,89-129:
#       `is_gen`), `Loss_Computation.generate_data` appending the K-normalised key points (detector_loss.py:148-173);
#   (2) the evaluation pass at batch 1 per image (DGDE/data/build.py:141-143): eval forward, PostProcessor.forward
#       (fused NMS + top-50, POI gather, uncertainty-weighted depth, edge solver over all 2628 pairs,
#       detector_infer.py:86-243) and the GMW records of DGDE/engine/inference.py:59-84.
# value = images / s of the whole pass.  The solver calls are timed with event pairs; the reference issues 3 x 2628 slice
# copies per decode (anno_encoder.py:313-390), ours one launch.
# ------------------------------------------------------------------------------------------------
def _gen_build(args, device):
    import torch
    from dcd_amd.config import get_cfg
    from dcd_amd.data.synthetic import make_batch
    from dcd_amd.engine.trainer import init_like_trained
    from dcd_amd.model.detector import KeypointDetector
    # random-init weights score every cell ~0.01, below the 0.2 threshold (runs/DGDE.yaml:78): a zero threshold keeps the
    # top DETECTIONS_PER_IMG = 50 cells of every image, i.e. the eval decode at its largest
    cfg = get_cfg(opts=["MODEL.PRETRAIN", False, "MODEL.DEVICE", str(device), "MODEL.USE_SYNC_BN", False, "TEST.GENERATE_GMW", True,
                        "TEST.DETECTIONS_THRESHOLD", 0.0])
    torch.manual_seed(0)
    model = KeypointDetector(cfg)
    init_like_trained(model, std=0.01, seed=0)
    model = model.to(device)
    # inference heads at the top-K cells only (the decode reads nothing else, detector_infer.py:101-110); DCD_GEN_DENSE_HEADS=1 keeps
    # the reference's dense 415-channel map
    model.heads.predictor.sparse_eval_heads = device.type == "cuda" and os.environ.get("DCD_GEN_DENSE_HEADS", "0") != "1"
    images, targets = make_batch(args.batch, seed=100, n_objects=args.objects, device=device if device.type == "cuda" else None)
    return cfg, model, images, targets


def _gen_pass(model, images, targets, torch, batched_eval=True):
    """Both halves of the pass over one batch; returns (objects written by the train half, detections of the eval half)."""
    from dcd_amd.engine.gen_data import infer_records
    lc = model.heads.loss_evaluator
    for k in lc.gen_data:
        lc.gen_data[k] = []
    model.train()
    for m in model.modules():                                   # freeze_bn (DGDE/engine/trainer.py:62-67)
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.eval()
    with torch.no_grad():
        model(images, targets)
        n_train = sum(len(x) for x in lc.gen_data["pred_rot"])
        model.eval()
        n_det = 0
        if batched_eval:
            # The reference infers image by image (TEST.IMS_PER_BATCH = 1, DGDE/engine/inference.py:59-84) and its decode assumes one
            # image (detector_infer.py:173,186,221).  Backbone and predictor in eval mode are per-sample independent, so they run
            # ONCE on the batch here; the decode keeps its one-image form on slices of their outputs: same rows per image.
            feats = model.backbone(images)
            preds = model.heads.predictor(feats, targets)
            # ... and since round 4 the decode too (PostProcessor.forward_batch: one NMS / top-K, one gather, one solver call, every
            # row with its own image's padding and intrinsics -> the rows of the image-by-image loop), records from ONE packed copy
            from dcd_amd.engine.gen_data import infer_records_batch
            rows, _, vis, image_of = model.heads.post_processor.forward_batch(preds, targets, test=model.test, features=feats)
            n_det += sum(len(r) for r in infer_records_batch(rows, vis, image_of, images.shape[0]))
        else:
            for i in range(images.shape[0]):
                result, _, vis = model(images[i:i + 1], targets[i:i + 1])
                n_det += len(infer_records(result, vis))
    return n_train, n_det


def run_gen(args):
    import torch
    from dcd_amd import _ext, ops
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    torch.backends.cudnn.benchmark = os.environ.get("DCD_MIOPEN_FIND", "0") == "1"
    cfg, model, images, targets = _gen_build(args, device)
    timer = DcnTimer(torch, _ext)
    pairs, orig = [], ops.pairs_kpts_depth
    timing = {"on": False}

    def timed_solver(*a, **k):
        if not timing["on"]:
            return orig(*a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = orig(*a, **k)
        e1.record()
        pairs.append((e0, e1, int(a[0].shape[0])))
        return out
    ops.pairs_kpts_depth = timed_solver
    for _ in range(args.warmup):
        _gen_pass(model, images, targets, torch)
    torch.cuda.synchronize()
    timer.enabled = timing["on"] = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        n_train, n_det = _gen_pass(model, images, targets, torch)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timer.enabled = timing["on"] = False
    ops.pairs_kpts_depth = orig
    B = images.shape[0]
    dcn_ms = timer.total_ms() / max(args.steps, 1)
    fl = sum(n * 2 * co * 9 * ci * h * w for ci, co, h, w, n in DCN_LAYERS) * 2 * B        # forward only; every image passes twice
    by = sum(n * 4 * ((ci + 27 + co) * h * w + 9 * ci * co + co) for ci, co, h, w, n in DCN_LAYERS) * 2 * B
    sol_us = [a.elapsed_time(b) * 1e3 for a, b, _ in pairs]
    sol_obj = sum(n for _, _, n in pairs)
    return {"metric": "images/sec DGDE --generate_for_GMW pass (bs=%d, 384x1280)" % B, "value": B * args.steps / elapsed, "unit": "images/s",
            "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "DGDE --generate_for_GMW pass bs=%d on 1xMI355X: no-grad train-mode forward (BN frozen) + loss-path "
                                   "decode + generate_data, then eval forward + PostProcessor + GMW records at batch 1 per image; synthetic "
                                   "KITTI 384x1280, %d objects/image" % (B, args.objects),
                       "global_batch": B, "per_gpu_batch": B, "input": "384x1280", "parallelism": "dp1",
                       "objects_written_per_step": n_train, "detections_per_step": n_det,
                       "score_threshold": 0.0, "note": "random-init weights: zero score threshold -> 50 detections per image (the cap)"},
            "roofline": {"bound": "mfma", "kernel": "DCNv2 forward, 16 layers, %d + %d x 1 images per step" % (B, B),
                         "achieved": fl / 1e12 / (dcn_ms / 1e3) if dcn_ms > 0 else None, "peak": MFMA_PEAK_TFLOPS["f32"], "unit": "TFLOP/s",
                         "frac": (fl / 1e12 / (dcn_ms / 1e3)) / MFMA_PEAK_TFLOPS["f32"] if dcn_ms > 0 else None, "traffic": None,
                         "flops": fl, "algorithmic_bytes": by, "ms_per_step": dcn_ms, "calls_per_step": len(timer.pairs) // max(args.steps, 1),
                         "source": "event pairs around every DCN call inside the timed passes"},
            "solver": {"kernel": "edge-constraint depth solve (csrc/heads.hip edge_depth_fwd): one launch per decode",
                       "calls_per_step": len(pairs) / max(args.steps, 1), "us_per_call": sum(sol_us) / max(len(sol_us), 1),
                       "objects_per_step": sol_obj / max(args.steps, 1),
                       "objects_per_s_in_kernel": sol_obj / (sum(sol_us) * 1e-6) if sol_us else None,
                       "launches_per_call": 1, "reference_launches_per_call": 3 * 2628,
                       "reference_source": "DGDE/model/anno_encoder.py:313-390: three get_up calls x 2628 slice copies per decode"}}


def gen_cpu_baseline_child(args):
    """The same pass on the host cores with the oracle patched in (kind "port"), on a bounded sample: 2 images."""
    import torch
    from dcd_amd import ops
    from dcd_amd.model.backbone.DCNv2 import dcn_v2
    from oracle import dcn_oracle, torch_ops
    for name in ("pairs_kpts_depth", "compute_z", "focal_loss", "giou_loss", "nms_hm", "select_topk",
                 "select_point_of_interest", "iou_3d"):
        setattr(ops, name, getattr(torch_ops, name))
    dcn_v2._backend = dcn_oracle
    args.batch = 2
    cfg, model, images, targets = _gen_build(args, torch.device("cpu"))
    _gen_pass(model, images[:1], targets[:1], torch)            # warm-up: one image
    t0 = time.perf_counter()
    _gen_pass(model, images, targets, torch)
    dt = time.perf_counter() - t0
    print(json.dumps({"value": 2 / dt, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port", "host_cpus": os.cpu_count(),
                      "sample": "the same pass over 2 images on the host (oracle DCNv2 + PyTorch CPU ops), after a 1-image warm-up; %.1f s" % dt}))


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves, the way the reference
    spawns its own (DGDE/engine/launch.py:50-55).  This process has not touched the GPU (nothing is imported before this
    point), it starts torch.distributed.run as a CHILD and exits with its code; rank 0's JSON line passes through."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    return subprocess.call(cmd, env=env)


def run_dry(args):
    """Launch / fence / max-over-ranks / one-JSON-line plumbing with no GPU in it (gloo on the host).  The step is a
    gradient-sized all-reduce of zeros; tests/test_bench_launch.py runs this at world size 2.  Never a measurement."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
    per_rank = args.batch if args.scaling == "weak" else max(args.batch // world, 1)
    grad = torch.zeros(1 << 16)

    def step():
        if world > 1:
            dist.all_reduce(grad)
    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.destroy_process_group()
    if rank != 0:
        return None
    return {"metric": "images/sec DGDE train step (bs=%d, 384x1280)" % (per_rank * world), "value": None, "unit": "images/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / max(args.steps, 1),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "none",
            "dry": True, "config": {"workload": "DRY RUN of the launch path, no GPU work", "global_batch": per_rank * world,
                                    "per_gpu_batch": per_rank, "parallelism": "dp%d" % world}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=("dgde", "gmw", "gen"), default="dgde",
                    help="dgde: the headline metric (default); gmw: SURVEY 8(f) rank 1 (BASELINE config 5); gen: the --generate_for_GMW "
                         "pass (BASELINE config 4; use --batch 16)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="global batch (strong, default) or per-GPU batch (weak)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="strong (default): --batch is the GLOBAL batch, split over the ranks like DGDE/data/build.py:63-67; "
                         "weak: --batch images on every rank")
    ap.add_argument("--no-weak", action="store_true", help="N > 1: skip the extra weak-scaling measurement")
    ap.add_argument("--objects", type=int, default=6)
    ap.add_argument("--no-split-line", action="store_true", help="skip the extra split-bf16 measurement of the default run")
    ap.add_argument("--no-op-line", action="store_true", help="skip the op-level DCN lines (roofline_op_2px / _0p5px)")
    ap.add_argument("--precision", choices=("f32", "bf16x3", "bf16"), default="f32",
                    help="matrix path of the DCN / 3x3 weight contractions everywhere (bf16x3: split bf16, ~2^-16 per product; "
                         "bf16: one product of bf16-rounded operands; fp32 in / fp32 out either way)")
    ap.add_argument("--amp", action="store_true", help="MODEL.FP16: backbone and predictor inside a bf16 precision scope (DCN and 3x3 "
                                                        "contractions on the bf16 matrix cores, fp32 accumulate and storage) "
                                                        "(BASELINE config 3: --gpus 4 --batch 32 --amp)")
    ap.add_argument("--dcn-steps", type=int, default=5, help="eager steps used to time the DCN calls when the timed steps are graph replays")
    ap.add_argument("--input", default="1280x384", help="WxH of the synthetic frames (the metric's size is 1280x384; other sizes are for "
                                                        "checks such as tools/check_graph_memsets.sh -- their line is not the metric)")
    ap.add_argument("--offset-std", type=float, default=0.01, help="std of the conv_offset_mask initialisation (SURVEY 8d: 0.01)")
    ap.add_argument("--offset-std-2px", type=float, default=0.08,
                    help="std of the extra `step_2px` model (0.08: mean |offset| 1.6 px = E|2 randn|, profiles/r06_offset_std.txt); 0 skips that leg")
    ap.add_argument("--recapture-every", type=int, default=0,
                    help="graphed step: capture the whole-step graph again every N replays (a captured graph freezes the DCN launch "
                         "policy of its capture; long runs refresh it)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=2, help="batch of the CPU baseline step (BASELINE.md section 3: bs 2)")
    ap.add_argument("--cpu-budget", type=float, default=40.0, help="seconds of TIMED CPU steps after the warm-up step (1..3 steps)")
    ap.add_argument("--cpu-timeout", type=int, default=420)
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-serial-dcn-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--dry", action="store_true", help="exercise the multi-rank launch path on the host (gloo), no GPU work")
    args = ap.parse_args()
    if args.cpu_serial_dcn_child:
        cpu_serial_dcn_child(args)
        return
    if args.cpu_baseline_child:
        {"gmw": gmw_cpu_baseline_child, "gen": gen_cpu_baseline_child}.get(args.workload, cpu_baseline_child)(args)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    # RCCL prints its version banner on stdout when the first communicator is created; the contract is ONE JSON line on
    # stdout, so everything else written to fd 1 during the run goes to stderr.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    out = run_dry(args) if args.dry else run_gmw(args) if args.workload == "gmw" else run_gen(args) if args.workload == "gen" else run_gpu(args)
    sys.stdout.flush()
    if out is not None:
        if args.gpus == 1 and not args.no_cpu_baseline and not args.dry:
            out["cpu_baseline"] = cpu_baseline(args)
        os.write(real_stdout, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
