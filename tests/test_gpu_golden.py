"""GPU parity against the reference: the same fixture checks as tests/test_host_golden.py, but with the real HIP
kernels underneath (no patching) -- i.e. reference Python on CPU  vs  dcd_amd on the MI355X, identical inputs.
Tolerances: 1e-3 relative (north_star) for model-level quantities; tighter where written."""
import numpy as np
import os

import pytest
import torch

import golden_inputs as gi
import test_host_golden as H

pytestmark = pytest.mark.gpu


# test_whole_model_in_mixed_bf16_precision: every large gradient tensor of the mixed-precision run against the fp32 run of the same
# model (cosine, norm ratio); calibrated on MI355X, see the test
MIXED_GRAD_COS, MIXED_GRAD_RATIO = 0.45, 1.35      # measured over three runs: worst cosine 0.565-0.570, worst ratio 1.19-1.21


def _one_image(preds, i):
    """Image i of a predictor output: tensors sliced, the top-K tuple of a `sparse_eval_heads` predictor slice by slice."""
    def cut(v):
        if torch.is_tensor(v):
            return v[i:i + 1]
        if isinstance(v, tuple):
            return tuple(cut(u) for u in v)
        return v
    return {k: cut(v) for k, v in preds.items()}


def test_native_library_is_loaded(cuda):
    from dcd_amd import _lib
    L = _lib.lib()
    assert L.dcd_version().startswith(b"dcd_hip")
    with open("/proc/self/maps") as f:
        assert "libdcd_hip.so" in f.read(), "the HIP extension must be the code that runs"


def test_anno_encoder_matches_reference(cuda):
    H.check_anno_encoder(cuda)


def test_edge_depth_matches_reference(cuda):
    """HIP solver vs the reference's decode_pairs_kpts_depth / GMW compute_z outputs (fixtures), incl. gradients."""
    from dcd_amd import ops
    import test_oracle_golden as OG
    g = H.load("edge_depth")
    kps, k3, rot, P, mask = gi.edge_inputs()
    a = torch.from_numpy(kps).to(cuda).requires_grad_()
    b = torch.from_numpy(k3).to(cuda).requires_grad_()
    r, Pt = torch.from_numpy(rot).to(cuda), torch.from_numpy(P).to(cuda)
    d, _ = ops.pairs_kpts_depth(a, b, r, Pt, training=False)
    np.testing.assert_allclose(d.detach().cpu().numpy(), g["eval_depth"], rtol=2e-4, atol=2e-4)
    (d * torch.from_numpy(gi.edge_grad_weights(d.shape)).to(cuda)).sum().backward()
    for got, key in ((a.grad, "eval_grad_kps"), (b.grad, "eval_grad_kps3d")):
        ref = g[key]
        assert np.abs(got.cpu().numpy() - ref).max() <= 1e-3 * np.abs(ref).max()
    a.grad = None
    b.grad = None
    depth, idx, pmask = ops._PairsDepth.apply(a, b, r, Pt, torch.from_numpy(mask).to(cuda), 1500, 2.0, 80.0, 0, 1)
    idx_np = idx.cpu().numpy().astype(np.int64)
    OG.assert_equal_up_to_ties(depth.detach().cpu().numpy(), g["train_depth"], kps, P, idx_np, 2e-4)
    OG.assert_equal_up_to_ties(pmask.cpu().numpy(), g["train_mask"], kps, P, idx_np, 0.0)
    depth.sum().backward()
    for got, key in ((a.grad, "train_grad_kps"), (b.grad, "train_grad_kps3d")):
        ref = g[key]
        assert np.abs(got.cpu().numpy() - ref).max() <= 1e-3 * np.abs(ref).max()
    kn = gi.normalise_kps(kps, P)
    z, zidx = ops.compute_z(torch.from_numpy(kn).to(cuda), torch.from_numpy(k3).to(cuda), r)
    np.testing.assert_allclose(z.cpu().numpy(), g["gmw_z"], rtol=2e-4, atol=2e-4)
    assert np.array_equal(np.sort(zidx.cpu().numpy(), 1), np.sort(g["gmw_idx"], 1)), "top-1500 SET must be identical"


def test_losses_match_reference(cuda):
    from dcd_amd import ops
    g = H.load("losses")
    pred, tgt = gi.focal_inputs()
    p = torch.from_numpy(pred).to(cuda).requires_grad_()
    loss, npos = ops.focal_loss(p, torch.from_numpy(tgt).to(cuda), 2, 4)
    assert npos.item() == float(g["focal_npos"])
    assert abs(loss.item() - float(g["focal_loss"])) <= 1e-4 * abs(float(g["focal_loss"]))
    loss.backward()
    assert np.abs(p.grad.cpu().numpy() - g["focal_grad"]).max() <= 1e-4 * np.abs(g["focal_grad"]).max()
    bp, bt = gi.giou_inputs()
    q = torch.from_numpy(bp).to(cuda).requires_grad_()
    losses, ious = ops.giou_loss(q, torch.from_numpy(bt).to(cuda))
    np.testing.assert_allclose(losses.detach().cpu().numpy(), g["giou_losses"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(ious.cpu().numpy(), g["giou_ious"], rtol=1e-5, atol=1e-6)
    losses.sum().backward()
    assert np.abs(q.grad.cpu().numpy() - g["giou_grad"]).max() <= 1e-4 * np.abs(g["giou_grad"]).max()


def test_decode_matches_reference_exactly(cuda):
    """nms_hm / select_topk / POI gather vs the reference's own functions: indices and values bit-exact."""
    from dcd_amd import ops
    g = H.load("decode")
    h = torch.from_numpy(gi.heat_inputs()).to(cuda)
    nms = ops.nms_hm(h)
    assert np.array_equal(nms.cpu().numpy(), g["nms"])
    for fused in (False, True):
        out = ops.select_topk(h if fused else nms, 50, fuse_nms=fused)
        for t, key in zip(out, ("scores", "inds", "clses", "ys", "xs")):
            assert np.array_equal(t.cpu().numpy(), g[key]), (key, fused)
    feat, pts = gi.poi_inputs()
    poi = ops.select_point_of_interest(feat.shape[0], torch.from_numpy(pts).to(cuda), torch.from_numpy(feat).to(cuda))
    assert np.array_equal(poi.cpu().numpy(), g["poi"])


def test_loss_computation_matches_reference(cuda):
    H.check_loss_computation(cuda, 1e-4)


def test_loss_computation_through_the_row_kernel_matches_reference(cuda, monkeypatch):
    """the reference's Loss_Computation fixture DIRECTLY through csrc/loss_rows.hip (the regression map gathered
    at the object centres, as the predictor hands it over in training): 13 losses, log dict and both gradients at 1e-4 -- not
    only transitively (kernel == op-by-op rows == reference)."""
    monkeypatch.setenv("DCD_LOSS_ROWS", "1")
    from dcd_amd import ops
    calls = []
    real = ops.loss_rows
    monkeypatch.setattr(ops, "loss_rows", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    H.check_loss_computation(cuda, 1e-4, through_row_kernel=True)
    assert calls, "the loss did not go through ops.loss_rows"


def test_gen_data_for_gmw_matches_reference(cuda):
    H.check_gen_data(cuda, 2e-3)      # the normalised keypoints pass through the solver's mean depth: 2e-3 of the value range


def test_post_processor_matches_reference(cuda):
    """The eval decode on the GPU path (fused NMS + top-K kernel, POI gather, HIP edge-depth solve, device sin/cos/atan2) on
    pinned predictor maps vs the reference's output: exact row order, every column within 1e-4 of its range."""
    H.check_post_processor(cuda, 1e-4)


@pytest.mark.parametrize("trunk_form", ["shift", "bmm"])
def test_whole_model_matches_reference(cuda, monkeypatch, trunk_form):
    """KeypointDetector on the GPU (HIP DCNv2 / Winograd / BN / heads kernels + MIOpen for the remaining convs) against the
    EXACT result: the reference model run in float64 (tests/golden/model_96x320_f64.npz; our shell reproduces that run bit for
    bit on the host, test_float64_model_equals_reference_float64).  Measured on MI355X (tools/model_dist_f64.py): activations
    1.2-1.6e-5 from exact (the reference's own fp32 CPU run: 1.0-1.5e-5), losses <= 6.3e-5 (reference fp32: <= 4.9e-6),
    per-parameter gradient norms <= 2.6e-3 (reference fp32: 1.25e-2 on the same yardstick).  Bars: 1e-4 on activations,
    3e-4 on the 13 losses (north_star: 1e-3), 6e-3 on gradient norms -- i.e. the GPU run must stay as close to the exact
    result as the reference's fp32 run is, not merely close to that run."""
    torch.backends.cudnn.benchmark = False
    monkeypatch.setenv("DCD_TRUNK_GRAM", trunk_form)       # both forms of the trunk statistics (trunk_moments.py; "auto" picks by size)
    H.check_model(cuda, 1e-4, 6e-3, truth="model_96x320_f64", loss_tol=3e-4)


def test_whole_model_in_mixed_bf16_precision(cuda):
    """BASELINE config 3's arithmetic against the EXACT result: the same model with every contraction of our kernels as one product
    of bf16-rounded operands (`_ext.precision_scope("bf16")`, what MODEL.FP16 turns on: DCNv2 forward / backward incl. the dense
    path's GEMMs, the Winograd 3x3 convolutions and their weight gradients on every map size) against the reference's float64 run
    (tests/golden/model_96x320_f64.npz).  bf16 carries 8 mantissa bits through ~90 layers of a randomly initialised net (and the
    Winograd transforms round TRANSFORMED values): measured on MI355X (tools/model_dist_f64.py with DCD_PRECISION=bf16) activations
    4.4-7.5e-2 of their range (max norm), the 13 losses <= 2.9e-2 each, per-parameter gradient norms 7.8e-2 in the median and up
    to 0.45 for the regression heads' last layers (a handful of objects feed them); bars 0.15 / 0.08 / 0.9 (north_star's 1e-3 is
    the fp32 bound: the fp32 test above holds 1e-4).  The decode's top-50 of nearly-equal scores moves with the maps' few per cent:
    at least a third of the rows must match."""
    from dcd_amd import _ext
    torch.backends.cudnn.benchmark = False
    with _ext.precision_scope("bf16"):
        H.check_model(cuda, 0.15, 0.9, truth="model_96x320_f64", loss_tol=0.08, decode_tol=0.1, decode_min_match=0.3, sparse_tol=0.1)
    assert _ext.get_precision() == "f32"
    # DIRECTION and SIZE of every large gradient tensor against the exact-fp32 run of the same model (the norm bar above
    # would pass a sign or scale error in one of the bf16 backward kernels -- direct weight gradient, 1x1 weight gradient,
    # one-product DCN sweep).  Measured (tools/probes/amp_grad_by_loss.py): at this random initialisation the backbone gradient
    # of EVERY loss taken alone has cosine 0.62-0.70 and norm ratio 0.95-1.10 against fp32 (BatchNorm's backward cancels most of
    # each layer's incoming gradient, bf16's 2^-9 is relative to the uncancelled size); `keypoint_depth_loss` (depth = f h / the
    # predicted keypoint height, a division by a random net's near-zero output) is the one chaotic term: ratio 0.48, cosine 0.39.
    # So: gradient of the other twelve losses, per tensor.  A flipped sign gives about -0.65, a factor 2 a ratio of 2.
    from dcd_amd.model.detector import KeypointDetector
    model = KeypointDetector(H.small_cfg(str(cuda))).to(cuda).train()
    images, targets = gi.model_inputs()
    images, targets = images.to(cuda), [t.to(cuda) for t in targets]
    keys = [k for k in H.LOSS_KEYS if k != "keypoint_depth_loss"]

    def grads(prec):
        gi.name_hashed_init(model)
        model.zero_grad()
        with _ext.precision_scope(prec):
            ld, _ = model(images, targets)
        sum(ld[k] for k in keys).backward()
        return {n: p.grad.detach().double().flatten() for n, p in model.named_parameters() if p.grad is not None and p.numel() >= 4096}
    g32, g16 = grads("f32"), grads("bf16")
    top = max(float(v.norm()) for v in g32.values())
    worst_cos, worst_ratio, n_big = 1.0, 1.0, 0
    for n, a in g32.items():
        if float(a.norm()) < 1e-6 * top:
            continue
        b = g16[n]
        n_big += 1
        cos, ratio = float(torch.dot(a, b) / (a.norm() * b.norm())), float(b.norm() / a.norm())
        worst_cos, worst_ratio = min(worst_cos, cos), max(worst_ratio, ratio, 1.0 / ratio)
        if os.environ.get("DCD_TEST_PRINT_GRAD_DEV") and (cos < 0.6 or max(ratio, 1 / ratio) > 1.15):
            print("%-60s cos %.3f ratio %.3f" % (n, cos, ratio))
    if os.environ.get("DCD_TEST_PRINT_GRAD_DEV"):
        print("mixed vs fp32 gradients over %d large tensors: worst cosine %.3f, worst norm ratio %.3f" % (n_big, worst_cos, worst_ratio))
    assert n_big >= 60
    assert worst_cos >= MIXED_GRAD_COS and worst_ratio <= MIXED_GRAD_RATIO, (worst_cos, worst_ratio)


def test_whole_model_in_split_bf16_precision(cuda, monkeypatch):
    """The same model with every split-bf16 kernel on (`_ext.set_precision("bf16x3")`: DCNv2 forward / backward incl. the dense
    path's GEMMs, and the Winograd 3x3 convolutions on every map size) against the float64 reference run, at north_star's bound:
    1e-3 on activations and losses (measured ~2e-5 / 1e-4), 2.2e-2 on the per-parameter gradient norms (measured 1.94-1.98e-2)."""
    from dcd_amd import _ext, ops
    torch.backends.cudnn.benchmark = False
    monkeypatch.setattr(ops, "_CONV_SPLIT_MIN_MAP", 0)
    # the stride-2 layers on our kernels too: MIOpen's implicit GEMMs for them split K with atomics, the one part of this run that
    # did not repeat (a deterministic path instead of a wider bar)
    monkeypatch.setattr(ops, "_S2D_MODE", "1")
    _ext.set_precision("bf16x3")
    try:
        # (decode: which of the nearly-equal scores of a random net make the top 50 moves with the 1e-5 of the split products AND
        # with the atomics of the edge-fusion scatter from run to run: 39-45 of 50 rows seen; the decode itself is pinned exactly
        # on fixed maps by check_post_processor)
        # gradient norms: the worst parameter is always the FIRST BatchNorm's weight (backbone.base.base_layer.1), where the
        # rounding of every later layer has accumulated: 1.94e-2 .. 1.98e-2 over the runs of this commit (the exact-fp32 run of
        # the same check: 2e-3; rounds 4-5, with the stock solver's atomics in the run: 1.90e-2 .. 2.03e-2 around a 2e-2 bar)
        H.check_model(cuda, 1e-3, 2.2e-2, truth="model_96x320_f64", loss_tol=1e-3, decode_min_match=0.7)
    finally:
        _ext.set_precision("f32")


def test_whole_model_close_to_reference_fp32_run(cuda):
    """Same model against the reference's fp32 CPU run: two fp32 runs each ~1.5e-5 from exact."""
    torch.backends.cudnn.benchmark = False
    H.check_model(cuda, 1e-4, 2e-2, loss_tol=3e-4)


def test_iou3d_kernel(cuda):
    from dcd_amd import ops
    from oracle import torch_ops
    from dcd_amd.model.anno_encoder import Anno_Encoder
    enc = Anno_Encoder(H.small_cfg("cpu"))
    d = gi.anno_inputs()
    rng = np.random.RandomState(5)
    a = enc.encode_box3d(torch.from_numpy(d["rotys"]), torch.from_numpy(d["dims"]), torch.from_numpy(d["locs"]))
    b = enc.encode_box3d(torch.from_numpy(d["rotys"] + rng.normal(0, 0.2, 12).astype(np.float32)),
                         torch.from_numpy(d["dims"] * 1.1),
                         torch.from_numpy(d["locs"] + rng.normal(0, 0.4, (12, 3)).astype(np.float32)))
    ref = torch_ops.iou_3d(a, b).numpy()
    got = ops.iou_3d(a.to(cuda), b.to(cuda)).cpu().numpy()
    assert ref.max() > 0.05
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-5)


def test_graphed_loss_equals_eager_loss_on_changing_targets(cuda):
    """The HIP-graph replay of the loss section (forward + backward) must equal the eager computation for every batch fed
    to it, not only the one it was captured on: two different synthetic batches, graph vs eager, losses and the gradients
    that flow back into the predictor outputs."""
    from dcd_amd.config import get_cfg
    from dcd_amd.data.synthetic import make_batch
    from dcd_amd.model.head.detector_loss import Loss_Computation
    cfg = get_cfg(opts=["MODEL.PRETRAIN", False, "MODEL.DEVICE", str(cuda), "INPUT.WIDTH_TRAIN", 320, "INPUT.HEIGHT_TRAIN", 96])
    lc = Loss_Computation(cfg)
    B, M, C, H, W = 2, cfg.DATASETS.MAX_OBJECTS, 415, 24, 80
    g = torch.Generator().manual_seed(0)
    for seed, n_obj in ((3, 3), (4, 5), (3, 3)):
        _, targets = make_batch(B, seed=seed, n_objects=n_obj, input_size=(320, 96), device=cuda)
        cls = torch.sigmoid(torch.randn(B, 1, H, W, generator=g)).clamp(1e-4, 1 - 1e-4).to(cuda)
        pois = (torch.randn(B, M, C, generator=g) * 0.3).to(cuda)
        res = []
        for use_graph in (False, True):
            lc.use_graph = use_graph
            c, p = cls.clone().requires_grad_(), pois.clone().requires_grad_()
            loss_dict, log = lc({'cls': c, 'reg': None, 'reg_pois': p}, targets)
            if use_graph:                      # the graph also forms the sum; train_step back-propagates that one tensor
                total = loss_dict.total
                assert abs(float(total) - sum(float(v) for v in loss_dict.values())) <= 1e-5 * abs(float(total))
                total.backward()
            else:                              # eager: back-propagate the dict's values one by one (the other way in)
                assert abs(float(loss_dict.total) - sum(float(v) for v in loss_dict.values())) <= 1e-5 * abs(float(loss_dict.total))
                sum(loss_dict.values()).backward()
            res.append(({k: float(v) for k, v in loss_dict.items()}, c.grad.clone(), p.grad.clone(), dict(log)))
        (l0, gc0, gp0, log0), (l1, gc1, gp1, log1) = res
        for k in l0:
            assert abs(l0[k] - l1[k]) <= 1e-5 * max(abs(l0[k]), 1e-3), (seed, k, l0[k], l1[k])
        assert (gc0 - gc1).abs().max().item() <= 1e-6 * max(gc0.abs().max().item(), 1e-6)
        assert (gp0 - gp1).abs().max().item() <= 1e-5 * max(gp0.abs().max().item(), 1e-6)
        assert set(log0) == set(log1)
    assert len(lc._graphs) == 1          # one capture served all three batches


def test_solver_fused_adamw_and_clip_match_reference(cuda):
    """SURVEY 8(f)-3 on the device path: the GPU step uses FUSED AdamW over two pooled groups and our own clip_grad_norm
    (engine/trainer.py), not the reference's per-parameter for-loop AdamW + torch.nn.utils.clip_grad_norm_
    (DGDE/solver/__init__.py:10-62, DGDE/engine/trainer.py:139-148).  Same fixture as the CPU test (three AdamW steps and a
    fourth from a loaded reference checkpoint, produced by the reference's solver package), now with everything on the GPU."""
    from dcd_amd.config import get_cfg
    from dcd_amd.engine.trainer import (build_optimizer, build_scheduler, clip_grad_norm, guard_nonfinite_step,
                                        load_checkpoint_state, step_schedulers)
    g = H.load("solver")
    cfg = get_cfg(opts=["MODEL.PRETRAIN", False, "MODEL.DEVICE", str(cuda), "SOLVER.LR_WARMUP", True, "SOLVER.WARMUP_STEPS", 200,
                        "SOLVER.MAX_ITERATION", 3000, "SOLVER.STEPS", (2000, 2600)])
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3)).to(cuda)
    opt = build_optimizer(net, cfg)
    assert opt.defaults["fused"] is True
    sched, warm = build_scheduler(opt, cfg)
    x = torch.linspace(-1, 1, 24).reshape(4, 6).to(cuda)
    for it in range(3):
        opt.zero_grad()
        net(x).square().sum().backward()
        opt.step()
        step_schedulers(sched, warm, it, cfg)
    got = np.concatenate([p.detach().cpu().numpy().ravel() for p in net.parameters()])
    np.testing.assert_allclose(got, g["params_after_3_steps"], rtol=5e-6, atol=2e-7)
    # resume from the reference's checkpoint (per-parameter groups) and take its 4th step with the fused optimizer
    net2 = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3)).to(cuda)
    opt2 = build_optimizer(net2, cfg)
    sched2, _ = build_scheduler(opt2, cfg)
    load_checkpoint_state(H._reference_checkpoint_from_fixture(g, net2), net2, opt2, sched2)
    opt2.zero_grad()
    net2(x).square().sum().backward()
    opt2.step()
    got = np.concatenate([p.detach().cpu().numpy().ravel() for p in net2.parameters()])
    np.testing.assert_allclose(got, g["params_after_4_steps"], rtol=5e-6, atol=2e-7)

    # clip_grad_norm == torch.nn.utils.clip_grad_norm_ (the reference's call), clipping and non-clipping cases
    gen = torch.Generator().manual_seed(3)
    shapes = [(64, 16, 3, 3), (64,), (27, 64, 3, 3), (5, 7), (1,)]
    for scale, max_norm in ((10.0, 15.0), (0.01, 15.0)):
        ps = [torch.nn.Parameter(torch.zeros(s)) for s in shapes]
        for p in ps:
            p.grad = torch.randn(p.shape, generator=gen) * scale
        qs = [torch.nn.Parameter(torch.zeros(s, device=cuda)) for s in shapes]
        for p, q in zip(ps, qs):
            q.grad = p.grad.to(cuda)
        ref_total = torch.nn.utils.clip_grad_norm_(ps, max_norm)
        total = clip_grad_norm(qs, max_norm)
        assert abs(float(total) - float(ref_total)) <= 2e-6 * float(ref_total)
        for p, q in zip(ps, qs):
            assert (q.grad.cpu() - p.grad).abs().max().item() <= 2e-6 * p.grad.abs().max().item()

    # a non-finite gradient must not reach the weights (fused kernel's found_inf flag, no host sync)
    before = [p.detach().clone() for p in net.parameters()]
    opt.zero_grad()
    (net(x).square().sum() * float("nan")).backward()
    guard_nonfinite_step(opt, clip_grad_norm(list(net.parameters()), 15.0))
    opt.step()
    assert all(torch.equal(a, b) for a, b in zip(before, net.parameters()))
    opt.zero_grad()
    net(x).square().sum().backward()
    guard_nonfinite_step(opt, clip_grad_norm(list(net.parameters()), 15.0))
    opt.step()
    assert not all(torch.equal(a, b) for a, b in zip(before, net.parameters()))


def test_model_fp16_flag_trains_in_bf16_autocast(cuda):
    """BASELINE config 3 (bs 8 per rank, bf16): MODEL.FP16 puts the backbone and the predictor into a mixed-precision region like
    the reference's autocast (DGDE/model/detector.py:34-36, head/detector_head.py:20-22) -- here the bf16 precision scope of our
    own kernels (operands rounded to bf16, one product on the bf16 matrix cores, fp32 accumulate and storage).  One train step at
    8 images per rank must (a) really run every DCN call and the 3x3 convolutions' three kernels in that precision, forward AND
    backward (the backward runs outside the scope), (b) give the fp32 run's losses to bf16 accuracy (15 % each, 3 % in total),
    (c) produce finite gradients for every parameter the fp32 run has gradients for, and step."""
    from dcd_amd import _ext, ops
    from dcd_amd.config import get_cfg
    from dcd_amd.data.synthetic import make_batch
    from dcd_amd.engine.trainer import build_optimizer, init_like_trained, train_step
    from dcd_amd.model.detector import KeypointDetector
    images, targets = make_batch(8, seed=5, n_objects=3, input_size=(320, 96), device=cuda)
    seen, seen_bwd, seen_conv = [], [], []
    fwd, bwd, wrw, call = _ext.dcn_v2_forward, _ext.dcn_v2_backward, ops._conv3x3_wrw_call, ops._conv3x3_call

    def spy(*a, **k):
        seen.append(k.get("precision"))
        return fwd(*a, **k)

    def spy_bwd(*a, **k):
        seen_bwd.append(k.get("precision"))
        return bwd(*a, **k)

    def spy_wrw(x, gy, wshape, prec=0):
        seen_conv.append(("wrw", prec))
        return wrw(x, gy, wshape, prec)

    def spy_call(inp, weight, out_channels, backward_data, *a, **k):
        t = k.get("transformed")
        seen_conv.append(("dgrad" if backward_data else "fwd", t.prec if isinstance(t, ops.SplitWeights) else k.get("prec", 0) or 0))
        return call(inp, weight, out_channels, backward_data, *a, **k)
    results = {}
    for fp16 in (False, True):
        cfg = get_cfg(opts=["MODEL.PRETRAIN", False, "MODEL.DEVICE", str(cuda), "MODEL.USE_SYNC_BN", False, "MODEL.FP16", fp16,
                            "INPUT.WIDTH_TRAIN", 320, "INPUT.HEIGHT_TRAIN", 96])
        torch.manual_seed(0)
        model = KeypointDetector(cfg).to(cuda).train()
        init_like_trained(model)
        opt = build_optimizer(model, cfg)
        before = [p.detach().clone() for p in model.parameters()]
        del seen[:], seen_bwd[:], seen_conv[:]
        _ext.dcn_v2_forward, _ext.dcn_v2_backward, ops._conv3x3_wrw_call, ops._conv3x3_call = spy, spy_bwd, spy_wrw, spy_call
        try:
            loss_dict, _ = train_step(model, opt, images, targets, cfg.SOLVER.GRAD_NORM_CLIP)
        finally:
            _ext.dcn_v2_forward, _ext.dcn_v2_backward, ops._conv3x3_wrw_call, ops._conv3x3_call = fwd, bwd, wrw, call
        want = "bf16" if fp16 else None                   # exact fp32: no keyword at all (the reference's own signature)
        assert len(seen) == 16 and all(p == want for p in seen), seen
        assert len(seen_bwd) == 16 and all(p == want for p in seen_bwd), seen_bwd
        assert _ext.get_precision() == "f32"
        kinds = {k for k, _ in seen_conv}
        assert kinds == {"fwd", "dgrad", "wrw"} and len(seen_conv) >= 3 * 15, (kinds, len(seen_conv))     # (96 x 320: two DLA levels qualify)
        assert all(p == (ops.PREC_BF16 if fp16 else ops.PREC_F32) for _, p in seen_conv), sorted(set(seen_conv))
        results[fp16] = ({k: float(v) for k, v in loss_dict.items()},
                         [None if p.grad is None else bool(torch.isfinite(p.grad).all()) for p in model.parameters()],
                         sum(int(not torch.equal(a, b)) for a, b in zip(before, model.parameters())))
    (l32, g32, moved32), (l16, g16, moved16) = results[False], results[True]
    assert g32 == g16 and all(v is not False for v in g16), "non-finite or missing gradients under MODEL.FP16"
    assert moved16 == moved32 > 0
    # bf16 carries 8 mantissa bits through ~90 layers: the individual regression losses of a randomly initialised net move by
    # up to ~10 % (measured: dims_loss 9 %), their sum by much less
    for k in l32:
        assert abs(l16[k] - l32[k]) <= 0.15 * max(abs(l32[k]), 1e-2), (k, l16[k], l32[k])
    assert abs(sum(l16.values()) - sum(l32.values())) <= 0.03 * sum(l32.values())


@pytest.mark.parametrize("data_parallel", [False, True])
def test_graphed_train_step_equals_eager(cuda, data_parallel):
    """The whole step replayed from one HIP graph (engine.trainer.GraphedTrainStep) against the eager `train_step`.
    data_parallel: the graph also holds the SyncBN all-reduces and the flat gradient all-reduce (RCCL, a one-rank group here:
    the mechanics of capturing collectives; the eager reference runs the same SyncBN code, and with one rank its
    unreduced gradients are the reduced ones).
    AdamW's first updates are +-lr whatever the gradient's size, so round-off in a near-zero gradient flips whole updates and
    two runs of the SAME eager code drift apart after one step; the comparison therefore holds the weights fixed (lr = 0:
    AdamW's decay is lr-scaled too) over three steps on three DIFFERENT batches -- the graph is captured on the first and must
    follow the copied-in inputs, including other intrinsics -- and checks losses, every gradient, the BN buffers and the
    optimizer's moments; a fourth step with lr > 0 (tensor-lr path, filled after the capture) must then move the weights
    like the eager one wherever the gradient is not noise."""
    from dcd_amd.config import get_cfg
    from dcd_amd.data.synthetic import make_batch
    from dcd_amd.engine.trainer import GraphedTrainStep, build_optimizer, init_like_trained, train_step
    from dcd_amd.model.detector import KeypointDetector
    cfg = get_cfg(opts=["MODEL.PRETRAIN", False, "MODEL.DEVICE", str(cuda), "MODEL.USE_SYNC_BN", False,
                        "INPUT.WIDTH_TRAIN", 320, "INPUT.HEIGHT_TRAIN", 96])
    batches = [make_batch(2, seed=s, n_objects=n, input_size=(320, 96), device=cuda) for s, n in ((3, 3), (4, 5), (7, 2), (3, 3))]
    batches[1][1][0].get_field("calib").f_u *= 1.01            # other intrinsics in the second batch
    lrs = (0.0, 0.0, 0.0, 3e-4)
    runs = []
    if data_parallel:
        import socket
        import torch.distributed as dist
        from dcd_amd.engine.trainer import prepare_data_parallel
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = s_.getsockname()[1]
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=cuda)
        request_cleanup = dist.destroy_process_group
    else:
        request_cleanup = lambda: None
    for graphed in (False, True):
        torch.manual_seed(0)
        model = KeypointDetector(cfg).to(cuda).train()
        init_like_trained(model)
        opt = build_optimizer(model, cfg)
        if data_parallel:                                   # both runs: SyncBN kernels + frozen dead projections; only the graphed one reduces
            cfg_dp = get_cfg(opts=["MODEL.PRETRAIN", False, "MODEL.DEVICE", str(cuda), "MODEL.USE_SYNC_BN", True,
                                   "INPUT.WIDTH_TRAIN", 320, "INPUT.HEIGHT_TRAIN", 96])
            prepare_data_parallel(model, cfg_dp)
        step = GraphedTrainStep(model, opt, cfg.SOLVER.GRAD_NORM_CLIP, distributed=data_parallel) if graphed else None
        trace = []
        for (images, targets), lr in zip(batches, lrs):
            for g_ in opt.param_groups:
                g_["lr"].fill_(lr)
            loss_dict, _ = step(images, targets) if graphed else train_step(model, opt, images, targets, cfg.SOLVER.GRAD_NORM_CLIP)
            trace.append(({k: float(v.detach()) for k, v in loss_dict.items()},
                          [None if p.grad is None else p.grad.detach().clone() for p in model.parameters()],
                          [b.detach().clone().float() for b in model.buffers()],
                          [p.detach().clone() for p in model.parameters()],
                          [opt.state[p]["exp_avg"].clone() for p in model.parameters() if p in opt.state]))
        runs.append(trace)
    request_cleanup()
    eager, graph = runs
    # Forward quantities are bit-stable run to run; gradients are not (fp32 atomics in the backward of DCN / BN-at-positions /
    # scatter-adds, amplified by the cancellation inside grad_offset on the deep layers: two runs of the SAME eager step differ
    # by 1e-4 .. 3e-2 element-wise on a few tensors, tools/debug_step_graph2.py), so gradients are compared by norm, like the
    # whole-model fixture test.
    def norms(ts):
        return [None if t is None else float(t.double().norm()) for t in ts]
    # Two losses sit behind the edge solver's top-1500 selection (DGDE/model/anno_encoder.py:355-377: pairs with the largest |dv|):
    # a near-tie at the 1500th pair flips with the 1e-7 noise of the forward (fp32 atomics in BN-at-positions / scatter-adds) and
    # moves them by a few 1e-4 -- seen between two runs of the SAME eager code (tools/probes/dp_graph_check.py: corner_loss
    # 0.200333 in five runs, 0.200249 in the sixth; every other loss unchanged).  Those two get the wider bar.
    behind_topk = ("corner_loss", "extra_kpts_depth_loss")
    for i, ((l0, g0, b0, w0, m0), (l1, g1, b1, w1, m1)) in enumerate(zip(eager, graph)):
        for k in l0:
            assert abs(l0[k] - l1[k]) <= (3e-3 if k in behind_topk else 1e-4) * max(abs(l0[k]), 1e-2), (i, k, l0[k], l1[k])
        n0, n1 = norms(g0), norms(g1)
        floor = 1e-5 * max(v for v in n0 if v is not None)
        for a, b in zip(n0, n1):
            assert (a is None) == (b is None)
            if a is not None:
                assert abs(a - b) <= 2e-2 * max(a, floor), (i, a, b)
        for a, b in zip(b0, b1):
            assert (a - b).abs().max().item() <= 1e-5 * max(a.abs().max().item(), 1e-3), i
        for a, b in zip(norms(m0), norms(m1)):
            assert abs(a - b) <= 2e-2 * max(a, floor), (i, a, b)
        if i < 3:
            assert all(torch.equal(a, b) for a, b in zip(w0, w1)), "lr = 0 must leave the weights alone"
    # step 4 (lr 3e-4): the two updates must point the same way and have the same size
    ue = torch.cat([(a - w).flatten() for w, a in zip(eager[2][3], eager[3][3])])
    ug = torch.cat([(b - w).flatten() for w, b in zip(graph[2][3], graph[3][3])])
    assert float(ue.norm()) > 0 and abs(float(ue.norm()) - float(ug.norm())) <= 0.05 * float(ue.norm())
    assert float(torch.dot(ue, ug) / (ue.norm() * ug.norm())) >= 0.9
    assert len(step._graphs) == 1
    # the replays stepped the optimizer behind autograd's version counters: no 3x3 layer's prepared (Winograd-domain) weights may
    # pass for current in an eager forward that follows (ops.invalidate_conv_weights in GraphedTrainStep.replay)
    from dcd_amd import ops
    table = ops._PREPARED.get((cuda.index if cuda.index is not None else torch.cuda.current_device(), False))
    assert table is not None and table.entries and all(table.lookup(e[0]()) is None for e in table.entries.values() if e[0]() is not None)


def test_loss_graph_inside_the_train_loop_equals_eager(cuda):
    """Regression for the round-2 finding (profiles/r02_graph_memset_hazard.txt): inside the real train loop at BASELINE's batch
    (8 images x 40 slots x 1500 pairs) the graphed loss section returned 0 / 2x / 1e5 sums from the second replay on, because
    ATen's multi-block reductions clear their semaphore with a memset node.  Eight train steps with the loss graph on; after
    each, the same predictions are evaluated eagerly: all 13 terms must agree, and the loss must come down, not explode."""
    from dcd_amd.config import get_cfg
    from dcd_amd.data.synthetic import make_batch
    from dcd_amd.engine.trainer import build_optimizer, init_like_trained, train_step
    from dcd_amd.model.detector import KeypointDetector
    from dcd_amd.model.head import detector_loss as DL
    cfg = get_cfg(opts=["MODEL.PRETRAIN", False, "MODEL.DEVICE", str(cuda), "MODEL.USE_SYNC_BN", False,
                        "INPUT.WIDTH_TRAIN", 640, "INPUT.HEIGHT_TRAIN", 192])
    images, targets = make_batch(8, seed=100, n_objects=6, input_size=(640, 192), device=cuda)
    torch.manual_seed(0)
    model = KeypointDetector(cfg)
    init_like_trained(model, std=0.01, seed=0)
    model = model.to(cuda).train()
    opt = build_optimizer(model, cfg)
    lc = model.heads.loss_evaluator
    assert lc.use_graph
    seen = {}
    orig = DL.Loss_Computation.__call__

    def spy(self, predictions, tg):
        out = orig(self, predictions, tg)
        seen["pred"] = {k: (None if v is None else v.detach().clone()) for k, v in predictions.items()}
        return out
    DL.Loss_Computation.__call__ = spy
    try:
        totals = []
        for it in range(8):
            loss_dict, _ = train_step(model, opt, images, targets, cfg.SOLVER.GRAD_NORM_CLIP)
            graphed = {k: float(v.detach()) for k, v in loss_dict.items()}
            lc.use_graph = False
            with torch.no_grad():
                eager, _ = orig(lc, seen["pred"], targets)
            lc.use_graph = True
            for k in graphed:
                e = float(eager[k])
                assert abs(graphed[k] - e) <= 1e-4 * max(abs(e), 1e-2), (it, k, graphed[k], e)
            totals.append(sum(graphed.values()))
    finally:
        DL.Loss_Computation.__call__ = orig
    assert len(lc._graphs) == 1
    assert totals[-1] < totals[0] and max(totals) < 2 * totals[0], totals


@pytest.mark.parametrize("size", [(352, 96, 3), (672, 224, 2)])
def test_train_steps_at_sizes_off_the_alignment_rules(cuda, size):
    """SMOKE (no parity assertion: finite losses and parameters after three steps).
    Input sizes whose deeper maps break the kernels' alignment rules (W % 4 for the pooling / Winograd / tiled DCN kernels,
    32-pixel tiles, H % 2): every dispatch must fall back cleanly -- three train steps, finite losses and parameters.
    (tools/check_odd_sizes.py runs the same at 1248x384.)"""
    from dcd_amd.config import get_cfg
    from dcd_amd.data.synthetic import make_batch
    from dcd_amd.engine.trainer import build_optimizer, init_like_trained, train_step
    from dcd_amd.model.detector import KeypointDetector
    w, h, b = size
    cfg = get_cfg(opts=["MODEL.PRETRAIN", False, "MODEL.DEVICE", str(cuda), "INPUT.WIDTH_TRAIN", w, "INPUT.HEIGHT_TRAIN", h])
    torch.manual_seed(0)
    model = KeypointDetector(cfg).to(cuda).train()
    init_like_trained(model)
    opt = build_optimizer(model, cfg)
    images, targets = make_batch(b, seed=3, n_objects=4, input_size=(w, h), image_size=(w - 10, h - 5), device=cuda)
    for _ in range(3):
        ld, _ = train_step(model, opt, images, targets, cfg.SOLVER.GRAD_NORM_CLIP)
        total = getattr(ld, "total", None)
        v = float((total if total is not None else sum(ld.values())).detach())
        assert v == v and abs(v) < 1e6
    assert all(bool(torch.isfinite(p).all()) for p in model.parameters())


def test_generate_for_gmw_pass_full_size(cuda):
    """PROPERTIES (size-independent invariants; parity of this pass is pinned at 96x320 by test_gen_data_for_gmw_matches_reference).
    BASELINE config 4 at full size (bench.py --workload gen): both halves of the --generate_for_GMW pass over a 384x1280
    batch.  Size-independent properties: the train half writes one record per annotated object with 73 K-normalised key points,
    every image yields DETECTIONS_PER_IMG rows at a zero score threshold, record fields have the wire format's shapes
    (DGDE/engine/inference.py:59-84), depths lie inside the solver's clamp [2, 80] - P[2,3] and everything is finite."""
    import argparse
    import sys
    root = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    from dcd_amd.engine.gen_data import infer_records
    args = argparse.Namespace(batch=2, objects=6)
    cfg, model, images, targets = bench._gen_build(args, cuda)
    n_train, n_det = bench._gen_pass(model, images, targets, torch)
    lc = model.heads.loss_evaluator
    assert n_train == 2 * 6 and n_det == 2 * cfg.TEST.DETECTIONS_PER_IMG
    k2 = np.array(lc.gen_data["kpts_2d"][0], np.float32)
    k3 = np.array(lc.gen_data["kpts_3d"][0], np.float32)
    assert k2.shape == (12, 73, 2) and k3.shape == (12, 73, 3) and np.isfinite(k2).all() and np.isfinite(k3).all()
    assert len(lc.gen_data["pred_location"][0]) == 12 and len(lc.gen_data["img_idx"][0]) == 12
    model.eval()
    with torch.no_grad():
        result, _, vis = model(images[:1], targets[:1])
    recs = infer_records(result, vis)
    assert result.shape == (50, 14) and len(recs) == 50 and bool(torch.isfinite(result).all())
    r = recs[0]
    assert (len(r["kpts_2d"]), len(r["kpts_2d"][0]), len(r["kpts_3d"][0]), len(r["box"]), len(r["dim"]), len(r["pred_location"])) == (73, 2, 3, 4, 3, 3)
    z = result[:, 11]
    assert float(z.min()) > 0.0 and float(z.max()) <= 101.0                    # decoded depths are clamped ([0.1, 100] direct, [2, 80] pair depths)
    # the pass's eval half runs backbone + predictor once on the batch and decodes image by image on slices: same rows as the
    # reference's one-image-at-a-time inference (eval mode is per-sample independent)
    with torch.no_grad():
        feats = model.backbone(images)
        preds = model.heads.predictor(feats, targets)
        for i in range(images.shape[0]):
            one = _one_image(preds, i)
            r_b, _, _ = model.heads.post_processor(one, targets[i:i + 1], test=model.test, features=feats[i:i + 1])
            r_1, _, _ = model(images[i:i + 1], targets[i:i + 1])
            assert r_b.shape == r_1.shape
            assert torch.equal(r_b[:, 0], r_1[:, 0])                                             # classes, in the same order
            assert (r_b - r_1).abs().max().item() <= 1e-4 * max(r_1.abs().max().item(), 1.0)


def test_batched_post_processor_equals_the_image_by_image_decode(cuda):
    """VERDICT r3 item 5 -- BASELINE config 4's batch (16 images): `PostProcessor.forward_batch` (one NMS / top-K, one gather, one
    solver call, one packed copy for the records) against the reference's loop of one-image decodes (DGDE/engine/inference.py:59-84,
    detector_infer.py:86-243) on the same predictions: identical rows in (image, rank) order -- with intrinsics and image sizes that
    DIFFER between the images, so a row decoded with another image's calibration would show -- and identical GMW records."""
    import argparse
    import sys
    root = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    from dcd_amd.engine.gen_data import infer_records, infer_records_batch
    args = argparse.Namespace(batch=16, objects=6)
    cfg, model, images, targets = bench._gen_build(args, cuda)
    for i, t in enumerate(targets):                                # per-image intrinsics (KITTI sequences differ slightly), padding, size
        from dcd_amd.data.calibration import Calibration
        P = np.array(t.get_field("calib").P, dtype=np.float64).copy()
        s_ = 1.0 + 0.01 * i
        P[0, 0] *= s_; P[1, 1] *= s_; P[0, 2] += i; P[1, 2] -= 0.5 * i
        t.add_field("calib", Calibration(P))
        t.add_field("pad_size", t.get_field("pad_size") + (i % 3))
        t.size = (t.size[0] - 2 * (i % 4), t.size[1] - (i % 2))
    model.eval()
    pp = model.heads.post_processor
    assert model.heads.predictor.sparse_eval_heads                 # bench._gen_build: heads at the top-K cells only ...
    with torch.no_grad():
        feats = model.backbone(images)
        sparse = model.heads.predictor(feats, targets)
        assert sparse['reg'] is None and sparse['reg_pois'].shape == (16, cfg.TEST.DETECTIONS_PER_IMG, 415)
        rows_s, _, _, image_of_s = pp.forward_batch(sparse, targets, test=model.test, features=feats)
        model.heads.predictor.sparse_eval_heads = False            # ... against the reference's dense map: same cells, same rows
        preds = model.heads.predictor(feats, targets)
        assert preds['reg'].shape == (16, 415, 96, 320)
        rows, _, vis, image_of = pp.forward_batch(preds, targets, test=model.test, features=feats)
        assert torch.equal(image_of, image_of_s) and torch.equal(rows[:, 0], rows_s[:, 0]) and torch.equal(rows[:, 13], rows_s[:, 13])
        assert (rows - rows_s).abs().max().item() <= 1e-4 * max(rows.abs().max().item(), 1.0)
        recs_b = infer_records_batch(rows, vis, image_of, images.shape[0])
        assert rows.shape == (16 * cfg.TEST.DETECTIONS_PER_IMG, 14) and bool(torch.isfinite(rows).all())
        assert image_of.tolist() == sorted(image_of.tolist())
        for i in range(images.shape[0]):
            one = _one_image(preds, i)
            r_1, _, vis_1 = pp(one, targets[i:i + 1], test=model.test, features=feats[i:i + 1])
            r_b = rows[image_of == i]
            assert r_b.shape == r_1.shape
            assert torch.equal(r_b[:, 0], r_1[:, 0]) and torch.equal(r_b[:, 13], r_1[:, 13])     # classes and scores: same cells, same order
            assert (r_b - r_1).abs().max().item() <= 1e-5 * max(r_1.abs().max().item(), 1.0), i
            recs_1 = infer_records(r_1, vis_1)
            assert len(recs_1) == len(recs_b[i])
            for a_, b_ in zip(recs_1, recs_b[i]):
                assert a_.keys() == b_.keys()
                for k in a_:
                    if k != "cat":
                        assert np.allclose(np.array(a_[k]), np.array(b_[k]), rtol=1e-5, atol=1e-5), (i, k)



def test_checkpoint_round_trip_keeps_device_learning_rates(cuda, tmp_path):
    """Advisor r2: a checkpoint written from the GPU optimizer (tensor learning rates, capturable fused AdamW) must be the
    reference's plain layout on disk (floats, capturable off: the reference adopts the saved groups verbatim after
    torch.load(map_location=cpu), DGDE/utils/check_point.py:138), and loading one -- ours or the reference's -- must keep the
    optimizer's learning-rate TENSORS (a captured step addresses them), take the values, and leave the scheduler in control."""
    from dcd_amd.config import get_cfg
    from dcd_amd.data.synthetic import make_batch
    from dcd_amd.engine.trainer import (build_optimizer, build_scheduler, checkpoint_state, init_like_trained,
                                        load_checkpoint_state, train_step)
    from dcd_amd.model.detector import KeypointDetector
    cfg = get_cfg(opts=["MODEL.PRETRAIN", False, "MODEL.DEVICE", str(cuda), "MODEL.USE_SYNC_BN", False,
                        "INPUT.WIDTH_TRAIN", 320, "INPUT.HEIGHT_TRAIN", 96])
    images, targets = make_batch(1, seed=3, n_objects=3, input_size=(320, 96), device=cuda)

    def fresh():
        torch.manual_seed(0)
        model = KeypointDetector(cfg).to(cuda).train()
        init_like_trained(model)
        opt = build_optimizer(model, cfg)
        sched, _ = build_scheduler(opt, cfg)
        return model, opt, sched
    model, opt, sched = fresh()
    assert all(torch.is_tensor(g["lr"]) and g["lr"].is_cuda for g in opt.param_groups)
    train_step(model, opt, images, targets, cfg.SOLVER.GRAD_NORM_CLIP)
    for g in opt.param_groups:
        g["lr"].fill_(1.25e-4)
    path = str(tmp_path / "ck.pth")
    live = {id(st): {k: v for k, v in st.items()} for st in opt.state.values()}
    torch.save(checkpoint_state(model, opt, sched, iteration=1, iter_per_epoch=10), path)
    # advisor r3: writing a checkpoint must not touch the RUNNING optimizer (state_dict() hands out its live dicts): same
    # tensor objects, still on the device, and the same optimizer takes another fused step afterwards
    for st in opt.state.values():
        for k, v in st.items():
            assert v is live[id(st)][k] and (not torch.is_tensor(v) or v.is_cuda)
    w_before = [p.detach().clone() for p in model.parameters()]
    train_step(model, opt, images, targets, cfg.SOLVER.GRAD_NORM_CLIP)
    assert any(not torch.equal(a, p) for a, p in zip(w_before, model.parameters()))
    assert all(abs(float(st["step"]) - 2.0) < 1e-6 for st in opt.state.values())
    data = torch.load(path, map_location="cpu", weights_only=False)
    assert all(abs(float(st["step"]) - 1.0) < 1e-6 for st in data["optimizer"]["state"].values())
    for g in data["optimizer"]["param_groups"]:
        assert isinstance(g["lr"], float) and abs(g["lr"] - 1.25e-4) < 1e-9 and g["capturable"] is False and g["fused"] is None
        assert not torch.is_tensor(g.get("initial_lr", 0.0))
    assert all(not v.is_cuda for st in data["optimizer"]["state"].values() for v in st.values() if torch.is_tensor(v))
    model2, opt2, sched2 = fresh()
    lr_objs = [g["lr"] for g in opt2.param_groups]
    extras = load_checkpoint_state(data, model2, opt2, sched2)
    assert extras["iteration"] == 1
    for g, t in zip(opt2.param_groups, lr_objs):
        assert g["lr"] is t and t.is_cuda and abs(float(t) - 1.25e-4) < 1e-10
    before = [p.detach().clone() for p in model2.parameters()]
    train_step(model2, opt2, images, targets, cfg.SOLVER.GRAD_NORM_CLIP)           # fused capturable AdamW with the restored state
    assert any(not torch.equal(a, p) for a, p in zip(before, model2.parameters()))
    for g in opt2.param_groups:
        g["lr"].fill_(7e-5)                                                         # what a scheduler step does to a tensor rate
    assert all(g["lr"] is t and abs(float(t) - 7e-5) < 1e-10 for g, t in zip(opt2.param_groups, lr_objs))


def test_whole_train_step_at_baseline_size(cuda):
    """PARITY: step 1 only (same weights in all four modes): graph == eager at 1e-4, split-bf16 within 1e-3 and MODEL.FP16 within 3 % of
    the fp32 total loss.  SMOKE: everything about steps 2-3 (finite, falling, within 10 % / 25 % / 1.5x / 3x -- run-to-run spread).

    VERDICT r3 item 8a -- BASELINE configs[1] as a test: one whole train step (forward, 13-term loss, backward, clip, fused
    AdamW) at bs 8, 384x1280, the size `bench.py` times.  Three steps each of (a) the eager step, (b) the whole-step HIP graph and
    (c) the eager step with the DCN products / 3x3 convolutions in split-bf16 (`bf16x3`), all from the same seed on the same
    batch.  Losses finite and falling; on the first step (same weights) graph == eager at 1e-4 and split-bf16 within north_star's
    1e-3 of the fp32 total loss; later steps within the run-to-run spread of the eager step itself (see below)."""
    import argparse
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    from dcd_amd import _ext
    from dcd_amd.engine import trainer

    def run(mode):
        args = argparse.Namespace(batch=8, objects=6, precision="bf16x3" if mode == "split" else "f32", scaling="weak", amp=mode == "fp16")
        cfg, model, optimizer, images, targets = bench.build_everything(args, cuda, 1, 0)[:5]
        clip = cfg.SOLVER.GRAD_NORM_CLIP
        step = trainer.GraphedTrainStep(model, optimizer, clip) if mode == "graph" else None
        losses = []
        try:
            for _ in range(3):
                ld, _ = step(images, targets) if step else trainer.train_step(model, optimizer, images, targets, clip)
                total = getattr(ld, "total", None)
                losses.append(float(total if total is not None else sum(ld.values())))
                assert len(ld) == 13
        finally:
            _ext.set_precision("f32")
        torch.cuda.synchronize()
        return losses

    eager, graph, split, fp16 = run("eager"), run("graph"), run("split"), run("fp16")
    # (d) MODEL.FP16 (BASELINE config 3's arithmetic at config 1's size): the bf16 precision scope around backbone and predictor
    # Step 1 (same weights) within 3 % of fp32.  Step 2 is held to 1.5x the first loss (it was "within 15 % of fp32" until the 1x1
    # convolutions got their bf16 kernels: the second loss of ONE batch overshoots by up to 30 % in every precision -- 19.2, 20.3,
    # 24.1, 27.7, 27.8 over the runs of profiles/r05_train_sanity.txt, fp32 included -- before the 120 steps converge alike).
    # Step 3 is only required to be finite and below 3x the first loss: AdamW's first updates move every weight by ~lr whatever the size of its gradient, so the SIGN of every near-zero
    # gradient component decides the third loss -- noise at fp32 accuracy already (see below), and bf16 products put many more
    # components there: the same code gave 19.9, 20.6, 25.3 and 34.1 at step 3 over the runs of one day (fp32: 19.2-20.2), while
    # the 120-step run of the same step converges like fp32 (profiles/r05_train_sanity.txt: 21.6 -> 4.47 against 4.57).
    assert abs(fp16[0] - eager[0]) <= 0.03 * eager[0], (eager, fp16)
    assert fp16[1] == fp16[1] and 0.0 < fp16[1] < 1.5 * eager[0], (eager, fp16)
    assert fp16[2] == fp16[2] and 0.0 < fp16[2] < 3.0 * eager[0] and min(fp16[1:]) < fp16[0], (eager, fp16)
    for ls in (eager, graph, split):
        assert all(l == l and 0.0 < l < 1e4 for l in ls), ls
        # three steps on one batch: the loss comes down -- not necessarily monotonically (the same sign noise: a 120-step run of
        # this step at bs 4 goes 21.6, 21.2, 24.1, 17.2, ... 4.6 in fp32 and 21.6, 21.2, 19.2, 17.2, ... 4.5 in mixed precision,
        # profiles/r05_train_sanity.txt)
        assert min(ls[1:]) < ls[0] and ls[2] < 1.5 * ls[0], ls
    # step 1 starts from identical weights: graph == eager to fp32 noise, split-bf16 within north_star's 1e-3.  Steps 2-3 follow
    # AdamW's first updates, which move every weight by ~lr whatever the size of its gradient -- the sign of a near-zero gradient
    # component is noise (fp32 atomics), so two runs of the SAME eager code are a few per cent apart by step 3 (measured here:
    # 21.82 / 19.48 against 21.76 / 20.21); the second step is therefore held to 10 %, the third to 25 %, the trend to "some later step below the first".
    assert abs(graph[0] - eager[0]) <= 1e-4 * eager[0], (eager, graph)
    assert abs(split[0] - eager[0]) <= 1e-3 * eager[0], (eager, split)
    for other in (graph, split):
        assert abs(eager[1] - other[1]) <= 0.1 * eager[1] and abs(eager[2] - other[2]) <= 0.25 * eager[2], (eager, other)


def test_graphed_training_gradients_track_an_eager_twin(cuda):
    """The whole-step HIP graph TRAINING (learning rate on) at bs 8, 384x1280: every five replays an eager twin takes the graphed
    model's weights and buffers and runs `trainer.train_step` with its learning rates at zero; the next replay's (clipped)
    gradients -- computed from the same weights -- must be the twin's.  split-bf16 products (forward and backward repeat to
    5e-6 eagerly; fp32 does not: MIOpen's stride-2 solvers split K with atomics), the DCN launch policy pinned to "never hand over"
    (`_ext.set_handover`: the graph keeps the launch sequence of its capture, an unpinned twin re-decides per call -- two correct
    sequences a seventh digit apart, which the loss's discrete pair selection can amplify): bar 2e-3 of each gradient's range.

    Round 5 found this the hard way: ATen reductions whose scratch semaphore is cleared by hipMemsetAsync (the broadcast backward
    of the head trunks' scale / shift, a flat bias-gradient sum) and MIOpen's memset + accumulate backward-data solver became
    memset nodes of the captured graph and returned wrong sums in some replays -- one trunk's BatchNorm gradients up to 85x too
    large at replay 10, 20, 25, ..., invisible in the loss of the first steps, and a graphed training run left the eager
    trajectory after ~40 steps (profiles/r05_graph_vs_eager.txt).  Every such site now reduces in kernels without memsets."""
    import argparse
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    from dcd_amd import _ext
    from dcd_amd.engine import trainer

    def build(zero_lr):
        args = argparse.Namespace(batch=8, objects=6, precision="bf16x3", scaling="weak", amp=False)
        r = bench.build_everything(args, cuda, 1, 0)[:5]
        if zero_lr:
            for g in r[2].param_groups:
                g["lr"].fill_(0.0)
                g["weight_decay"] = 0.0
        return r
    try:
        _ext.set_handover("never")
        cfg, A, optA, images, targets = build(False)
        _, B, optB, _, _ = build(True)
        clip = cfg.SOLVER.GRAD_NORM_CLIP
        step = trainer.GraphedTrainStep(A, optA, clip)
        worst_seen = 0.0
        for it in range(41):
            check = it > 0 and it % 5 == 0                    # (the wrong sums came and went: 7 of 12 such checks in one run, 0 of 4 in another)
            if check:
                torch.cuda.synchronize()
                with torch.no_grad():
                    for p, q in zip(A.parameters(), B.parameters()):
                        q.copy_(p)
                    for p, q in zip(A.buffers(), B.buffers()):
                        q.copy_(p)
                ld_b, _ = trainer.train_step(B, optB, images, targets, clip)
                torch.cuda.synchronize()
                gb = {n: p.grad.detach().clone() for n, p in B.named_parameters() if p.grad is not None}
            ld, _ = step(images, targets)
            if check:
                torch.cuda.synchronize()
                assert abs(float(sum(ld.values())) - float(sum(ld_b.values()))) <= 1e-5 * float(sum(ld_b.values()))
                for n, p in A.named_parameters():
                    # (a bias in front of a BatchNorm has a zero gradient: what is there is rounding noise of either run)
                    if p.grad is None or n.endswith("conv.bias") or float(gb[n].abs().max()) < 1e-7:
                        continue
                    rel = float((p.grad - gb[n]).abs().max() / gb[n].abs().max())
                    worst_seen = max(worst_seen, rel)
                    # Typical worst value over the eight checks: 3e-5 (five runs: 2.7e-5 .. 3.6e-5).  With the policy unpinned about one
                    # run in 15 showed 1e-3 .. 5e-3 in the heads at replay 35-40 (graph and twin on different, equally correct DCN
                    # launch sequences: tools/twin_catch.py, docs/HISTORY.md); the hazard this test exists for gave 85x.
                    assert rel <= 2e-3, "replay %d: gradient of %s differs from the eager twin's by %.2e of its range" % (it, n, rel)
        assert worst_seen > 0.0
        if os.environ.get("DCD_TEST_PRINT_GRAD_DEV"):
            print("worst graph-vs-twin gradient difference %.3e of a tensor's range" % worst_seen)
    finally:
        _ext.set_handover(None)
        _ext.set_precision("f32")
