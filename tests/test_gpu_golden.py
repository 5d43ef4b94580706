"""GPU parity against the reference: the same fixture checks as tests/test_host_golden.py, but with the real HIP
kernels underneath (no patching) -- i.e. reference Python on CPU  vs  dcd_amd on the MI355X, identical inputs.
Tolerances: 1e-3 relative (north_star) for model-level quantities; tighter where written."""
import numpy as np
import pytest
import torch

import golden_inputs as gi
import test_host_golden as H

pytestmark = pytest.mark.gpu


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


def test_gen_data_for_gmw_matches_reference(cuda):
    H.check_gen_data(cuda, 2e-3)      # the normalised keypoints pass through the solver's mean depth: 2e-3 of the value range


def test_whole_model_matches_reference(cuda):
    """KeypointDetector on the GPU (MIOpen convs + HIP DCNv2 + HIP losses) vs the reference on CPU:
    features, predictions, all 13 losses, per-parameter gradient norms, BN statistics and the eval decode.

    Tolerance: this compares TWO stock convolution back ends (MIOpen on the GPU, oneDNN on the CPU) through ~90
    layers, 16 of which are deformable: a 1e-4 difference in a predicted sampling offset moves every later sample, so
    the end-to-end deviation (measured layer by layer with tools/diag_layers.py: <2e-4 after the encoder, growing
    ~2x per deformable stage) is set by the conv back ends, not by our kernels.  Our kernels are held to <=1e-4
    against the oracle in tests/test_gpu_dcn.py / test_gpu_heads.py, and the host logic to 2e-4 against the reference
    on identical back ends in tests/test_host_golden.py.  Here: 1e-2 on activations / losses, 3e-2 on gradient norms."""
    torch.backends.cudnn.benchmark = False
    H.check_model(cuda, 1e-2, 3e-2)


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
            else:
                assert getattr(loss_dict, "total", None) is None
                sum(loss_dict.values()).backward()
            res.append(({k: float(v) for k, v in loss_dict.items()}, c.grad.clone(), p.grad.clone(), dict(log)))
        (l0, gc0, gp0, log0), (l1, gc1, gp1, log1) = res
        for k in l0:
            assert abs(l0[k] - l1[k]) <= 1e-5 * max(abs(l0[k]), 1e-3), (seed, k, l0[k], l1[k])
        assert (gc0 - gc1).abs().max().item() <= 1e-6 * max(gc0.abs().max().item(), 1e-6)
        assert (gp0 - gp1).abs().max().item() <= 1e-5 * max(gp0.abs().max().item(), 1e-6)
        assert set(log0) == set(log1)
    assert len(lc._graphs) == 1          # one capture served all three batches
