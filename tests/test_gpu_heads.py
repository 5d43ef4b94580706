"""GPU parity: edge-depth solver, focal, GIoU, heat-map decode and POI gather (C ABI via dcd_amd.ops)
against the numpy oracle (oracle/heads_oracle.py) on identical seeded inputs.

Bars: indices (top-k pair sets, heat-map winners) bit-exact; floating point within 1e-3 relative
(north_star), with the actual bound written next to each assert.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

P2 = np.array([[721.5377, 0, 609.5593, 44.85728], [0, 721.5377, 172.854, 0.2163791], [0, 0, 1, 0.002745884]],
              dtype=np.float32)


def synth_objects(N, K=73, seed=0, noise=0.0):
    """Objects with exact projections: returns kps (N,K,2), kps3d (N,K,3), rot (N,1), P, z."""
    rng = np.random.RandomState(seed)
    z = rng.uniform(8, 50, N).astype(np.float32)
    x = (rng.uniform(-0.25, 0.25, N) * z).astype(np.float32)
    y = np.full(N, 1.0, np.float32)
    dims = np.stack([rng.normal(3.9, 0.3, N), rng.normal(1.5, 0.1, N), rng.normal(1.6, 0.1, N)], 1).astype(np.float32)
    rot = rng.uniform(-np.pi, np.pi, N).astype(np.float32)
    k3 = (rng.uniform(-0.5, 0.5, (N, K, 3)) * dims[:, None, :]).astype(np.float32)
    c, s = np.cos(rot)[:, None], np.sin(rot)[:, None]
    Xc = k3[:, :, 0] * c + k3[:, :, 2] * s + x[:, None]
    Yc = k3[:, :, 1] + y[:, None]
    Zc = -k3[:, :, 0] * s + k3[:, :, 2] * c + z[:, None]
    u = (P2[0, 0] * Xc + P2[0, 2] * Zc + P2[0, 3]) / (Zc + P2[2, 3])
    v = (P2[1, 1] * Yc + P2[1, 2] * Zc + P2[1, 3]) / (Zc + P2[2, 3])
    kps = np.stack([u, v], -1).astype(np.float32)
    kps += rng.normal(0, noise, kps.shape).astype(np.float32)
    P = np.tile(P2[None], (N, 1, 1))
    return kps, k3, rot[:, None], P, z


@pytest.mark.parametrize("N,K", [(12, 73), (1, 73), (5, 10), (3, 128), (40, 73)])
def test_edge_depth_eval(cuda, N, K):
    from dcd_amd import ops
    from oracle import heads_oracle as ho
    kps, k3, rot, P, _ = synth_objects(N, K, seed=N + K, noise=0.5)
    ref, _, _ = ho.pairs_kpts_depth(kps, k3, rot, P, training=False)
    got, gm = ops.pairs_kpts_depth(*(torch.from_numpy(a).to(cuda) for a in (kps, k3, rot, P)), training=False)
    assert gm is None and tuple(got.shape) == (N, K * (K - 1) // 2)
    # bound: 2e-4 relative (device vs host sin/cos, amplified by cancellation in the pair differences); north_star allows 1e-3
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("N", [12, 1, 33])
def test_edge_depth_train_topk_exact(cuda, N):
    from dcd_amd import ops
    from oracle import heads_oracle as ho
    K = 73
    kps, k3, rot, P, _ = synth_objects(N, K, seed=N, noise=0.3)
    rng = np.random.RandomState(N)
    mask = rng.rand(N, K) > 0.2
    ref, rmask, ridx = ho.pairs_kpts_depth(kps, k3, rot, P, kmask=mask, training=True)
    t = [torch.from_numpy(a).to(cuda) for a in (kps, k3, rot, P)]
    depth, idx, pmask = ops._PairsDepth.apply(t[0], t[1], t[2], t[3], torch.from_numpy(mask).to(cuda), 1500, 2.0, 80.0, 0, 1)
    assert np.array_equal(idx.cpu().numpy().astype(np.int64), ridx), "top-1500 pair indices must be bit-exact"
    assert np.array_equal(pmask.cpu().numpy(), rmask)
    np.testing.assert_allclose(depth.cpu().numpy(), ref, rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("K,topk", [(64, 1500), (65, 1500), (91, 1500), (91, 4095), (92, 1500), (20, 100)])
def test_edge_depth_topk_order_at_both_forms_of_the_sort(cuda, K, topk):
    """PARITY.  The ordering runs as a register / lane-exchange network when the pairs pad to 4 096 (65 <= K <= 91: the reference's
    73 keypoints) and as LDS passes otherwise: the pair indices are bit-exact on both sides of both boundaries, with ties."""
    from dcd_amd import ops
    from oracle import heads_oracle as ho
    N = 7
    kps, k3, rot, P, _ = synth_objects(N, K, seed=K, noise=0.3)
    kps[:, K // 2:, 1] = kps[:, K // 2:K // 2 + 1, 1]          # half the keypoints share a row: ties at dv == 0
    ref, _, ridx = ho.pairs_kpts_depth(kps, k3, rot, P, training=True, num_k=topk)
    t = [torch.from_numpy(a).to(cuda) for a in (kps, k3, rot, P)]
    depth, idx, _ = ops._PairsDepth.apply(t[0], t[1], t[2], t[3], None, topk, 2.0, 80.0, 0, 1)
    assert np.array_equal(idx.cpu().numpy().astype(np.int64), ridx)
    np.testing.assert_allclose(depth.cpu().numpy(), ref, rtol=2e-4, atol=2e-4)


def test_edge_depth_ties_lower_index_first(cuda):
    """Many identical v (degenerate keypoints) -> ties in |dv|; rule: lower pair index first."""
    from dcd_amd import ops
    from oracle import heads_oracle as ho
    kps, k3, rot, P, _ = synth_objects(4, 73, seed=9)
    kps[:, 20:, 1] = kps[:, 20:21, 1]          # 53 keypoints share one row -> 1378 pairs with dv == 0
    ref, _, ridx = ho.pairs_kpts_depth(kps, k3, rot, P, training=True)
    t = [torch.from_numpy(a).to(cuda) for a in (kps, k3, rot, P)]
    depth, idx, _ = ops._PairsDepth.apply(t[0], t[1], t[2], t[3], None, 1500, 2.0, 80.0, 0, 1)
    assert np.array_equal(idx.cpu().numpy().astype(np.int64), ridx)
    np.testing.assert_allclose(depth.cpu().numpy(), ref, rtol=2e-4, atol=2e-4)


def test_edge_depth_recovers_true_depth(cuda):
    """Domain property at any size: on exact projections the mean over pairs is the object depth."""
    from dcd_amd import ops
    kps, k3, rot, P, z = synth_objects(64, 73, seed=3, noise=0.0)
    got, _ = ops.pairs_kpts_depth(*(torch.from_numpy(a).to(cuda) for a in (kps, k3, rot, P)), training=True)
    est = got.mean(1).cpu().numpy()
    assert np.max(np.abs(est - z) / z) < 2e-2


def test_edge_depth_backward_matches_autograd(cuda):
    """Gradient w.r.t. kps and kps3d vs torch autograd of the restated formula (float64 on CPU).
    Eval mode uses random output weights; train mode uses sum() so the top-k ORDER cannot matter."""
    from dcd_amd import ops
    N, K = 6, 73
    kps, k3, rot, P, _ = synth_objects(N, K, seed=5, noise=0.4)
    for training in (False, True):
        a = torch.from_numpy(kps).to(cuda).requires_grad_()
        b = torch.from_numpy(k3).to(cuda).requires_grad_()
        d, _ = ops.pairs_kpts_depth(a, b, torch.from_numpy(rot).to(cuda), torch.from_numpy(P).to(cuda), training=training)
        gen = torch.Generator().manual_seed(1)
        gw = torch.ones(d.shape) if training else torch.randn(d.shape, generator=gen)
        (d * gw.to(cuda)).sum().backward()
        A = torch.from_numpy(kps).double().requires_grad_()
        Bk = torch.from_numpy(k3).double().requires_grad_()
        Pt = torch.from_numpy(P).double()
        r = torch.from_numpy(rot).double()
        v = (A[:, :, 1] - Pt[:, None, 1, 2]) / Pt[:, None, 1, 1]
        C = Bk[:, :, 0] * torch.sin(r) - Bk[:, :, 2] * torch.cos(r)
        H2 = v * C
        iu = torch.triu_indices(K, K, 1)
        hm = (Bk[:, iu[0], 1] - Bk[:, iu[1], 1]) + (H2[:, iu[0]] - H2[:, iu[1]])
        dv = (v[:, iu[0]] - v[:, iu[1]]).abs()
        zz = (hm.abs() / dv.clamp_min(1e-10)).clamp_min(2.0).clamp_max(80.0)
        if training:
            _, idx = torch.topk(dv, 1500, dim=-1)
            zz = zz.gather(-1, idx)
        (zz * gw.double()).sum().backward()
        for got, ref, nm in ((a.grad, A.grad, "kps"), (b.grad, Bk.grad, "kps3d")):
            ref = ref.float()
            err = (got.cpu() - ref).abs().max().item()
            scale = ref.abs().max().item()
            assert err <= 1e-3 * scale, (nm, training, err, scale)   # 1e-3 relative (north_star)


def test_compute_z_gmw(cuda):
    from dcd_amd import ops
    from oracle import heads_oracle as ho
    kps, k3, rot, P, _ = synth_objects(7, 73, seed=2, noise=0.2)
    kn = kps.copy()
    kn[:, :, 0] = (kps[:, :, 0] - P2[0, 2]) / P2[0, 0]
    kn[:, :, 1] = (kps[:, :, 1] - P2[1, 2]) / P2[1, 1]
    ref, _, _ = ho.pairs_kpts_depth(kn, k3, rot, P, training=False, zmin=0.1, normalized=True, sub_b3=False)
    _, _, ridx = ho.pairs_kpts_depth(kn, k3, rot, P, training=True, zmin=0.1, normalized=True, sub_b3=False)
    z, idx = ops.compute_z(torch.from_numpy(kn).to(cuda), torch.from_numpy(k3).to(cuda), torch.from_numpy(rot).to(cuda))
    np.testing.assert_allclose(z.cpu().numpy(), ref, rtol=2e-4, atol=2e-4)
    assert np.array_equal(idx.cpu().numpy(), ridx)


# ---------------------------------------------------------------------------------------------
def heat_and_target(B, H, W, seed):
    rng = np.random.RandomState(seed)
    pred = 1 / (1 + np.exp(-rng.normal(-2, 1.5, (B, 1, H, W))))
    pred = np.clip(pred, 1e-4, 1 - 1e-4).astype(np.float32)
    tgt = np.zeros((B, 1, H, W), np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    for b in range(B):
        for _ in range(6):
            cy, cx, s = rng.randint(0, H), rng.randint(0, W), rng.uniform(1, 4)
            g = np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s * s)).astype(np.float32)
            tgt[b, 0] = np.maximum(tgt[b, 0], g)
            tgt[b, 0, cy, cx] = 1.0
    return pred, tgt


@pytest.mark.parametrize("shape", [(2, 96, 320), (8, 96, 320), (1, 5, 7)])
def test_focal_loss(cuda, shape):
    from dcd_amd import ops
    from oracle import heads_oracle as ho
    pred, tgt = heat_and_target(*shape, seed=shape[0])
    ref_loss, ref_np = ho.focal_loss(pred, tgt)
    p = torch.from_numpy(pred).to(cuda).requires_grad_()
    loss, npos = ops.focal_loss(p, torch.from_numpy(tgt).to(cuda), 2, 4)
    assert npos.item() == ref_np
    assert abs(loss.item() - ref_loss) <= 1e-4 * abs(ref_loss)          # bound 1e-4 relative
    loss.backward()
    # gradient vs torch autograd of the reference formula (CPU float64)
    pt = torch.from_numpy(pred).double().requires_grad_()
    tt = torch.from_numpy(tgt).double()
    pc = pt.clamp(1e-10, 1 - 1e-10)
    l = -(torch.log(pc) * (1 - pc) ** 2 * (tt == 1)) - torch.log(1 - pc) * pc ** 2 * (1 - tt) ** 4 * ((tt < 1) & (tt >= 0))
    l.sum().backward()
    err = (p.grad.cpu().double() - pt.grad).abs().max().item()
    assert err <= 1e-4 * pt.grad.abs().max().item()


def test_giou_loss(cuda):
    from dcd_amd import ops
    from oracle import heads_oracle as ho
    rng = np.random.RandomState(0)
    pred = rng.uniform(0, 30, (64, 4)).astype(np.float32)
    tgt = rng.uniform(0.5, 30, (64, 4)).astype(np.float32)
    pred[:4] = 0.0                                    # relu'd predictions are often exactly zero
    rl, ri = ho.giou_loss(pred, tgt)
    p = torch.from_numpy(pred).to(cuda).requires_grad_()
    losses, ious = ops.giou_loss(p, torch.from_numpy(tgt).to(cuda))
    np.testing.assert_allclose(losses.detach().cpu().numpy(), rl, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(ious.cpu().numpy(), ri, rtol=1e-5, atol=1e-6)
    losses.sum().backward()
    pt = torch.from_numpy(pred).double().requires_grad_()
    tt = torch.from_numpy(tgt).double()
    ta = (tt[:, 0] + tt[:, 2]) * (tt[:, 1] + tt[:, 3])
    pa = (pt[:, 0] + pt[:, 2]) * (pt[:, 1] + pt[:, 3])
    wi = torch.min(pt[:, 0], tt[:, 0]) + torch.min(pt[:, 2], tt[:, 2])
    gw = torch.max(pt[:, 0], tt[:, 0]) + torch.max(pt[:, 2], tt[:, 2])
    hi = torch.min(pt[:, 3], tt[:, 3]) + torch.min(pt[:, 1], tt[:, 1])
    gh = torch.max(pt[:, 3], tt[:, 3]) + torch.max(pt[:, 1], tt[:, 1])
    ac = gw * gh + 1e-7
    ai = wi * hi
    au = ta + pa - ai
    iou = (ai + 1) / (au + 1)
    (1 - (iou - (ac - au) / ac)).sum().backward()
    err = (p.grad.cpu().double() - pt.grad).abs().max().item()
    assert err <= 1e-4 * pt.grad.abs().max().item() + 1e-7


@pytest.mark.parametrize("B,C,H,W,K", [(2, 1, 96, 320, 50), (16, 1, 96, 320, 50), (2, 3, 24, 40, 50), (1, 1, 8, 8, 50),
                                      (3, 1, 96, 320, 100)])
def test_heatmap_decode_exact(cuda, B, C, H, W, K):
    from dcd_amd import ops
    from oracle import heads_oracle as ho
    rng = np.random.RandomState(B * 7 + C)
    heat = np.clip(1 / (1 + np.exp(-rng.normal(-2, 1.5, (B, C, H, W)))), 1e-4, 1 - 1e-4).astype(np.float32)
    # inject ties: plateaus of equal values and duplicated peaks
    heat[:, :, 3:5, 3:6] = 0.77
    heat[:, :, H - 1, W - 1] = 0.9
    heat[:, :, 0, 0] = 0.9
    ref_nms = ho.nms_hm(heat)
    h = torch.from_numpy(heat).to(cuda)
    got_nms = ops.nms_hm(h)
    assert np.array_equal(got_nms.cpu().numpy(), ref_nms), "nms_hm must be bit-exact"
    ref = ho.select_topk(ref_nms, K)
    for fused in (False, True):
        got = ops.select_topk(h if fused else got_nms, K, fuse_nms=fused)
        for g_, r_, nm in zip(got, ref, ("scores", "inds", "clses", "ys", "xs")):
            assert np.array_equal(g_.cpu().numpy(), r_), "%s (fused=%s) must be bit-exact" % (nm, fused)


def test_heatmap_few_maxima(cuda):
    """Fewer than K non-zero maxima: zeros fill the tail in index order."""
    from dcd_amd import ops
    from oracle import heads_oracle as ho
    heat = np.zeros((2, 1, 96, 320), np.float32)
    heat[0, 0, 10, 10] = 0.5
    heat[0, 0, 50, 300] = 0.7
    heat[1, 0, 95, 319] = 0.3
    ref = ho.select_topk(ho.nms_hm(heat), 50)
    got = ops.select_topk(torch.from_numpy(heat).to(cuda), 50, fuse_nms=True)
    for g_, r_ in zip(got, ref):
        assert np.array_equal(g_.cpu().numpy(), r_)


def test_poi_gather_and_scatter(cuda):
    from dcd_amd import ops
    from oracle import heads_oracle as ho
    rng = np.random.RandomState(4)
    B, C, H, W, M = 3, 415, 24, 40, 40
    feat = rng.normal(size=(B, C, H, W)).astype(np.float32)
    pts = np.stack([rng.randint(0, W, (B, M)), rng.randint(0, H, (B, M))], -1).astype(np.int32)
    pts[:, 1] = pts[:, 0]                              # duplicate centres must accumulate in backward
    ref = ho.select_point_of_interest(pts.astype(np.int64), feat)
    f = torch.from_numpy(feat).to(cuda).requires_grad_()
    got = ops.select_point_of_interest(B, torch.from_numpy(pts).to(cuda), f)
    assert np.array_equal(got.detach().cpu().numpy(), ref)
    gen = torch.Generator().manual_seed(0)
    go = torch.randn(got.shape, generator=gen)
    got.backward(go.to(cuda))
    ft = torch.from_numpy(feat).requires_grad_()
    idx = torch.from_numpy((pts[:, :, 1] * W + pts[:, :, 0]).astype(np.int64))
    r = ft.permute(0, 2, 3, 1).reshape(B, H * W, C).gather(1, idx[:, :, None].expand(B, M, C))
    r.backward(go)
    assert (f.grad.cpu() - ft.grad).abs().max().item() < 1e-5


def test_target_encoding_matches_reference_fixture(cuda):
    """`dcd_encode_targets` (one launch per batch, csrc/targets.hip) against the fixture the REFERENCE's
    `KITTIDataset.__getitem__` produced (tests/golden/make_golden_targets.py): three images in one batch, two image sizes,
    truncated objects (approximate centres + 1-D edge heat maps), filtered / behind-camera objects, an object without
    point-cloud key points.  Integer / mask fields bit-exact, float fields to 1e-6 of the field's range."""
    import test_oracle_targets as OT
    from dcd_amd.config import get_cfg
    from dcd_amd.data.target_encoder import encode_targets
    g = OT.load()
    cfg = get_cfg(opts=["MODEL.PRETRAIN", False, "MODEL.DEVICE", str(cuda)])
    n = int(g["n_images"])
    samples = [OT.raw_inputs(g, i) for i in range(n)]
    targets = encode_targets(samples, cfg, cuda)
    assert len(targets) == n
    for i, t in enumerate(targets):
        got = {name: t.get_field(name).cpu().numpy() for name in t.fields() if torch.is_tensor(t.get_field(name))}
        OT.compare(got, g, i, 1e-6)
        assert tuple(t.size) == tuple(int(v) for v in g["out%d_size" % i])
        for name in got:                                  # dtypes the model relies on
            ref = g["out%d_%s" % (i, name)] if "out%d_%s" % (i, name) in g.files else None
            if ref is not None and name not in ("edge_len", "final_output_w", "final_output_h"):
                assert got[name].dtype == ref.dtype or (ref.dtype == np.bool_ and got[name].dtype == np.bool_), (name, got[name].dtype, ref.dtype)


def test_target_encoding_matches_oracle_on_random_scenes(cuda):
    """The same kernel against the numpy oracle on 16 seeded random scenes in ONE batch (up to 12 objects each, many near or
    beyond the image border): every field of every image."""
    from oracle import target_oracle as TO
    import test_oracle_targets as OT
    from dcd_amd.config import get_cfg
    from dcd_amd.data.target_encoder import encode_targets
    cfg = get_cfg(opts=["MODEL.PRETRAIN", False, "MODEL.DEVICE", str(cuda)])
    P = np.array([[721.5377, 0.0, 609.5593, 44.85728], [0.0, 721.5377, 172.854, 0.2163791], [0.0, 0.0, 1.0, 0.002745884]])
    rng = np.random.RandomState(3)
    samples = []
    for b in range(16):
        n = rng.randint(1, 13)
        z = rng.uniform(4, 60, n)
        x = rng.uniform(-0.62, 0.62, n) * z
        hwl = np.stack([rng.normal(1.5, 0.1, n), rng.normal(1.6, 0.1, n), rng.normal(3.9, 0.3, n)], 1)
        t = np.stack([x, np.full(n, 1.65), z], 1).astype(np.float32)
        ry = rng.uniform(-np.pi, np.pi, n)
        alpha = ry - np.arctan2(t[:, 0], t[:, 2])
        alpha = (alpha + np.pi) % (2 * np.pi) - np.pi
        u = (721.5377 * t[:, 0] + 609.5593 * t[:, 2]) / t[:, 2]
        box = np.stack([np.clip(u - 400 / z, 0, 1241), np.clip(172 - 500 / z, 0, 374), np.clip(u + 400 / z, 0, 1241),
                        np.clip(172 + 700 / z, 0, 374)], 1).astype(np.float32)
        k3 = rng.uniform(-0.5, 0.5, (n, 63, 3)) * hwl[:, None, [2, 0, 1]]
        k3[:, :, 1] -= hwl[:, 0][:, None] / 2
        samples.append(dict(image_size=np.array([1242, 375]), P=P, trunc_occ=np.stack([rng.uniform(0, 1, n), rng.randint(0, 3, n).astype(float)], 1),
                            box2d=box, hwl=hwl, t=t, ry=ry, alpha=alpha, find_pcl=rng.randint(0, 2, n).astype(np.int32), kpts3d=k3))
    targets = encode_targets(samples, cfg, cuda)
    kept = 0
    for s, t in zip(samples, targets):
        try:
            ref = TO.encode_image(**s)
        except (TypeError, AssertionError):      # the reference itself fails on such a sample (approx centre without a box centre inside)
            continue
        for name, val in ref.items():
            got = t.get_field(name).cpu().numpy()
            if name in OT.INT_FIELDS:
                np.testing.assert_array_equal(got.astype(np.int64), np.asarray(val).astype(np.int64), err_msg=name)
            else:
                scale = max(np.abs(val).max(), 1.0)
                assert np.abs(got.astype(np.float64) - np.asarray(val, np.float64)).max() <= 1e-6 * scale, name
        kept += int(ref["reg_mask"].sum())
    assert kept > 40


@pytest.mark.parametrize("relu,norm", [(True, True), (False, True), (True, False)])
def test_edge_branch_on_the_device_matches_the_stock_modules(cuda, relu, norm):
    """EdgeBranch (detector_predictor.py: GEMM-form Conv1d + BatchNorm1d / ReLU on csrc/norm.hip) against the same four stock
    modules evaluated in float64 at the size of the step (8 x 256 x 832 border cells): output, input gradient, every parameter
    gradient and the BatchNorm running estimates, 2e-5 of each tensor's scale (fp32 sums in another order)."""
    from torch import nn
    from dcd_amd.model.head.detector_predictor import EdgeBranch
    torch.manual_seed(3)
    C, K, B = 256, 832, 8

    def mods():
        return (nn.Conv1d(C, C, 3, padding=1, padding_mode="replicate"), nn.BatchNorm1d(C) if norm else nn.Identity(),
                nn.ReLU(inplace=True) if relu else nn.Identity(), nn.Conv1d(C, 2, 1))
    fast = EdgeBranch(*mods()).to(cuda).train()
    ref = nn.Sequential(*mods()).double().train()
    ref.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in fast.state_dict().items()})
    if norm:
        with torch.no_grad():
            fast[1].weight.uniform_(0.5, 1.5)
            fast[1].bias.uniform_(-0.5, 0.5)
            ref[1].weight.copy_(fast[1].weight.double().cpu())
            ref[1].bias.copy_(fast[1].bias.double().cpu())
    x = torch.randn(B, C, K)
    g = torch.randn(B, 2, K)
    xf = x.to(cuda).transpose(1, 2).contiguous().transpose(1, 2).requires_grad_(True)    # a (B, K, C) buffer viewed as (B, C, K), like the gather's
    xr = x.double().requires_grad_(True)
    yf = fast(xf)
    yf.backward(g.to(cuda))
    yr = ref(xr)
    yr.backward(g.double())

    def close(a, b, what):
        err = (a.double().cpu() - b).abs().max().item()
        assert err <= 2e-5 * max(b.abs().max().item(), 1e-6), (what, err, b.abs().max().item())
    close(yf, yr, "output")
    close(xf.grad, xr.grad, "grad_input")
    for (n, p), q in zip(fast.named_parameters(), ref.parameters()):
        if n == "0.bias" and norm:
            # in front of a BatchNorm the bias gradient is exactly zero (the mean is removed): what is left is the fp32 rounding
            # of a sum of B * K = 6 656 gradient values of order one that cancel
            assert p.grad.abs().max().item() <= 1e-4 and q.grad.abs().max().item() <= 1e-10
            continue
        close(p.grad, q.grad, n)
    if norm:
        close(fast[1].running_mean, ref[1].running_mean, "running_mean")
        close(fast[1].running_var, ref[1].running_var, "running_var")
        assert int(fast[1].num_batches_tracked) == 1
