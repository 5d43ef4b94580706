"""Seeded input builders shared by tests/golden/make_golden.py (which feeds them to the REFERENCE) and by the
parity tests (which feed them to the oracle / the HIP path).  Only numpy RandomState is used for values so the
inputs are bit-identical on every machine."""
import zlib

import numpy as np
import torch

from dcd_amd.data.calibration import Calibration, KITTI_P2
from dcd_amd.data.synthetic import make_target, scaled_P2

P2 = KITTI_P2.astype(np.float32)


def synth_objects(N, K=73, seed=0, noise=0.0):
    """Objects with exact projections: kps (N,K,2) px, kps3d (N,K,3) object frame, rot (N,1), P (N,3,4), depth (N)."""
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
    return kps, k3, rot[:, None], np.tile(P2[None], (N, 1, 1)), z


def edge_inputs():
    kps, k3, rot, P, _ = synth_objects(12, 73, seed=11, noise=0.4)
    mask = np.random.RandomState(12).rand(12, 73) > 0.2
    return kps, k3, rot, P, mask


def edge_grad_weights(shape):
    return np.random.RandomState(13).normal(size=tuple(shape)).astype(np.float32)


def normalise_kps(kps, P):
    kn = kps.copy()
    kn[:, :, 0] = (kps[:, :, 0] - P[:, None, 0, 2]) / P[:, None, 0, 0]
    kn[:, :, 1] = (kps[:, :, 1] - P[:, None, 1, 2]) / P[:, None, 1, 1]
    return kn


def heat_like(B, C, H, W, seed):
    rng = np.random.RandomState(seed)
    return np.clip(1 / (1 + np.exp(-rng.normal(-2, 1.5, (B, C, H, W)))), 1e-4, 1 - 1e-4).astype(np.float32)


def focal_inputs(B=2, H=96, W=320, seed=21):
    pred = heat_like(B, 1, H, W, seed)
    rng = np.random.RandomState(seed + 1)
    tgt = np.zeros((B, 1, H, W), np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    for b in range(B):
        for _ in range(6):
            cy, cx, s = rng.randint(0, H), rng.randint(0, W), rng.uniform(1, 4)
            g = np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s * s)).astype(np.float32)
            tgt[b, 0] = np.maximum(tgt[b, 0], g)
            tgt[b, 0, cy, cx] = 1.0
    return pred, tgt


def giou_inputs():
    rng = np.random.RandomState(31)
    pred = rng.uniform(0, 30, (64, 4)).astype(np.float32)
    tgt = rng.uniform(0.5, 30, (64, 4)).astype(np.float32)
    pred[:4] = 0.0
    return pred, tgt


def regweighted_inputs():
    rng = np.random.RandomState(41)
    return (rng.normal(size=(9, 73, 2)).astype(np.float32), rng.normal(size=(9, 73, 2)).astype(np.float32),
            np.array([1.0, 3.0, 4.999, 5.0, 5.5, 8.0, 20.0, 45.0, 70.0], np.float32))


def multibin_inputs():
    rng = np.random.RandomState(51)
    from dcd_amd.data.synthetic import encode_alpha_multibin
    vo = rng.normal(size=(10, 16)).astype(np.float32)
    go = np.stack([encode_alpha_multibin(a, 4) for a in rng.uniform(-np.pi, np.pi, 10)]).astype(np.float32)
    return vo, go


def anno_inputs(N=12, seed=61):
    rng = np.random.RandomState(seed)
    batch_idxs = np.sort(rng.randint(0, 3, N)).astype(np.int64)
    batch_idxs[:3] = [0, 1, 2]
    batch_idxs = np.sort(batch_idxs)
    P_img = np.stack([KITTI_P2 * 1.0, KITTI_P2 * 1.0, KITTI_P2 * 1.0])
    P_img[1, 0, 0] = P_img[1, 1, 1] = 718.856      # per-sequence intrinsics differ slightly in KITTI
    P_img[1, 0, 2], P_img[1, 1, 2] = 607.1928, 185.2157
    P_img[2, 0, 3] = 45.38225
    return dict(
        rotys=rng.uniform(-np.pi, np.pi, N).astype(np.float32),
        dims=np.abs(rng.normal([3.9, 1.5, 1.6], 0.3, (N, 3))).astype(np.float32),
        locs=np.stack([rng.uniform(-10, 10, N), rng.uniform(0.5, 2, N), rng.uniform(5, 60, N)], 1).astype(np.float32),
        depth_off=rng.normal(-3, 1.5, N).astype(np.float32),
        cls=np.zeros(N, np.int64),
        dims_off=rng.normal(0, 0.2, (N, 3)).astype(np.float32),
        points=np.stack([rng.randint(5, 315, N), rng.randint(1, 94, N)], 1).astype(np.float32),
        offsets=rng.uniform(0, 1, (N, 2)).astype(np.float32),
        depths=rng.uniform(5, 60, N).astype(np.float32),
        pad=np.tile(np.array([[19, 4]], np.int64), (3, 1)),
        batch_idxs=batch_idxs,
        kp10=rng.normal(0, 6, (N, 10, 2)).astype(np.float32),
        ori=rng.normal(size=(N, 16)).astype(np.float32),
        kp73=rng.normal(0, 5, (N, 73, 2)).astype(np.float32),
        P_img=P_img)


def ref_like_calibs(P_img):
    return [Calibration(P) for P in P_img]


def heat_inputs():
    return heat_like(2, 1, 96, 320, 71)      # continuous random values: no ties among the top-50


def poi_inputs():
    rng = np.random.RandomState(81)
    feat = rng.normal(size=(2, 415, 24, 80)).astype(np.float32)
    pts = np.stack([rng.randint(0, 80, (2, 40)), rng.randint(0, 24, (2, 40))], -1).astype(np.int32)
    return feat, pts


SMALL = (320, 96)       # reduced input (W, H): stride-4 map 24x80, still divisible by 32


def small_targets(with_ori_img=True):
    return [make_target(2000 + i, n_objects=3, input_size=SMALL, with_ori_img=with_ori_img) for i in range(2)]


def loss_inputs():
    rng = np.random.RandomState(91)
    cls = heat_like(2, 1, 24, 80, 92)
    reg = (rng.normal(size=(2, 415, 24, 80)) * 0.5).astype(np.float32)
    return {"cls": cls, "reg": reg}, small_targets()


def model_inputs():
    rng = np.random.RandomState(101)
    images = torch.from_numpy(rng.normal(size=(2, 3, SMALL[1], SMALL[0])).astype(np.float32))
    return images, small_targets()


def name_hashed_init(model):
    """Deterministic, construction-order independent initialisation: every parameter / buffer is filled from a
    numpy RandomState seeded by crc32 of its state-dict name.  Applied to the reference model when the fixture is
    generated and to our model in the test, so equal names <=> equal weights."""
    with torch.no_grad():
        for name, t in model.state_dict().items():
            rng = np.random.RandomState(zlib.crc32(name.encode()) & 0x7fffffff)
            if name.endswith("num_batches_tracked"):
                t.zero_()
            elif name.endswith("running_mean"):
                t.zero_()
            elif name.endswith("running_var"):
                t.fill_(1.0)
            elif "conv_offset_mask.weight" in name:
                t.copy_(torch.from_numpy(rng.normal(0, 0.01, tuple(t.shape)).astype(np.float32)))
            elif t.dim() > 1:
                fan_in = int(np.prod(t.shape[1:]))
                a = 1.0 / np.sqrt(fan_in)
                t.copy_(torch.from_numpy(rng.uniform(-a, a, tuple(t.shape)).astype(np.float32)))
            elif name.endswith("weight"):            # norm scales
                t.copy_(torch.from_numpy(rng.uniform(0.5, 1.5, tuple(t.shape)).astype(np.float32)))
            else:                                    # biases
                t.copy_(torch.from_numpy(rng.uniform(-0.1, 0.1, tuple(t.shape)).astype(np.float32)))
