"""Randomised comparison of the loss-rows kernel with the op-by-op rows on the GPU: batch sizes, slot counts, masks and flags drawn
at random from the reference-pinned fixture's objects (a stress run, not a test)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
os.environ["DCD_LOSS_ROWS"] = "0"
import test_host_rows as T
from test_host_golden import small_cfg
from dcd_amd.model.head.detector_loss import Loss_Computation

dev = torch.device("cuda:0")
host_loss, tv0, pois0 = T._rows_inputs(False, False)
l0 = Loss_Computation(small_cfg("cuda:0"))
os.environ["DCD_LOSS_ROWS"] = "1"
l1 = Loss_Computation(small_cfg("cuda:0"))
B0, M0 = tv0['reg_mask'].shape
valid = tv0['reg_mask'].reshape(-1).nonzero().reshape(-1).tolist()
rng = np.random.RandomState(int(os.environ.get("SEED", "0")))
worst = 0.0
for trial in range(int(os.environ.get("TRIALS", "40"))):
    B, M = int(rng.choice([1, 2, 3, 8])), int(rng.choice([1, 4, 17, 40]))
    # every slot draws an object of the fixture (annotated or not) -- per-image fields are taken from image 0 / 1 alternately
    src = torch.from_numpy(rng.choice(valid, size=B * M))
    img = torch.arange(B) % B0
    tv = {}
    for k, v in tv0.items():
        if not torch.is_tensor(v):
            tv[k] = [v[i % B0] for i in range(B)]
        elif v.dim() >= 2 and v.shape[0] == B0 and v.shape[1] == M0 and k != 'pad_size':
            flat = v.reshape(B0 * M0, *v.shape[2:])
            tv[k] = flat[src].reshape(B, M, *v.shape[2:]).clone()
        else:
            tv[k] = v[img].clone()
    tv['reg_mask'] = torch.from_numpy((rng.rand(B, M) < 0.6).astype(np.uint8))
    if rng.rand() < 0.2:
        tv['reg_mask'][0] = 0
    tv['trunc_mask'] = torch.from_numpy((rng.rand(B, M) < 0.3).astype(np.uint8))
    tv['find_pcl'] = torch.from_numpy(rng.rand(B, M) < 0.8)
    tv['ori_mask'] = torch.from_numpy(rng.rand(B, M) < 0.8)
    tv['keypoints_depth_mask'] = torch.from_numpy((rng.rand(B, M, 3) < 0.7).astype(np.float32))
    tv['extra_kpts_2d'][..., 2] *= torch.from_numpy((rng.rand(B, M, tv['extra_kpts_2d'].shape[2]) < 0.85).astype(np.float32))
    if int(tv['reg_mask'].sum()) == 0:
        continue
    pois = torch.from_numpy(rng.normal(0, 0.5, (B, M, pois0.shape[2])).astype(np.float32))
    tvd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in tv.items()}
    gs = torch.from_numpy(rng.uniform(0.5, 1.5, 25).astype(np.float32)).to(dev)
    p0 = pois.to(dev).requires_grad_()
    S0, ix = l0._rows({'reg_pois': p0, 'reg': None}, tvd)
    (S0 * gs).sum().backward()
    p1 = pois.to(dev).requires_grad_()
    S1, _ = l1._fused_rows({'reg_pois': p1}, tvd, 1.0)
    (S1 * gs).sum().backward()
    names = {v: k for k, v in ix.items()}
    for c in range(25):
        a, b = float(S0[c]), float(S1[c])
        if a != a and b != b:
            continue
        err = abs(a - b) / max(abs(a), 1.0)
        assert err <= 1e-4, (trial, B, M, names[c], a, b)
        worst = max(worst, err)
    g0, g1 = p0.grad, p1.grad
    gerr = (g0 - g1).abs().max().item() / max(g0.abs().max().item(), 1e-6)
    assert gerr <= 1e-4 and torch.isfinite(g1).all(), (trial, B, M, gerr)
    worst = max(worst, gerr)
print("trials ok, worst relative difference %.2e" % worst)
