"""Generates tests/golden/gmw.npz by RUNNING THE REFERENCE'S GMW CODE (imported from /root/reference/GMW) on the CPU:
one train step of GMW/main.py:447-466 -- compute_z, GMW.forward, correspondenceLoss, compute_reg_loss, backward -- on a
seeded batch of 2 objects x 73 keypoints (2628 edges).

Run in the build container only (`python tests/golden/make_golden_gmw.py`, its own process: GMW's top-level packages are
called `model`, `lib`, `utilities` like DGDE's).  Nothing is copied: the modules are imported where they lie.
  * `cv2` gets an empty stand-in (imported by GMW/model/model.py:6, never called);
  * GMW/main.py itself is not importable (tensorboard, matplotlib, argparse at import time): `compute_z`, `get_up` and
    `compute_reg_loss` are compiled from the file's AST and executed as they are;
  * `torch.cholesky` (removed from torch 2.x, used at GMW/lib/optimal_transport.py:111) is aliased to
    `torch.linalg.cholesky` -- the same lower-triangular factorisation under its current name.
The model is initialised by `torch.manual_seed(0)` + the reference constructor; the fixture stores a checksum of every
parameter so the test can prove that our constructor consumes the generator identically, instead of 9 MB of weights.
"""
import ast
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/GMW"


def inputs(seed=7, B=2, K=73):
    """Seeded GMW batch: K-normalised 2-D keypoints, object-frame 3-D keypoints, yaw, location (shared with the test)."""
    rng = np.random.default_rng(seed)
    dims = np.array([3.9, 1.5, 1.6], dtype=np.float32)
    k3 = ((rng.random((B, K, 3)) - 0.5) * dims).astype(np.float32)
    rot = (rng.random((B, 1)) * 2 * np.pi - np.pi).astype(np.float32)
    loc = np.stack([(rng.random(B) - 0.5) * 10, np.full(B, 1.65), 8 + rng.random(B) * 40], 1).astype(np.float32)
    c, s = np.cos(rot[:, 0]), np.sin(rot[:, 0])
    xc = k3[:, :, 0] * c[:, None] + k3[:, :, 2] * s[:, None] + loc[:, None, 0]
    yc = k3[:, :, 1] + loc[:, None, 1]
    zc = -k3[:, :, 0] * s[:, None] + k3[:, :, 2] * c[:, None] + loc[:, None, 2]
    k2 = np.stack([xc / zc, yc / zc], -1).astype(np.float32)
    k2 += (rng.standard_normal(k2.shape) * 2e-3).astype(np.float32)
    return k2, k3, rot, loc


def main():
    sys.argv = sys.argv[:1]                       # yi2018cvpr/config.py parses the command line
    sys.modules["cv2"] = types.ModuleType("cv2")
    if not hasattr(torch, "cholesky") or True:
        torch.cholesky = torch.linalg.cholesky
    sys.path.insert(0, REF)
    from model.model import GMW                   # noqa: E402
    from lib.losses import correspondenceLoss     # noqa: E402

    tree = ast.parse(open(os.path.join(REF, "main.py")).read())
    fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("compute_z", "get_up", "compute_reg_loss")]
    ns = {"torch": torch}
    exec(compile(ast.Module(body=fns, type_ignores=[]), "GMW/main.py", "exec"), ns)

    torch.manual_seed(0)
    model = GMW(None).train()
    k2, k3, rot, loc = (torch.from_numpy(a) for a in inputs())
    out = {}
    names = [n for n, _ in model.named_parameters()]
    out["param_names"] = np.array(names)
    out["param_sums"] = np.array([float(p.double().sum()) for _, p in model.named_parameters()])
    out["param_abs_sums"] = np.array([float(p.double().abs().sum()) for _, p in model.named_parameters()])

    pre_depths, good_idx = ns["compute_z"](k2, k3, rot)
    reg_weights, edge_P = model(k2, k3, rot, None)
    edge_P.retain_grad()
    cls_loss = correspondenceLoss(edge_P, torch.eye(edge_P.shape[1]).expand_as(edge_P))
    reg_loss, pred_depth = ns["compute_reg_loss"](pre_depths, reg_weights, loc[:, -1], good_idx)
    loss = 0.1 * cls_loss + 1.0 * reg_loss           # the weights of the regression phase, GMW/main.py:313-315
    loss.backward()

    out["reg_weights"] = reg_weights.detach().numpy()
    out["pred_depth"] = pred_depth.detach().numpy()
    out["cls_loss"], out["reg_loss"], out["loss"] = float(cls_loss), float(reg_loss), float(loss)
    P = edge_P.detach()
    out["P_diag"] = P.diagonal(dim1=-2, dim2=-1).numpy()
    out["P_row_sums"], out["P_col_sums"] = P.sum(-1).numpy(), P.sum(-2).numpy()
    out["P_block"] = P[:, :64, :64].numpy()
    out["P_sum_sq"] = np.array([float((P[b].double() ** 2).sum()) for b in range(P.shape[0])])
    out["grad_norms"] = np.array([float(p.grad.double().norm()) if p.grad is not None else -1.0 for _, p in model.named_parameters()])
    out["grad_sums"] = np.array([float(p.grad.double().sum()) if p.grad is not None else 0.0 for _, p in model.named_parameters()])
    np.savez_compressed(os.path.join(HERE, "gmw.npz"), **out)
    print("gmw.npz: loss %.6f (cls %.6f reg %.6f), %d parameters, pred_depth %s vs z %s" % (
        out["loss"], out["cls_loss"], out["reg_loss"], len(names), out["pred_depth"], loc[:, -1].numpy()))
    print(names[:8])


if __name__ == "__main__":
    main()
