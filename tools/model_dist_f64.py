"""Whole-model distances (tests/golden/model_96x320*.npz): our model on `device` vs the reference's float64 run and vs the
reference's fp32 run -- the numbers behind the tolerances of tests/test_gpu_golden.py::test_whole_model_matches_reference."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import golden_inputs as gi
import test_host_golden as H
from dcd_amd.model.detector import KeypointDetector

dev = torch.device(sys.argv[1] if len(sys.argv) > 1 else "cuda:0")
if os.environ.get("DCD_PRECISION"):                       # bf16x3 / bf16: the numbers behind the mixed-precision tests' bars
    from dcd_amd import _ext
    _ext.set_precision(os.environ["DCD_PRECISION"])
    print("precision:", _ext.get_precision())
torch.backends.cudnn.benchmark = False
g32, g64 = H.load("model_96x320"), H.load("model_96x320_f64")
model = KeypointDetector(H.small_cfg(str(dev))).to(dev)
gi.name_hashed_init(model)
model.train()
images, targets = gi.model_inputs()
images = images.to(dev)
targets = [t.to(dev) for t in targets]
feats = model.backbone(images)
model.heads.predictor.sparse_training_heads = False
pred = model.heads.predictor(feats, targets)
model.heads.predictor.sparse_training_heads = True


def rel(a, ref):
    return np.abs(a - ref).max() / (np.abs(ref).max() + 1e-300)


for key, t in (("feat_slice", feats[:, :4, ::6, ::16]), ("cls_slice", pred["cls"][:, :, ::4, ::8]),
               ("reg_slice", pred["reg"][:, ::25, ::6, ::16])):
    a = t.detach().cpu().numpy()
    print("%-12s ours-f64 %.2e   ours-ref32 %.2e   ref32-f64 %.2e" % (key, rel(a, g64[key]), rel(a, g32[key]), rel(g32[key], g64[key])))
gi.name_hashed_init(model)
model.zero_grad()
loss_dict, log = model(images, targets)
sum(loss_dict.values()).backward()
for k in H.LOSS_KEYS:
    a, r64, r32 = float(loss_dict[k]), float(g64["loss_" + k]), float(g32["loss_" + k])
    d = max(abs(r64), 1e-3)
    print("%-22s ours-f64 %.2e   ours-ref32 %.2e   ref32-f64 %.2e" % (k, abs(a - r64) / d, abs(a - r32) / d, abs(r32 - r64) / d))
n64 = dict(zip(g64["param_names"], g64["grad_norms"]))
n32 = dict(zip(g32["param_names"], g32["grad_norms"]))
floor = 1e-6 * max(n64.values())
rows = []
for n, p in model.named_parameters():
    got = 0.0 if p.grad is None else float(p.grad.double().norm())
    rows.append((abs(got - n64[n]) / max(n64[n], floor), abs(n32[n] - n64[n]) / max(n64[n], floor), n, n64[n]))
rows.sort(reverse=True)
print("gradient norms: worst ours-f64 %.2e (ref32-f64 worst %.2e), median ours %.2e ref32 %.2e" % (
    rows[0][0], max(r[1] for r in rows), float(np.median([r[0] for r in rows])), float(np.median([r[1] for r in rows]))))
for r in rows[:8]:
    print("   %.2e (ref32 %.2e)  %-55s |g| = %.3e" % r)
big = [r for r in rows if r[3] > 1e-3 * max(n64.values())]
print("parameters with |g| > 1e-3 max: worst ours-f64 %.2e, ref32-f64 %.2e" % (max(r[0] for r in big), max(r[1] for r in big)))
