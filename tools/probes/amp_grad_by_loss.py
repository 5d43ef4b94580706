"""Which loss terms make the mixed-precision gradients of the 96x320 test model decorrelate from the fp32 ones (round 6).  For every
one of the 13 losses ALONE: cosine and norm ratio between the backbone gradient of the bf16-scope run and of the fp32 run (same
weights, same inputs), and the loss values.   python tools/probes/amp_grad_by_loss.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import golden_inputs as gi
import test_host_golden as H
from dcd_amd import _ext
from dcd_amd.model.detector import KeypointDetector

dev = torch.device("cuda:0")
model = KeypointDetector(H.small_cfg(str(dev))).to(dev)
gi.name_hashed_init(model)
model.train()
images, targets = gi.model_inputs()
images = images.to(dev)
targets = [t.to(dev) for t in targets]
params = [p for n, p in model.named_parameters() if n.startswith("backbone.") and p.numel() >= 4096]


def grads(prec, keys):
    gi.name_hashed_init(model)
    model.zero_grad()
    with _ext.precision_scope(prec):
        ld, _ = model(images, targets)
    vals = {k: float(ld[k]) for k in ld}
    sum(ld[k] for k in keys).backward()
    return torch.cat([p.grad.flatten().double() if p.grad is not None else torch.zeros(p.numel(), dtype=torch.float64, device=dev)
                      for p in params]), vals


for keys in [[k] for k in H.LOSS_KEYS] + [H.LOSS_KEYS]:
    a, va = grads("f32", keys)
    a2, _ = grads("f32", keys)
    b, vb = grads("bf16", keys)
    cos = lambda x, y: float(torch.dot(x, y) / (x.norm() * y.norm() + 1e-300))
    print("%-24s fp32 loss %10.4f bf16 %10.4f | grad norm fp32 %.3e bf16/fp32 %.3f cos %.4f | fp32 run-to-run cos %.6f" % (
        "+".join(keys) if len(keys) == 1 else "ALL", sum(va[k] for k in keys), sum(vb[k] for k in keys), float(a.norm()),
        float(b.norm() / (a.norm() + 1e-300)), cos(a, b), cos(a, a2)), flush=True)
