"""Which tensors of one training step differ bit for bit between two runs from the same weights and inputs (96x320 test model):
per precision scope, the loss terms and every parameter gradient, in the order of the parameters.  After one settling run (the first call of a DCN layer takes another
kernel sequence than the later ones: launch policy), a difference means a sum whose order is not fixed (atomics) upstream of it.   python tools/probes/run_to_run.py [f32|bf16|bf16x3 ...]"""
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
model.train()
images, targets = gi.model_inputs()
images = images.to(dev)
targets = [t.to(dev) for t in targets]


def run(prec):
    gi.name_hashed_init(model)
    model.zero_grad(set_to_none=True)
    with _ext.precision_scope(prec):
        ld, _ = model(images, targets)
    sum(ld[k] for k in H.LOSS_KEYS).backward()
    return {k: ld[k].detach().clone() for k in H.LOSS_KEYS}, {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


for prec in (sys.argv[1:] or ["f32", "bf16x3", "bf16"]):
    run(prec)                      # the first call of a layer also settles its DCN launch policy (another kernel sequence)
    la, ga = run(prec)
    lb, gb = run(prec)
    bad_l = [k for k in la if not torch.equal(la[k], lb[k])]
    bad_g = [n for n in ga if not torch.equal(ga[n], gb[n])]
    print("%-7s losses that differ: %s" % (prec, bad_l or "none"))
    print("%-7s gradients that differ: %d of %d" % (prec, len(bad_g), len(ga)))
    for n in bad_g[:6] + (["..."] if len(bad_g) > 12 else []) + bad_g[-6:]:
        if n == "...":
            print("        ...")
            continue
        d = (ga[n] - gb[n]).abs().max().item() / (ga[n].abs().max().item() + 1e-30)
        print("        %-60s max |d| / max |g| = %.2e" % (n, d))
