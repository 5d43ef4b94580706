"""First module whose OUTPUT differs bit for bit between two forward passes from the same weights and inputs (96x320 test model, train
mode, after one settling pass).   python tools/probes/first_divergence.py [f32|bf16x3|bf16] [own]
(eager processes run the stock solver on the stride-2 layers of the small maps: that solver is not reproducible; `own` = what a
graph process runs)"""
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
prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
if len(sys.argv) > 2 and sys.argv[2] == "own":          # the stride-2 layers as a graph process runs them (no stock solver)
    from dcd_amd import ops
    ops.stride2_on_own_kernels()
store = None


def hook(name):
    def f(mod, inp, out):
        if store is not None and torch.is_tensor(out):
            store.append((name, type(mod).__name__, out.detach().clone()))
    return f


for n, m in model.named_modules():
    if not list(m.children()):
        m.register_forward_hook(hook(n))


def run(backward):
    gi.name_hashed_init(model)
    model.zero_grad(set_to_none=True)
    with _ext.precision_scope(prec):
        ld, _ = model(images, targets)
    if backward:
        sum(ld[k] for k in H.LOSS_KEYS).backward()


for backward in (False, True):
    run(backward)
    store = []
    run(backward)
    a, store = store, []
    run(backward)
    b, store = store, None
    bad = [(i, n, t) for i, ((n, t, x), (_, _, y)) in enumerate(zip(a, b)) if not torch.equal(x, y)]
    print("%s, backward between the passes: %s -- %d of %d module outputs differ" % (prec, backward, len(bad), len(a)))
    for i, n, t in bad[:8]:
        x, y = a[i][2], b[i][2]
        print("   #%3d %-55s %-22s max |d| %.3e of %.3e" % (i, n, t, (x - y).abs().max().item(), x.abs().max().item()))
