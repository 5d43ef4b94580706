"""amp: where in the backward do two outcomes appear?  Gradients of module outputs (decoder's last stage and the backbone output)
over four identical runs."""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench

dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "amp"
args = argparse.Namespace(batch=8, objects=6, precision="f32" if mode == "amp" else mode, scaling="weak", amp=mode == "amp")
cfg, model, optimizer, images, targets = bench.build_everything(args, dev, 1, 0)[:5]
store = {}
cur = {}

def watch(mod, name):
    orig = mod.forward
    def fwd(*a, **k):
        out = orig(*a, **k)
        t = out if torch.is_tensor(out) else None
        if t is not None and t.requires_grad:
            t.register_hook(lambda g, n=name: cur.__setitem__(n + ".gout", g.detach().clone()))
        for i, x in enumerate(a):
            if torch.is_tensor(x) and x.requires_grad and i == 0:
                x.register_hook(lambda g, n=name: cur.__setitem__(n + ".gin", g.detach().clone()))
        return out
    mod.forward = fwd

names = ["backbone.ida_up.node_2", "backbone.ida_up.node_2.conv", "backbone.ida_up.node_2.conv.conv_offset_mask", "backbone.ida_up.node_2.actf",
         "backbone.ida_up.proj_2", "backbone.ida_up.up_2", "backbone.dla_up.ida_2.node_3", "heads.predictor.class_head", "backbone"]
mods = dict(model.named_modules())
for n in names:
    if n in mods:
        watch(mods[n], n)
runs = []
for r in range(int(os.environ.get('RUNS', '4'))):
    cur.clear()
    model.zero_grad(set_to_none=True)
    loss_dict, _ = model(images, targets)
    total = getattr(loss_dict, "total", None)
    total = total if total is not None else sum(loss_dict.values())
    total.backward()
    torch.cuda.synchronize()
    runs.append(dict(cur))
for n in sorted(runs[0]):
    g = runs[1][n]
    s = max(g.abs().max().item(), 1e-20)
    print("%-60s %s |g| %.2e" % (n, " ".join("%.0e" % ((runs[k][n] - g).abs().max().item() / s) for k in range(len(runs))), s), flush=True)

# pattern of the differing elements of the last decoder layer's input gradient
n = "backbone.ida_up.node_2.conv.gin"
import itertools
for a, b in [(0, k) for k in range(1, len(runs))]:
    d = (runs[a][n] - runs[b][n]).abs()
    s = runs[a][n].abs().max().item()
    idx = (d > 2e-6 * s).nonzero()
    if idx.numel() == 0:
        print("pattern %d v %d: none" % (a, b)); continue
    print("pattern %d v %d: %d elements differ (of %d), max %.2e;  b %s  c [%d..%d] n=%d  y [%d..%d] n=%d  x [%d..%d] n=%d" % (
        a, b, idx.shape[0], d.numel(), d.max().item() / s, sorted(set(idx[:, 0].tolist())), idx[:, 1].min(), idx[:, 1].max(), idx[:, 1].unique().numel(),
        idx[:, 2].min(), idx[:, 2].max(), idx[:, 2].unique().numel(), idx[:, 3].min(), idx[:, 3].max(), idx[:, 3].unique().numel()), flush=True)
    ys = idx[:, 2].unique().tolist(); xs = idx[:, 3].unique().tolist()
    print("    rows", ys[:40], "cols", xs[:48])
