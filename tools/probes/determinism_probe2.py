"""Which parameter gradients differ between identical runs (fp32 / amp), in module order; run 1 vs 2 (both after the first call)."""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench

dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "f32"
args = argparse.Namespace(batch=8, objects=6, precision="f32" if mode == "amp" else mode, scaling="weak", amp=mode == "amp")
cfg, model, optimizer, images, targets = bench.build_everything(args, dev, 1, 0)[:5]
runs = []
for r in range(4):
    model.zero_grad(set_to_none=True)
    loss_dict, _ = model(images, targets)
    total = getattr(loss_dict, "total", None)
    total = total if total is not None else sum(loss_dict.values())
    total.backward()
    torch.cuda.synchronize()
    runs.append({n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
for n in runs[0]:
    if n.endswith("conv.bias"):
        continue
    g = runs[1][n]
    s = max(g.abs().max().item(), 1e-12)
    d12 = (runs[2][n] - g).abs().max().item() / s
    d13 = (runs[3][n] - g).abs().max().item() / s
    d01 = (runs[0][n] - g).abs().max().item() / s
    if max(d12, d13, d01) > 1e-4:
        print("%-66s 0v1 %.1e  1v2 %.1e  1v3 %.1e  |g| %.2e" % (n, d01, d12, d13, s), flush=True)
