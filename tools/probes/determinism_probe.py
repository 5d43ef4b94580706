"""Run-to-run spread of the gradients of one bs-8 train forward+backward from identical weights and inputs, per precision:
max |g_run1 - g_run2| / max |g| per parameter, worst and median (fp32 atomics reorder sums; anything far above that is a race)."""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench

dev = torch.device("cuda:0")
for mode in ("f32", "bf16x3", "amp"):
    args = argparse.Namespace(batch=8, objects=6, precision="f32" if mode == "amp" else mode, scaling="weak", amp=mode == "amp")
    cfg, model, optimizer, images, targets = bench.build_everything(args, dev, 1, 0)[:5]
    runs = []
    for r in range(3):
        model.zero_grad(set_to_none=True)
        loss_dict, _ = model(images, targets)
        total = getattr(loss_dict, "total", None)
        total = total if total is not None else sum(loss_dict.values())
        total.backward()
        torch.cuda.synchronize()
        runs.append((float(total), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
    base = runs[0][1]
    for r in (1, 2):
        rel = []
        for n, g in base.items():
            d = (runs[r][1][n] - g).abs().max().item()
            rel.append((d / max(g.abs().max().item(), 1e-12), n))
        rel.sort(reverse=True)
        med = rel[len(rel) // 2][0]
        if r == 1:
            keep = [(v, n) for v, n in rel if not n.endswith("conv.bias")]
            for v, n in keep[:14]:
                print("      %.3e  %-70s |g|max %.3e" % (v, n, base[n].abs().max().item()))
            for v, n in rel:
                if n in ("backbone.base.base_layer.0.weight", "backbone.base.level2.tree1.conv1.weight", "heads.predictor.class_head.0.weight",
                         "backbone.dla_up.ida_0.proj_1.conv.weight", "backbone.dla_up.ida_0.proj_1.conv.conv_offset_mask.weight", "backbone.base.level5.tree1.conv1.weight"):
                    print("      key  %.3e  %s" % (v, n))
        print("%-6s run %d vs 0: loss %.9f vs %.9f | worst %.3e (%s) second %.3e (%s) median %.3e" % (
            mode, r, runs[r][0], runs[0][0], rel[0][0], rel[0][1], rel[1][0], rel[1][1], med), flush=True)
    from dcd_amd import _ext
    _ext.set_precision("f32")
