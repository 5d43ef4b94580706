"""Which parameters move abnormally in graph mode?  |w_40 - w_0| per parameter (relative to lr * steps), graph vs eager."""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from dcd_amd.engine import trainer

N = int(os.environ.get("N", "45"))
dev = torch.device("cuda:0")
res = {}
for mode in ("eager", "graph"):
    args = argparse.Namespace(batch=8, objects=6, precision=os.environ.get("PREC", "f32"), scaling="weak", amp=False)
    cfg, model, optimizer, images, targets = bench.build_everything(args, dev, 1, 0)[:5]
    clip = cfg.SOLVER.GRAD_NORM_CLIP
    w0 = {n: p.detach().clone() for n, p in model.named_parameters()}
    step = trainer.GraphedTrainStep(model, optimizer, clip) if mode == "graph" else None
    for it in range(N):
        ld, _ = step(images, targets) if step else trainer.train_step(model, optimizer, images, targets, clip)
    torch.cuda.synchronize()
    loss = float(sum(float(v) for v in ld.values()))
    res[mode] = ({n: float((p.detach() - w0[n]).abs().mean()) / (3e-4 * N) for n, p in model.named_parameters()}, loss)
    del step, model, optimizer
    torch.cuda.empty_cache()
print("loss after %d steps: eager %.3f graph %.3f" % (N, res["eager"][1], res["graph"][1]))
rows = sorted(((res["graph"][0][n] / max(res["eager"][0][n], 1e-6), res["graph"][0][n], res["eager"][0][n], n) for n in res["eager"][0]), reverse=True)
print("mean |dw| / (lr steps): parameters that moved most differently (ratio graph / eager)")
for r, g, e, n in rows[:14]:
    print("  x%5.2f  graph %.3f eager %.3f  %s" % (r, g, e, n))
for r, g, e, n in rows[-6:]:
    print("  x%5.2f  graph %.3f eager %.3f  %s" % (r, g, e, n))
import statistics
print("median ratio %.2f" % statistics.median(r for r, _, _, _ in rows))
