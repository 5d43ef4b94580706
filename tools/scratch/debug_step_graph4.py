import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dcd_amd.config import get_cfg
from dcd_amd.data.synthetic import make_batch
from dcd_amd.engine.trainer import build_optimizer, init_like_trained, train_step
from dcd_amd.model.detector import KeypointDetector
cuda = torch.device("cuda:0")
cfg = get_cfg(opts=["MODEL.PRETRAIN", False, "MODEL.DEVICE", str(cuda), "MODEL.USE_SYNC_BN", False])
images, targets = make_batch(8, seed=100, n_objects=6, device=cuda)
torch.manual_seed(0)
model = KeypointDetector(cfg)
init_like_trained(model, std=0.01, seed=0)
model = model.to(cuda).train()
opt = build_optimizer(model, cfg)
lc = model.heads.loss_evaluator
seen = {}
orig = lc.__class__.__call__
def spy(self, predictions, tg):
    out = orig(self, predictions, tg)
    seen["pred"] = {k: (None if v is None else v.detach().clone()) for k, v in predictions.items()}
    return out
lc.__class__.__call__ = spy
for it in range(16):
    ld, _ = train_step(model, opt, images, targets, cfg.SOLVER.GRAD_NORM_CLIP)
    torch.cuda.synchronize()
    tot_now = sum(float(v.detach()) for v in ld.values())
    lc.use_graph = False
    with torch.no_grad():
        ld2, _ = orig(lc, seen["pred"], targets)
    lc.use_graph = True
    bad = {k: (round(float(ld[k].detach()), 4), round(float(ld2[k]), 4)) for k in ld2 if abs(float(ld[k].detach()) - float(ld2[k])) > 1e-3 * max(abs(float(ld2[k])), 1e-2)}
    print("   differing terms (graphed, eager):", bad)
    print(it, "graphed-loss step: read-after-step total %.4f ; eager re-evaluation on the same predictions %.4f ; ld.total %.4f" % (
        tot_now, sum(float(v) for v in ld2.values()), float(ld.total)), flush=True)
