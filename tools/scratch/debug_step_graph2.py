import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dcd_amd.config import get_cfg
from dcd_amd.data.synthetic import make_batch
from dcd_amd.engine.trainer import GraphedTrainStep, build_optimizer, init_like_trained, train_step
from dcd_amd.model.detector import KeypointDetector
cuda = torch.device("cuda:0")
FULL = len(sys.argv) > 1 and sys.argv[1] == "full"
cfg = get_cfg(opts=["MODEL.PRETRAIN", False, "MODEL.DEVICE", str(cuda), "MODEL.USE_SYNC_BN", False] + ([] if FULL else ["INPUT.WIDTH_TRAIN", 320, "INPUT.HEIGHT_TRAIN", 96]))
images, targets = make_batch(8, seed=100, n_objects=6, device=cuda) if FULL else make_batch(2, seed=3, n_objects=3, input_size=(320, 96), device=cuda)
def run(kind):
    torch.manual_seed(0)
    model = KeypointDetector(cfg).to(cuda).train()
    init_like_trained(model)
    opt = build_optimizer(model, cfg)
    for g_ in opt.param_groups: g_["lr"].fill_(0.0)
    if "noloss" in kind: model.heads.loss_evaluator.use_graph = False
    clip = 0.0 if "noclip" in kind else cfg.SOLVER.GRAD_NORM_CLIP
    step = GraphedTrainStep(model, opt, clip) if kind.startswith("graph") else None
    ld, _ = step(images, targets) if step else train_step(model, opt, images, targets, clip)
    torch.cuda.synchronize()
    return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}, {k: float(v.detach()) for k, v in ld.items()}
res = {k: run(k) for k in ("eager", "eager2", "eager_noloss", "eager_noloss2")}
for k in res: print(k, {n: round(v, 4) for n, v in list(res[k][1].items())[:4]}, "gnorm %.4f" % float(torch.cat([g.flatten() for g in res[k][0].values()]).norm()))
def cmp(a, b):
    rows = []
    for n in res[a][0]:
        x, y = res[a][0][n], res[b][0][n]
        rows.append((float((x - y).abs().max() / (x.abs().max() + 1e-20)), n, float(x.abs().max())))
    gmax = max(r[2] for r in rows)
    rows = [r for r in rows if r[2] > 1e-4 * gmax]
    rows.sort(reverse=True)
    print(a, "vs", b, " ".join("%s:%.1e(|g|%.1e)" % (n[-40:], e, m) for e, n, m in rows[:4]))
cmp("eager", "eager2"); cmp("eager", "eager_noloss"); cmp("eager_noloss", "eager_noloss2")
