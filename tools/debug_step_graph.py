import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dcd_amd.config import get_cfg
from dcd_amd.data.synthetic import make_batch
from dcd_amd.engine.trainer import GraphedTrainStep, build_optimizer, init_like_trained, train_step
from dcd_amd.model.detector import KeypointDetector
cuda = torch.device("cuda:0")
W, H, B = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (320, 96, 2)
cfg = get_cfg(opts=["MODEL.PRETRAIN", False, "MODEL.DEVICE", str(cuda), "MODEL.USE_SYNC_BN", False, "INPUT.WIDTH_TRAIN", W, "INPUT.HEIGHT_TRAIN", H])
batches = [make_batch(B, seed=s, n_objects=n, input_size=(W, H), device=cuda) for s, n in ((3, 3), (4, 5), (7, 2))]
mode = sys.argv[4] if len(sys.argv) > 4 else "all"
def run(kind):
    torch.manual_seed(0)
    model = KeypointDetector(cfg).to(cuda).train()
    init_like_trained(model)
    opt = build_optimizer(model, cfg)
    if kind == "eager_noloss_graph":
        model.heads.loss_evaluator.use_graph = False
    step = GraphedTrainStep(model, opt, cfg.SOLVER.GRAD_NORM_CLIP) if kind.startswith("graph") else None
    out = []
    for images, targets in batches:
        ld, _ = step(images, targets) if step else train_step(model, opt, images, targets, cfg.SOLVER.GRAD_NORM_CLIP)
        torch.cuda.synchronize()
        out.append([float(v) for v in ld.values()])
    w = torch.cat([p.detach().flatten() for p in model.parameters()])
    return out, w
res = {}
for kind in ("eager", "eager_noloss_graph", "graph", "graph2"):
    res[kind] = run(kind)
    print(kind, [["%.6f" % v for v in step[:3]] for step in res[kind][0]], flush=True)
for a in res:
    for b in res:
        if a < b:
            dl = max(abs(x - y) / max(abs(x), 1e-2) for sa, sb in zip(res[a][0], res[b][0]) for x, y in zip(sa, sb))
            dw = float((res[a][1] - res[b][1]).abs().max() / res[a][1].abs().max())
            print("%s vs %s: loss rel %.2e  weights rel %.2e" % (a, b, dl, dw))
