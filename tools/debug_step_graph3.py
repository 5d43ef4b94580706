import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dcd_amd.config import get_cfg
from dcd_amd.data.synthetic import make_batch
from dcd_amd.engine.trainer import GraphedTrainStep, build_optimizer, init_like_trained, train_step
from dcd_amd.model.detector import KeypointDetector
cuda = torch.device("cuda:0")
cfg = get_cfg(opts=["MODEL.PRETRAIN", False, "MODEL.DEVICE", str(cuda), "MODEL.USE_SYNC_BN", False])
images, targets = make_batch(8, seed=100, n_objects=6, device=cuda)
for kind in sys.argv[1:]:
    torch.manual_seed(0)
    model = KeypointDetector(cfg)
    init_like_trained(model, std=0.01, seed=0)
    model = model.to(cuda).train()
    opt = build_optimizer(model, cfg)
    step = GraphedTrainStep(model, opt, cfg.SOLVER.GRAD_NORM_CLIP) if kind == "graph" else None
    if kind == "eager_noloss":
        model.heads.loss_evaluator.use_graph = False
    for it in range(14):
        ld, _ = step(images, targets) if step else train_step(model, opt, images, targets, cfg.SOLVER.GRAD_NORM_CLIP)
        torch.cuda.synchronize()
        tot = sum(float(v.detach()) for v in ld.values())
        offmax = max(float(m.conv_offset_mask.weight.detach().abs().max()) for m in model.modules() if hasattr(m, "conv_offset_mask"))
        wmax = max(float(p.detach().abs().max()) for p in model.parameters())
        print(kind, it, "loss %.4f  max|w_offset| %.4f max|w| %.3f step %s" % (tot, offmax, wmax, float(next(iter(opt.state.values()))["step"])), flush=True)
