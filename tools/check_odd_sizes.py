"""Train steps at input sizes whose deeper maps break the kernels' alignment rules (W % 4, H % 2, 32-pixel tiles): every fallback
path must run and give finite losses.   python tools/check_odd_sizes.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dcd_amd.config import get_cfg
from dcd_amd.data.synthetic import make_batch
from dcd_amd.engine.trainer import build_optimizer, init_like_trained, train_step
from dcd_amd.model.detector import KeypointDetector

dev = torch.device("cuda", 0)
for (w, h, b) in ((1248, 384, 2), (352, 96, 3), (1280, 384, 1), (672, 224, 2)):
    cfg = get_cfg(opts=["MODEL.PRETRAIN", False, "MODEL.DEVICE", "cuda:0", "INPUT.WIDTH_TRAIN", w, "INPUT.HEIGHT_TRAIN", h])
    torch.manual_seed(0)
    model = KeypointDetector(cfg).to(dev).train()
    init_like_trained(model)
    opt = build_optimizer(model, cfg)
    images, targets = make_batch(b, seed=3, n_objects=4, input_size=(w, h), image_size=(w - 10, h - 5), device=dev)
    vals = []
    for _ in range(3):
        ld, _ = train_step(model, opt, images, targets, cfg.SOLVER.GRAD_NORM_CLIP)
        total = getattr(ld, "total", None)
        vals.append(float(total if total is not None else sum(ld.values())))
    torch.cuda.synchronize()
    ok = all(v == v and abs(v) < 1e6 for v in vals) and all(bool(torch.isfinite(p).all()) for p in model.parameters())
    print("%4dx%-4d bs %d: losses %s  %s" % (w, h, b, ["%.3f" % v for v in vals], "ok" if ok else "NOT FINITE"))
    assert ok
