import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
dev = torch.device("cuda:0")
args = argparse.Namespace(batch=2, objects=6)
cfg, model, images, targets = bench._gen_build(args, dev)
model.eval()
S = torch.cuda.synchronize
with torch.no_grad():
    for b in (2, 1):
        im, tg = images[:b], targets[:b]
        feats = model.backbone(im); S(); print("backbone ok", b, flush=True)
        p = model.heads.predictor
        from dcd_amd.model.layers.utils import select_topk
        from dcd_amd.model.head import trunk_moments
        fc = p.class_head[:-1](feats); oc = p.class_head[-1](fc); S(); print("class head ok", flush=True)
        oc = p._edge_fusion_cls(fc, oc, tg); S(); print("edge cls ok", flush=True)
        from dcd_amd.model.layers.utils import sigmoid_hm
        heat = sigmoid_hm(oc).float(); topk = select_topk(heat, K=50, fuse_nms=True); S(); print("topk ok", [t.shape for t in topk], topk[1].min().item(), topk[1].max().item(), flush=True)
        preds = p(feats, tg); S(); print("predictor ok", preds['reg_pois'].shape, flush=True)
        out = model.heads.post_processor(preds, tg, test=model.test, features=feats) if b == 1 else model.heads.post_processor.forward_batch(preds, tg, test=model.test, features=feats)
        S(); print("post ok", out[0].shape, flush=True)
