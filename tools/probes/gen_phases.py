"""Wall time of the phases of the --generate_for_GMW pass at bs 16 (host clock with synchronisation: the pass has host syncs)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from dcd_amd.engine.gen_data import infer_records
dev = torch.device("cuda:0")
args = argparse.Namespace(batch=16, objects=6)
cfg, model, images, targets = bench._gen_build(args, dev)
for _ in range(2):
    bench._gen_pass(model, images, targets, torch)


def T():
    torch.cuda.synchronize()
    return time.perf_counter()


acc = [0.0] * 4
N = 4
for _ in range(N):
    lc = model.heads.loss_evaluator
    for k in lc.gen_data:
        lc.gen_data[k] = []
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.eval()
    with torch.no_grad():
        t0 = T()
        model(images, targets)
        t1 = T()
        model.eval()
        feats = model.backbone(images)
        preds = model.heads.predictor(feats, targets)
        t2 = T()
        from dcd_amd.engine.gen_data import infer_records_batch
        tb0 = T()
        feats2 = model.backbone(images)
        tb1 = T()
        preds2 = model.heads.predictor(feats2, targets)
        tb2 = T()
        rows, _, vis, image_of = model.heads.post_processor.forward_batch(preds, targets, test=model.test, features=feats)
        ta = T()
        recs = sum(len(r) for r in infer_records_batch(rows, vis, image_of, 16))
        td = T() - ta
        t3 = T()
        extra = (tb1 - tb0, tb2 - tb1)
        t3 -= (tb2 - tb0)
    acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2 - td; acc[3] += td
print("train half %.1f ms | eval backbone + predictor %.1f (backbone %.1f, predictor %.1f) | batched decode %.2f ms | records %.2f ms" % (
    acc[0] / N * 1e3, acc[1] / N * 1e3, extra[0] * 1e3, extra[1] * 1e3, acc[2] / N * 1e3, acc[3] / N * 1e3))
# train half split: backbone / predictor / loss-path decode
model.train()
for m in model.modules():
    if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
        m.eval()
with torch.no_grad():
    t0 = T(); f = model.backbone(images); t1 = T(); p = model.heads.predictor(f, targets); t2 = T()
    model.heads.loss_evaluator(p, targets); t3 = T()
print("train half: backbone %.1f | predictor %.1f | loss-path decode + generate_data %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
bench._gen_pass(model, images, targets, torch); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
