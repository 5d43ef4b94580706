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
        recs = 0
        td = 0.0
        for i in range(16):
            one = {k: (v[i:i + 1] if torch.is_tensor(v) else v) for k, v in preds.items()}
            result, _, vis = model.heads.post_processor(one, targets[i:i + 1], test=model.test, features=feats[i:i + 1])
            ta = T()
            recs += len(infer_records(result, vis))
            td += T() - ta
        t3 = T()
    acc[0] += t1 - t0; acc[1] += t2 - t1; acc[2] += t3 - t2 - td; acc[3] += td
print("train half %.1f ms | eval backbone + predictor %.1f | decode 16 x %.2f ms | records 16 x %.2f ms" % (
    acc[0] / N * 1e3, acc[1] / N * 1e3, acc[2] / N * 1e3 / 16, acc[3] / N * 1e3 / 16))
