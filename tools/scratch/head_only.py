"""Predictor forward + backward alone, N iterations (for a kernel trace of the head)."""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from dcd_amd.structures.image_list import to_image_list
args = argparse.Namespace(batch=int(os.environ.get("B", "8")), objects=6, precision="f32", scaling="weak", amp=False)
dev = torch.device("cuda:0")
cfg, model, optimizer, images, targets = bench.build_everything(args, dev, 1, 0)[:5]
with torch.no_grad():
    feats = model.backbone(to_image_list(images).tensors)
feats = feats.detach().requires_grad_(True)
pred = model.heads.predictor
N = int(os.environ.get("N", "6"))
for it in range(N):
    out = pred(feats, targets)
    total = out['reg_pois'].sum() + out['cls'].sum()
    total.backward()
torch.cuda.synchronize()
print("done", N)
