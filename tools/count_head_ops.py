"""ATen ops of the prediction head (forward, and backward down to the backbone's feature map) by issuing function; same
counter as tools/count_loss_ops.py.  Own HIP kernels are autograd Functions and show up as the few ATen ops around them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import torch
from tools.count_loss_ops import Counter


def main():
    import bench
    from dcd_amd.structures.image_list import to_image_list
    args = argparse.Namespace(batch=8, objects=6, precision="f32", scaling="weak", amp=False)
    dev = torch.device("cuda:0")
    cfg, model, optimizer, images, targets = bench.build_everything(args, dev, 1, 0)[:5]
    with torch.no_grad():
        feats = model.backbone(to_image_list(images).tensors)
    feats = feats.detach().requires_grad_(True) if torch.is_tensor(feats) else [f.detach().requires_grad_(True) for f in feats]
    pred = model.heads.predictor
    for _ in range(2):
        out = pred(feats, targets)
    with Counter() as c:
        out = pred(feats, targets)
    top = int(os.environ.get("TOP", "40"))
    print("forward: %d ops" % sum(c.by_fn.values()))
    for k, v in c.by_fn.most_common(top):
        print("  %4d  %s" % (v, k))
    total = sum(v.float().sum() for v in out.values() if torch.is_tensor(v) and v.requires_grad)
    with Counter() as c2:
        total.backward()
    print("backward: %d ops" % sum(c2.by_op.values()))
    for k, v in c2.by_op.most_common(25):
        print("  %4d  %s" % (v, k))


if __name__ == "__main__":
    main()
