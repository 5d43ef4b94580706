"""Autograd nodes of one training step whose OUTPUT is consumed more than once (each extra consumer is an accumulation `add` launch in
the backward), with the producing node, its consumers and the tensor size (96x320 test model on the GPU).
python tools/probes/grad_fanin.py"""
import os, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import golden_inputs as gi
import test_host_golden as H
from dcd_amd.model.detector import KeypointDetector

dev = torch.device("cuda:0")
model = KeypointDetector(H.small_cfg(str(dev))).to(dev)
gi.name_hashed_init(model)
model.train()
images, targets = gi.model_inputs()
ld, _ = model(images.to(dev), [t.to(dev) for t in targets])
loss = sum(ld[k] for k in H.LOSS_KEYS)

consumers = defaultdict(list)          # (node, output index) -> consumer names
seen, stack = set(), [loss.grad_fn]
while stack:
    n = stack.pop()
    if n is None or n in seen:
        continue
    seen.add(n)
    for fn, idx in n.next_functions:
        if fn is not None:
            consumers[(fn, idx)].append(type(n).__name__)
            stack.append(fn)
rows = []
for (fn, idx), cs in consumers.items():
    if len(cs) > 1 and type(fn).__name__ != "AccumulateGrad":
        meta = getattr(fn, "_input_metadata", None)
        shape = None
        try:
            shape = tuple(fn._input_metadata[idx].shape)
        except Exception:
            pass
        rows.append((shape, type(fn).__name__, idx, cs))
rows.sort(key=lambda r: -(torch.Size(r[0]).numel() if r[0] else 0))
print("%d outputs with more than one consumer, %d extra accumulations" % (len(rows), sum(len(r[3]) - 1 for r in rows)))
for shape, name, idx, cs in rows:
    print("%-22s %-34s out %d <- %s" % (shape, name, idx, ", ".join(cs)))
