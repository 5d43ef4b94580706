"""Hunt for the graph-vs-twin outlier of tests/test_gpu_golden.py::test_graphed_training_gradients_track_an_eager_twin: repeat the
test's loop until one tensor's gradient differs by more than 1e-3 of its range, then save the twin's weights (/tmp/twin_catch.pt),
and both gradients of every DCN offset convolution.  argv: attempts."""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from dcd_amd import _ext
from dcd_amd.engine import trainer

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10

def build(zero_lr):
    args = argparse.Namespace(batch=8, objects=6, precision="bf16x3", scaling="weak", amp=False)
    r = bench.build_everything(args, dev, 1, 0)[:5]
    if zero_lr:
        for g in r[2].param_groups:
            g["lr"].fill_(0.0)
            g["weight_decay"] = 0.0
    return r

for attempt in range(N):
    cfg, A, optA, images, targets = build(False)
    _, B, optB, _, _ = build(True)
    clip = cfg.SOLVER.GRAD_NORM_CLIP
    step = trainer.GraphedTrainStep(A, optA, clip)
    worst = (0.0, None, -1)
    for it in range(41):
        check = it > 0 and it % 5 == 0
        if check:
            torch.cuda.synchronize()
            with torch.no_grad():
                for p, q in zip(A.parameters(), B.parameters()):
                    q.copy_(p)
                for p, q in zip(A.buffers(), B.buffers()):
                    q.copy_(p)
            pre = {k: v.detach().clone() for k, v in B.state_dict().items()}
            trainer.train_step(B, optB, images, targets, clip)
            torch.cuda.synchronize()
            gb = {n: p.grad.detach().clone() for n, p in B.named_parameters() if p.grad is not None}
        step(images, targets)
        if check:
            torch.cuda.synchronize()
            for n, p in A.named_parameters():
                if p.grad is None or n.endswith("conv.bias") or float(gb[n].abs().max()) < 1e-7:
                    continue
                rel = float((p.grad - gb[n]).abs().max() / gb[n].abs().max())
                if rel > worst[0]:
                    worst = (rel, n, it)
            if worst[0] > 1e-3:
                torch.save({"state": pre, "name": worst[1], "it": worst[2], "rel": worst[0],
                            "graph": {n: p.grad.detach().cpu() for n, p in A.named_parameters() if p.grad is not None},
                            "twin": {n: g.cpu() for n, g in gb.items()}}, "/tmp/twin_catch.pt")
                print("CAUGHT attempt %d: %s at replay %d differs by %.3e" % (attempt, worst[1], worst[2], worst[0]), flush=True)
                rels = sorted(((float((A.get_parameter(n).grad - g).abs().max() / g.abs().max()), n) for n, g in gb.items()
                               if A.get_parameter(n).grad is not None and float(g.abs().max()) >= 1e-7 and not n.endswith("conv.bias")), reverse=True)[:8]
                for r_, n_ in rels:
                    print("    %.3e  %s" % (r_, n_))
                sys.exit(0)
    print("attempt %d clean: worst %.3e (%s, replay %d)" % (attempt, worst[0], worst[1], worst[2]), flush=True)
    del step, A, B, optA, optB
    _ext.set_precision("f32")
    torch.cuda.empty_cache()
print("nothing caught")
sys.exit(1)
