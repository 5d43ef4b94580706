"""Gradients of one bs-8 forward+backward from the bench's weights and batch, in one mode, saved to a file (argv: mode out).
mode: f32 | amp.  Toggle DCD_CONV_BF16_DIRECT / DCD_CONV_WRW_DIRECT in the environment; compare files with amp_grad_cmp.py."""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench

dev = torch.device("cuda:0")
mode, out = sys.argv[1], sys.argv[2]
args = argparse.Namespace(batch=int(os.environ.get("DUMP_BATCH", "8")), objects=6, precision="f32", scaling="weak", amp=mode == "amp")
cfg, model, optimizer, images, targets = bench.build_everything(args, dev, 1, 0)[:5]
model.zero_grad(set_to_none=True)
loss_dict, _ = model(images, targets)
total = getattr(loss_dict, "total", None)
total = total if total is not None else sum(loss_dict.values())
total.backward()
torch.cuda.synchronize()
torch.save({"loss": float(total), "grads": {n: p.grad.detach().cpu() for n, p in model.named_parameters() if p.grad is not None}}, out)
print(mode, "loss", float(total))
