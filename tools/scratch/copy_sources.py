"""Which Python lines issue the copy / add / fill kernels of a train step (torch.profiler with stacks), bs 8."""
import argparse, collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from dcd_amd.engine import trainer
from torch.profiler import profile, ProfilerActivity

args = argparse.Namespace(batch=int(os.environ.get("B", "8")), objects=6, precision="f32", scaling="weak", amp=False)
dev = torch.device("cuda:0")
cfg, model, optimizer, images, targets = bench.build_everything(args, dev, 1, 0)[:5]
os.environ["DCD_LOSS_GRAPH"] = "0"
clip = cfg.SOLVER.GRAD_NORM_CLIP
for _ in range(3):
    trainer.train_step(model, optimizer, images, targets, clip)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False, record_shapes=True) as prof:
    trainer.train_step(model, optimizer, images, targets, clip)
    torch.cuda.synchronize()
want = ("aten::copy_", "aten::add", "aten::add_", "aten::fill_", "aten::zero_", "aten::clone", "aten::contiguous", "aten::cat",
        "aten::mul", "aten::sum", "aten::_to_copy")
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.name.startswith("aten::") and ev.device_time_total > 0 and (os.environ.get("ALL") or ev.name in want):
        where = str([tuple(x) for x in (ev.input_shapes or []) if x][:3])
        k = (ev.name, where[:90])
        agg[k][0] += 1
        agg[k][1] += ev.device_time_total
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows)
print("selected ops: %.2f ms device time, %d calls" % (tot / 1e3, sum(v[0] for _, v in rows)))
for (name, where), (n, us) in rows[:int(os.environ.get('TOPN', '60'))]:
    print("%8.1f us %4d  %-16s %s" % (us, n, name, where))
