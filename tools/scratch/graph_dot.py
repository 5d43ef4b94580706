"""Can the captured step's node types be listed?  CUDAGraph.enable_debug_mode() + debug_dump() -> DOT text; count node kinds."""
import argparse, os, re, sys, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from dcd_amd.engine import trainer

dev = torch.device("cuda:0")
orig = torch.cuda.CUDAGraph
made = []
class DebugGraph(orig):
    def __new__(cls, *a, **k):
        g = super().__new__(cls, *a, **k)
        return g
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.enable_debug_mode()
        made.append(self)
torch.cuda.CUDAGraph = DebugGraph
args = argparse.Namespace(batch=int(os.environ.get("B", "8")), objects=6, precision=os.environ.get("PREC", "f32"), scaling="weak", amp=False)
cfg, model, optimizer, images, targets = bench.build_everything(args, dev, 1, 0)[:5]
step = trainer.GraphedTrainStep(model, optimizer, cfg.SOLVER.GRAD_NORM_CLIP)
step(images, targets)
torch.cuda.synchronize()
print("graphs made:", len(made))
for k, g in enumerate(made):
    path = "/tmp/step_graph_%d.dot" % k
    try:
        g.debug_dump(path)
    except Exception as e:
        print("debug_dump failed:", e); continue
    txt = open(path).read()
    print("dot bytes", len(txt))
    kinds = collections.Counter(re.findall(r'label="[^"]*?(MEMSET|MEMCPY|KERNEL|memset|memcpy|kernel|Memset|Memcpy|Kernel)', txt))
    print(kinds)
    ms = [l for l in txt.splitlines() if re.search("emset", l)]
    print(len(ms), "lines mention memset;", ms[:3])
