"""GMW train step: 120 steps on one synthetic batch with the extractor graphs (default) and without (DCD_GMW_GRAPH=0): the losses
must follow the same trajectory (deterministic-ish GEMM / Conv1d kernels: any gap is the graph)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from dcd_amd.gmw import GMW, gmw_train_step

dev = torch.device("cuda:0")
N = int(os.environ.get("N", "121"))
for mode in ("graph", "eager"):
    os.environ["DCD_GMW_GRAPH"] = "1" if mode == "graph" else "0"
    torch.manual_seed(0)
    model = GMW().to(dev).train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.999))
    batch = bench._gmw_inputs(8, 100, dev)
    hist = []
    for it in range(N):
        out = gmw_train_step(model, opt, *batch, 0.1, 1.0)
        if it % 20 == 0:
            vals = out if isinstance(out, (tuple, list)) else [out]
            nums = []
            for v in vals:
                if torch.is_tensor(v) and v.numel() == 1:
                    nums.append(float(v))
                elif isinstance(v, dict):
                    nums += [float(x) for x in v.values() if torch.is_tensor(x) and x.numel() == 1]
            hist.append((it, nums[:3]))
    print("%-6s %s" % (mode, "  ".join("%d:%s" % (i, "/".join("%.5f" % x for x in v)) for i, v in hist)), flush=True)
