"""Where does the GPU forward leave the exact result?  Layer by layer: the GPU model (HIP kernels + MIOpen, fp32) and the CPU
model in fp32 (oracle DCN + oneDNN) are both compared with the SAME model run in float64 on the host (f64 build of the
oracle) -- which reproduces the reference's float64 run bit for bit (tests/test_host_golden.py::test_float64_model_...).
Diagnostic only (tests/ and tools/ may use the oracle).  Usage: python tools/diag_layers_f64.py [threshold]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import golden_inputs as gi
from dcd_amd import _ext
from dcd_amd.config import get_cfg
from dcd_amd.model.backbone.DCNv2 import dcn_v2
from dcd_amd.model.detector import KeypointDetector
from oracle import dcn_oracle


class Switch:
    @staticmethod
    def dcn_v2_forward(x, *a, **k):
        return (_ext if x.is_cuda else dcn_oracle).dcn_v2_forward(x, *a, **k)

    @staticmethod
    def dcn_v2_backward(x, *a, **k):
        return (_ext if x.is_cuda else dcn_oracle).dcn_v2_backward(x, *a, **k)


dcn_v2._backend = Switch
thr = float(sys.argv[1]) if len(sys.argv) > 1 else 5e-5
cfg = get_cfg(opts=["MODEL.PRETRAIN", False, "MODEL.USE_SYNC_BN", False, "INPUT.WIDTH_TRAIN", 320, "INPUT.HEIGHT_TRAIN", 96])
torch.backends.cudnn.benchmark = False
models = {}
for name in ("f64", "cpu32", "gpu32"):
    m = KeypointDetector(cfg)
    gi.name_hashed_init(m)
    m.train()
    models[name] = m.double() if name == "f64" else m.cuda() if name == "gpu32" else m
images, _ = gi.model_inputs()
acts = {k: {} for k in models}


def hook(store, name):
    def f(m, i, o):
        if isinstance(o, torch.Tensor):
            store[name] = o.detach().double().cpu()
    return f


for k, m in models.items():
    for n, mod in m.backbone.named_modules():
        if len(list(mod.children())) == 0:
            mod.register_forward_hook(hook(acts[k], n))
with torch.no_grad():
    out = {"f64": models["f64"].backbone(images.double()), "cpu32": models["cpu32"].backbone(images),
           "gpu32": models["gpu32"].backbone(images.cuda())}


def dist(a, b):
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-300)


print("%-52s %-22s %10s %10s" % ("layer", "shape", "cpu32", "gpu32"))
for n, ref in acts["f64"].items():
    ec, eg = dist(acts["cpu32"][n], ref), dist(acts["gpu32"][n], ref)
    if max(ec, eg) > thr:
        print("%-52s %-22s %10.2e %10.2e" % (n, tuple(ref.shape), ec, eg))
print("final: cpu32 %.2e  gpu32 %.2e" % (dist(out["cpu32"].double(), out["f64"]), dist(out["gpu32"].double().cpu(), out["f64"])))
