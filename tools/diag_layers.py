"""Layer-by-layer comparison of the GPU model (HIP + MIOpen) against the CPU model (oracle + PyTorch CPU)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import golden_inputs as gi
from dcd_amd.config import get_cfg
from dcd_amd.model.detector import KeypointDetector
from dcd_amd.model.backbone.DCNv2 import dcn_v2
from oracle import dcn_oracle
from dcd_amd import _ext


class Switch:
    """routes to the oracle for CPU tensors and to the HIP library for GPU tensors (diagnostic only)"""
    @staticmethod
    def dcn_v2_forward(x, *a, **k):
        return (_ext if x.is_cuda else dcn_oracle).dcn_v2_forward(x, *a, **k)
    @staticmethod
    def dcn_v2_backward(x, *a, **k):
        return (_ext if x.is_cuda else dcn_oracle).dcn_v2_backward(x, *a, **k)


dcn_v2._backend = Switch
cfg = get_cfg(opts=["MODEL.PRETRAIN", False, "MODEL.USE_SYNC_BN", False, "INPUT.WIDTH_TRAIN", 320, "INPUT.HEIGHT_TRAIN", 96])
torch.backends.cudnn.benchmark = False
mc = KeypointDetector(cfg); gi.name_hashed_init(mc); mc.train()
mg = KeypointDetector(cfg); gi.name_hashed_init(mg); mg.train().cuda()
images, _ = gi.model_inputs()
acts_c, acts_g = {}, {}
def hook(store):
    def mk(name):
        def f(m, i, o):
            if isinstance(o, torch.Tensor): store[name] = o.detach().float().cpu()
        return f
    return mk
for (n, m) in mc.backbone.named_modules():
    if len(list(m.children())) == 0: m.register_forward_hook(hook(acts_c)(n))
for (n, m) in mg.backbone.named_modules():
    if len(list(m.children())) == 0: m.register_forward_hook(hook(acts_g)(n))
with torch.no_grad():
    fc = mc.backbone(images); fg = mg.backbone(images.cuda())
worst = []
for n in acts_c:
    a, b = acts_c[n], acts_g[n]
    e = (a - b).abs().max().item() / (a.abs().max().item() + 1e-12)
    worst.append((e, n, tuple(a.shape)))
for e, n, s in worst:
    if e > 2e-4: print("%.2e %-50s %s" % (e, n, s))
print("final", (fc - fg.cpu()).abs().max().item() / fc.abs().max().item())
