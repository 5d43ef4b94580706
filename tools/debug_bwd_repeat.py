import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dcd_amd.config import get_cfg
from dcd_amd.data.synthetic import make_batch
from dcd_amd.engine.trainer import init_like_trained
from dcd_amd.model.detector import KeypointDetector
cuda = torch.device("cuda:0")
cfg = get_cfg(opts=["MODEL.PRETRAIN", False, "MODEL.DEVICE", str(cuda), "MODEL.USE_SYNC_BN", False, "INPUT.WIDTH_TRAIN", 320, "INPUT.HEIGHT_TRAIN", 96])
images, targets = make_batch(2, seed=3, n_objects=3, input_size=(320, 96), device=cuda)
torch.manual_seed(0)
model = KeypointDetector(cfg).to(cuda).train()
init_like_trained(model)
model.heads.loss_evaluator.use_graph = False
state = {k: v.clone() for k, v in model.state_dict().items()}
order, store = [], {}
def fmk(name):
    def hook(mod, i, o):
        if isinstance(o, torch.Tensor):
            acts.setdefault(name, []).append(o.detach().clone())
            if o.requires_grad:
                def gh(g, name=name):
                    if name not in store: order.append(name)
                    store.setdefault(name, []).append([g.detach().clone()])
                o.register_hook(gh)
    return hook
acts = {}
for n, m in model.backbone.named_modules():
    if len(list(m.children())) == 0:
        m.register_forward_hook(fmk(n))
for rep in range(3):
    model.load_state_dict(state)
    model.zero_grad(set_to_none=True)
    junk = torch.randn(1 << (18 + rep), device=cuda) * 1e6; del junk
    ld, _ = model(images, targets)
    sum(ld.values()).backward()
torch.cuda.synchronize()
print("forward activations that differ between repetitions:")
for n, v in acts.items():
    d = max(float((v[0] - x).abs().max() / (v[0].abs().max() + 1e-20)) for x in v[1:])
    if d > 1e-5: print("  %-60s %.2e" % (n, d))
print("backward (in execution order): modules whose grad_output / grad_input differ")
for n in order:
    v = store[n]
    ds = []
    for k in range(len(v[0])):
        if v[0][k] is None: ds.append(0.0); continue
        ds.append(max(float((v[0][k] - r[k]).abs().max() / (v[0][k].abs().max() + 1e-20)) for r in v[1:]))
    if max(ds) > 1e-4: print("  %-60s %s" % (n, " ".join("%.1e" % d for d in ds)))
