import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dcd_amd.config import get_cfg
from dcd_amd.data.synthetic import make_batch
from dcd_amd.model.head.detector_loss import Loss_Computation
cuda = torch.device("cuda:0")
B, nobj = int(sys.argv[1]), int(sys.argv[2])
W, H = (1280, 384) if len(sys.argv) < 4 else (int(sys.argv[3]), int(sys.argv[4]))
cfg = get_cfg(opts=["MODEL.PRETRAIN", False, "MODEL.DEVICE", str(cuda), "INPUT.WIDTH_TRAIN", W, "INPUT.HEIGHT_TRAIN", H])
lc = Loss_Computation(cfg)
M, C = cfg.DATASETS.MAX_OBJECTS, 415
_, targets = make_batch(B, seed=100, n_objects=nobj, input_size=(W, H), device=cuda)
g = torch.Generator().manual_seed(0)
for it in range(6):
    cls = torch.sigmoid(torch.randn(B, 1, H // 4, W // 4, generator=g) - 2).clamp(1e-4, 1 - 1e-4).to(cuda)
    pois = (torch.randn(B, M, C, generator=g) * 0.3).to(cuda)
    res = []
    for use_graph in (False, True):
        lc.use_graph = use_graph
        c, p = cls.clone().requires_grad_(), pois.clone().requires_grad_()
        ld, log = lc({'cls': c, 'reg': None, 'reg_pois': p}, targets)
        tot = ld.total if getattr(ld, "total", None) is not None else sum(ld.values())
        tot.backward()
        torch.cuda.synchronize()
        res.append((float(tot), c.grad.clone(), p.grad.clone(), {k: float(v) for k, v in ld.items()}))
    (t0, gc0, gp0, l0), (t1, gc1, gp1, l1) = res
    bad = {k: (l0[k], l1[k]) for k in l0 if abs(l0[k] - l1[k]) > 1e-4 * max(abs(l0[k]), 1e-3)}
    print(it, "eager %.5f graph %.5f  dcls %.1e dpois %.1e" % (t0, t1, float((gc0 - gc1).abs().max() / gc0.abs().max()), float((gp0 - gp1).abs().max() / gp0.abs().max())), bad, flush=True)
