import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dcd_amd.config import get_cfg
from dcd_amd.data.synthetic import make_batch
from dcd_amd.engine.trainer import build_optimizer, init_like_trained, train_step
from dcd_amd.model.detector import KeypointDetector
from dcd_amd.model.head import detector_loss as DL
cuda = torch.device("cuda:0")
cfg = get_cfg(opts=["MODEL.PRETRAIN", False, "MODEL.DEVICE", str(cuda), "MODEL.USE_SYNC_BN", False])
images, targets = make_batch(8, seed=100, n_objects=6, device=cuda)
torch.manual_seed(0)
model = KeypointDetector(cfg)
init_like_trained(model, std=0.01, seed=0)
model = model.to(cuda).train()
opt = build_optimizer(model, cfg)
lc = model.heads.loss_evaluator
bufs = {}
origc = DL.Loss_Computation.compute_pairs_kpts_loss
def patched(self, preds, pred_targets, batch_weight):
    slot = bufs.setdefault(self._dbg_slot, {})
    def put(k, v):
        v = v.detach().double().reshape(-1)
        if k not in slot: slot[k] = torch.zeros_like(v)
        slot[k].copy_(v)
    m2d = pred_targets['extra_kpts_2d_mask'].float()
    instance_num = pred_targets['obj_valid'].sum()
    scale = instance_num / batch_weight
    pred = preds['pairs_kpt_depths_all']
    pmask = preds['pairs_kpt_depths_mask'] > 0
    found = pred_targets['find_pcl'].bool().unsqueeze(-1)
    valid = (pmask & found).float()
    invalid = ((~pmask) & found).float()
    target = pred_targets['depth_3D'].unsqueeze(-1).expand_as(pred)
    w = self.loss_weights['pairs_kpts_depth_loss']
    per = w * self.reg_loss_fnc(pred, target, reduction='none')
    valid_l = per * valid
    invalid_l = w * self.reg_loss_fnc(pred.detach(), target, reduction='none') * invalid
    n_valid, n_invalid = valid.sum(), invalid.sum()
    put("instance_num", instance_num); put("scale", scale); put("n_valid", n_valid); put("n_invalid", n_invalid)
    put("valid_l_sum", valid_l.sum()); put("invalid_l_sum", invalid_l.sum()); put("per_sum", per.sum()); put("per_rowsum_sum", per.sum(1).sum())
    put("valid_total", valid_l.sum() / torch.clamp(n_valid, min=1) * scale)
    return origc(self, preds, pred_targets, batch_weight)
DL.Loss_Computation.compute_pairs_kpts_loss = patched
seen = {}
orig = DL.Loss_Computation.__call__
def spy(self, predictions, tg):
    out = orig(self, predictions, tg)
    seen["pred"] = {k: (None if v is None else v.detach().clone()) for k, v in predictions.items()}
    return out
DL.Loss_Computation.__call__ = spy
for it in range(5):
    lc._dbg_slot = "graph"
    ld, _ = train_step(model, opt, images, targets, cfg.SOLVER.GRAD_NORM_CLIP)
    torch.cuda.synchronize()
    g = {k: v.clone() for k, v in bufs["graph"].items()}
    lc.use_graph = False
    lc._dbg_slot = "eager"
    with torch.no_grad():
        ld2, _ = orig(lc, seen["pred"], targets)
    lc.use_graph = True
    e = bufs["eager"]
    print(it, "extra_kpts_depth_loss graph %.4f eager %.4f" % (float(ld['extra_kpts_depth_loss'].detach()), float(ld2['extra_kpts_depth_loss'])))
    for k in g:
        print("     %-16s graph %.6g eager %.6g" % (k, float(g[k].sum()), float(e[k].sum())))
