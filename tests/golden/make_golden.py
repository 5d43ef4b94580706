"""Generates tests/golden/*.npz by RUNNING THE REFERENCE'S OWN PYTHON (imported from /root/reference) on the CPU.

Run in the build container only (`python tests/golden/make_golden.py`); /root/reference does not exist on the
GPU box, so only the produced vectors travel.  Nothing from the reference is copied: its modules are imported
where they lie and executed unmodified.

What has to be supplied for the import to work in this image (and why it does not touch the arithmetic):
  * third-party packages the reference imports but the image lacks -- yacs, cv2, shapely, skimage, torchvision,
    pycocotools, iopath -- get empty stand-in modules.  None of them computes anything on the paths captured
    here, with ONE exception: `shapely.geometry.Polygon` feeds the logging-only '3D_IoU' value; the stand-in
    implements rectangle intersection by convex clipping, so that one logged value is NOT reference-pinned
    (it is excluded from the fixtures' compared keys and noted in DESIGN.md).
  * the reference's compiled `_ext` module (DCNv2).  Its C++ cannot be built here (needs <TH/TH.h>), so `_ext`
    is the C oracle (oracle/dcn_v2_oracle.c), itself pinned by the reference's known-answer and gradcheck tests.
    Fixtures that involve `_ext` therefore pin the MODEL SHELL (DLA-34 wiring, heads, losses) given that op.
  * `torch.cuda.FloatTensor` is aliased to `torch.FloatTensor` while `select_topk` runs, because the function
    asserts CUDA tensors (DGDE/model/layers/utils.py:83-84,93); its body is executed unchanged.
  * GMW/main.py cannot be imported (argparse / tensorboard / cv2 at import time); `compute_z` and `get_up` are
    compiled from the file's AST and executed as they are.
Inputs are regenerated in the tests from the same numpy seeds (helpers in tests/golden_inputs.py).
"""
import ast
import os
import sys
import types
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/DGDE"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import golden_inputs as gi  # noqa: E402  (shared seeded input builders)
from oracle import dcn_oracle, torch_ops  # noqa: E402


def install_stubs():
    from dcd_amd.config.cfgnode import CfgNode

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("yacs")
    mod("yacs.config", CfgNode=CfgNode)
    mod("cv2", setNumThreads=lambda n: None)

    class Polygon:  # stand-in for shapely (logging metric only)
        def __init__(self, pts):
            self.pts = torch.as_tensor(np.asarray(pts), dtype=torch.float64).reshape(-1, 2)
        is_valid = True

        @property
        def area(self):
            return float(torch_ops._poly_area(self.pts)) if self.pts.shape[0] >= 3 else 0.0

        def intersection(self, other):
            out = torch_ops._clip_convex(self.pts, other.pts)
            return Polygon(out.numpy()) if out.shape[0] >= 3 else Polygon(np.zeros((0, 2)))
    mod("shapely")
    mod("shapely.geometry", Polygon=Polygon)
    mod("skimage")
    mod("skimage.transform")
    tv = mod("torchvision")
    tv.ops = mod("torchvision.ops")
    tv.ops.roi_align = mod("torchvision.ops.roi_align")
    tv.transforms = mod("torchvision.transforms")
    tv.transforms.functional = mod("torchvision.transforms.functional")
    mod("pycocotools")
    mod("pycocotools.mask")
    mod("iopath")
    mod("iopath.common")
    mod("iopath.common.file_io", PathManager=object)
    sys.modules["_ext"] = dcn_oracle          # DCNv2 native op := the C oracle (see module docstring)


def ref_cfg(opts=()):
    from config import cfg
    c = cfg.clone()
    c.defrost()
    c.merge_from_file(os.path.join(REF, "runs", "DGDE.yaml"))
    c.merge_from_list(["MODEL.DEVICE", "cpu", "MODEL.PRETRAIN", False, "MODEL.USE_SYNC_BN", False] + list(opts))
    return c


def to_ref_targets(targets):
    """Our synthetic ParamsList objects -> the reference's ParamsList with the reference's Calibration."""
    from structures.params_3d import ParamsList as RefParams
    from data.datasets.kitti_utils import Calibration as RefCalib
    out = []
    for t in targets:
        r = RefParams(t.size, is_train=True)
        for name in t.fields():
            v = t.get_field(name)
            if name == "calib":
                c = RefCalib.__new__(RefCalib)
                c.P = np.asarray(v.P, dtype=np.float64)
                c.c_u, c.c_v, c.f_u, c.f_v = c.P[0, 2], c.P[1, 2], c.P[0, 0], c.P[1, 1]
                c.b_x, c.b_y = c.P[0, 3] / (-c.f_u), c.P[1, 3] / (-c.f_v)
                v = c
            r.add_field(name, v)
        out.append(r)
    return out


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print("wrote %-28s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def T(x):
    return x.detach().cpu().numpy()


# ------------------------------------------------------------------------------------------------
def golden_edge_depth():
    from model.anno_encoder import Anno_Encoder
    enc = Anno_Encoder(ref_cfg())
    kps, k3, rot, P, mask = gi.edge_inputs()
    a = torch.from_numpy(kps).requires_grad_()
    b = torch.from_numpy(k3).requires_grad_()
    out = {}
    d_eval, _ = enc.decode_pairs_kpts_depth(a, b, torch.from_numpy(rot), torch.from_numpy(P), training=False)
    out["eval_depth"] = T(d_eval)
    (d_eval * torch.from_numpy(gi.edge_grad_weights(d_eval.shape))).sum().backward()
    out["eval_grad_kps"], out["eval_grad_kps3d"] = T(a.grad), T(b.grad)
    a.grad = None
    b.grad = None
    d_tr, m_tr = enc.decode_pairs_kpts_depth(a, b, torch.from_numpy(rot), torch.from_numpy(P), training=True,
                                             kpts_2d_mask=torch.from_numpy(mask))
    out["train_depth"], out["train_mask"] = T(d_tr), T(m_tr)
    d_tr.sum().backward()
    out["train_grad_kps"], out["train_grad_kps3d"] = T(a.grad), T(b.grad)
    # GMW twin: compile compute_z / get_up from the file's AST (the module itself is not importable here)
    src = open("/root/reference/GMW/main.py").read()
    tree = ast.parse(src)
    fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("compute_z", "get_up")]
    ns = {"torch": torch}
    exec(compile(ast.Module(body=fns, type_ignores=[]), "GMW/main.py", "exec"), ns)
    kn = gi.normalise_kps(kps, P)
    z, idx = ns["compute_z"](torch.from_numpy(kn), torch.from_numpy(k3), torch.from_numpy(rot))
    out["gmw_z"], out["gmw_idx"] = T(z), T(idx)
    save("edge_depth", **out)


def golden_losses():
    from model.layers.focal_loss import FocalLoss
    from model.layers.iou_loss import IOULoss
    from model.head.depth_losses import RegWeightedL1Loss
    from model.head.detector_loss import Real_MultiBin_loss
    cfg = ref_cfg()
    pred, tgt = gi.focal_inputs()
    p = torch.from_numpy(pred).requires_grad_()
    loss, npos = FocalLoss(cfg.MODEL.HEAD.LOSS_PENALTY_ALPHA, cfg.MODEL.HEAD.LOSS_BETA, cfg=cfg)(p, torch.from_numpy(tgt))
    loss.backward()
    out = dict(focal_loss=T(loss), focal_npos=T(npos), focal_grad=T(p.grad))
    bp, bt = gi.giou_inputs()
    q = torch.from_numpy(bp).requires_grad_()
    losses, ious = IOULoss("giou")(q, torch.from_numpy(bt))
    losses.sum().backward()
    out.update(giou_losses=T(losses), giou_ious=T(ious), giou_grad=T(q.grad))
    kp, kt, dep = gi.regweighted_inputs()
    out["regweighted"] = T(RegWeightedL1Loss()(torch.from_numpy(kp), torch.from_numpy(kt), torch.from_numpy(dep)))
    vo, go = gi.multibin_inputs()
    out["multibin"] = T(Real_MultiBin_loss(torch.from_numpy(vo), torch.from_numpy(go), num_bin=4))
    save("losses", **out)


def golden_anno_encoder():
    from model.anno_encoder import Anno_Encoder
    enc = Anno_Encoder(ref_cfg())
    d = gi.anno_inputs()
    t = {k: torch.from_numpy(v) for k, v in d.items()}
    calibs = gi.ref_like_calibs(d["P_img"])
    out = {}
    out["encode_box3d"] = T(enc.encode_box3d(t["rotys"], t["dims"], t["locs"]))
    out["decode_depth"] = T(enc.decode_depth(t["depth_off"], None))
    out["decode_dimension"] = T(enc.decode_dimension(t["cls"], t["dims_off"]))
    out["decode_location"] = T(enc.decode_location_flatten(t["points"], t["offsets"], t["depths"], calibs, t["pad"],
                                                           t["batch_idxs"]))
    out["kp_depths"] = T(enc.decode_depth_from_keypoints_batch(t["kp10"], t["dims"], calibs, t["batch_idxs"]))
    ro, al = enc.decode_axes_orientation(t["ori"].clone(), t["locs"])
    out["rotys"], out["alphas"] = T(ro), T(al)
    out["kpts_2d_img"] = T(enc.decode_kpts_2d_img(t["kp73"], t["points"], t["offsets"],
                                                  t["pad"][t["batch_idxs"]].unsqueeze(1).expand_as(t["kp73"])))
    save("anno_encoder", **out)


def golden_decode():
    from model.layers import utils as U
    heat = gi.heat_inputs()
    h = torch.from_numpy(heat)
    nms = U.nms_hm(h)
    orig = torch.cuda.FloatTensor
    torch.cuda.FloatTensor = torch.FloatTensor          # let the CUDA-only asserts pass on CPU tensors
    try:
        scores, inds, clses, ys, xs = U.select_topk(nms, K=50)
    finally:
        torch.cuda.FloatTensor = orig
    feat, pts = gi.poi_inputs()
    poi = U.select_point_of_interest(feat.shape[0], torch.from_numpy(pts), torch.from_numpy(feat))
    save("decode", nms=T(nms), scores=T(scores), inds=T(inds), clses=T(clses), ys=T(ys), xs=T(xs), poi=T(poi))


def golden_loss_computation():
    from model.head.detector_loss import Loss_Computation
    cfg = ref_cfg(["INPUT.WIDTH_TRAIN", 320, "INPUT.HEIGHT_TRAIN", 96])
    preds, targets = gi.loss_inputs()
    cls = torch.from_numpy(preds["cls"]).requires_grad_()
    reg = torch.from_numpy(preds["reg"]).requires_grad_()
    loss_dict, log = Loss_Computation(cfg)({"cls": cls, "reg": reg}, to_ref_targets(targets))
    sum(loss_dict.values()).backward()
    out = {"loss_" + k: T(v) for k, v in loss_dict.items()}
    out.update({"log_" + k: np.float64(v) for k, v in log.items()})
    out["grad_cls"], out["grad_reg_sum_c"] = T(cls.grad), T(reg.grad.sum(1))
    out["grad_reg_abs_per_channel"] = T(reg.grad.abs().sum((0, 2, 3)))
    save("loss_computation", **out)


def golden_gen_data():
    """TEST.GENERATE_GMW: the per-object records Loss_Computation collects for GMW (detector_loss.py:148-173)."""
    from model.head.detector_loss import Loss_Computation
    cfg = ref_cfg(["INPUT.WIDTH_TRAIN", 320, "INPUT.HEIGHT_TRAIN", 96, "TEST.GENERATE_GMW", True])
    preds, targets = gi.loss_inputs()
    lc = Loss_Computation(cfg)
    with torch.no_grad():
        lc({"cls": torch.from_numpy(preds["cls"]), "reg": torch.from_numpy(preds["reg"])}, to_ref_targets(targets))
    gd = lc.gen_data
    save("gen_data", kpts_2d=np.array(gd["kpts_2d"][0], np.float32), kpts_3d=np.array(gd["kpts_3d"][0], np.float32),
         pred_rot=np.array(gd["pred_rot"][0], np.float32), gt_location=np.array(gd["gt_location"][0], np.float32),
         pred_location=np.array(gd["pred_location"][0], np.float32), img_idx=np.array(gd["img_idx"][0]),
         keys=np.array(list(gd.keys())))


def golden_post_processor():
    """PostProcessor.forward (detector_infer.py:86-213) on PINNED predictor outputs: image 0 of the `loss_inputs()` maps, at the
    configured threshold (TEST.DETECTIONS_THRESHOLD) and once more with TEST.GENERATE_GMW.  `select_topk` asserts CUDA tensors
    (layers/utils.py:83-84,93), so torch.cuda.FloatTensor is aliased to torch.FloatTensor while it runs."""
    from model.head.detector_infer import make_post_processor
    cfg = ref_cfg(["INPUT.WIDTH_TRAIN", 320, "INPUT.HEIGHT_TRAIN", 96])
    preds, targets = gi.loss_inputs()
    ref_targets = to_ref_targets(targets)[:1]
    pp = make_post_processor(cfg)
    cls, reg = torch.from_numpy(preds["cls"][:1]), torch.from_numpy(preds["reg"][:1])
    orig = torch.cuda.FloatTensor
    torch.cuda.FloatTensor = torch.FloatTensor
    try:
        with torch.no_grad():
            result, info, vis = pp({"cls": cls, "reg": reg}, ref_targets)
            pp.generate_data = True
            result_g, _, vis_g = pp({"cls": cls, "reg": reg}, ref_targets)
    finally:
        torch.cuda.FloatTensor = orig
    assert torch.equal(result, result_g)
    save("post_processor", result=T(result), vis_scores=T(info["vis_scores"]), uncertainty_conf=T(info["uncertainty_conf"]),
         estimated_depth_error=T(info["estimated_depth_error"]), keypoints=T(vis["keypoints"]), proj_center=T(vis["proj_center"]),
         min_uncertainty=T(vis["min_uncertainty"]), pred_extra_kpts_2d=T(vis["pred_extra_kpts_2d"]),
         pred_extra_kpts_3d=T(vis["pred_extra_kpts_3d"]), gen_kpts_2d=T(vis_g["gen_pred_extra_kpts_2d"]),
         gen_kpts_3d=T(vis_g["gen_pred_extra_kpts_3d"]), threshold=np.float64(pp.det_threshold))


def golden_solver():
    """Trainer harness: per-parameter learning rates, the warm-up / step-decay schedule driven exactly as
    DGDE/engine/trainer.py:152-155 drives it, and one AdamW step -- from the reference's own solver package.
    `solver/fastai_optim.py:3` needs `collections.Iterable` (removed in Python 3.10): aliased to collections.abc.Iterable."""
    import collections, collections.abc
    collections.Iterable = collections.abc.Iterable
    from solver import build_optimizer, build_scheduler
    cfg = ref_cfg(["SOLVER.LR_WARMUP", True, "SOLVER.WARMUP_STEPS", 200, "SOLVER.MAX_ITERATION", 3000, "SOLVER.STEPS", (2000, 2600)])
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    opt = build_optimizer(net, cfg)
    sched, warm = build_scheduler(opt, 100, cfg.SOLVER)
    x = torch.linspace(-1, 1, 24).reshape(4, 6)
    lrs_w, lrs_b = [], []
    import copy
    snap = {}
    for it in range(3000):
        if it < 4:
            if it == 3:       # the state a checkpoint written after iteration 2 holds (utils/check_point.py:31-43)
                snap = {"opt": copy.deepcopy(opt.state_dict()), "sched": {k: v for k, v in sched.state_dict().items() if k != "lr_lambdas"},
                        "p3": np.concatenate([p.detach().numpy().ravel() for p in net.parameters()])}
            opt.zero_grad()
            net(x).square().sum().backward()
            opt.step()
            if it == 3:
                snap["p4"] = np.concatenate([p.detach().numpy().ravel() for p in net.parameters()])
        if it < cfg.SOLVER.WARMUP_STEPS:
            warm.step(it)
        else:
            sched.step(it)
        lrs_w.append(opt.param_groups[0]["lr"])      # first parameter is a weight, second a bias
        lrs_b.append(opt.param_groups[1]["lr"])
    osd = snap["opt"]
    n = len(osd["param_groups"])
    save("solver", lr_weight=np.array(lrs_w), lr_bias=np.array(lrs_b),
         params_after_3_steps=snap["p3"], params_after_4_steps=snap["p4"], n_groups=np.array(n),
         ckpt_group_lr=np.array([g["lr"] for g in osd["param_groups"]]),
         ckpt_group_initial_lr=np.array([g["initial_lr"] for g in osd["param_groups"]]),
         ckpt_group_params=np.array([g["params"] for g in osd["param_groups"]]),
         ckpt_group_wd=np.array([g["weight_decay"] for g in osd["param_groups"]]),
         ckpt_step=np.array([float(osd["state"][i]["step"]) for i in range(n)]),
         ckpt_sched_base_lrs=np.array(snap["sched"]["base_lrs"]), ckpt_sched_last_lr=np.array(snap["sched"]["_last_lr"]),
         ckpt_sched_last_epoch=np.array(snap["sched"]["last_epoch"]),
         **{"ckpt_exp_avg_%d" % i: osd["state"][i]["exp_avg"].numpy() for i in range(n)},
         **{"ckpt_exp_avg_sq_%d" % i: osd["state"][i]["exp_avg_sq"].numpy() for i in range(n)})


def golden_config():
    """The reference's config tree (config/defaults.py merged with runs/DGDE.yaml) flattened to dotted keys -> cfg.json; pins
    dcd_amd/config/{defaults,dgde_run}.py key by key."""
    import json
    from config import cfg
    c = cfg.clone()
    c.defrost()
    c.merge_from_file(os.path.join(REF, "runs", "DGDE.yaml"))

    def flat(node, prefix=""):
        out = {}
        for k, v in node.items():
            if isinstance(v, dict):
                out.update(flat(v, prefix + k + "."))
            else:
                out[prefix + k] = list(v) if isinstance(v, tuple) else v
        return out
    with open(os.path.join(HERE, "cfg.json"), "w") as f:
        json.dump(flat(c), f, indent=0, sort_keys=True)
    print("cfg.json: %d keys" % len(flat(c)))


def golden_model():
    from model.detector import KeypointDetector
    cfg = ref_cfg(["INPUT.WIDTH_TRAIN", 320, "INPUT.HEIGHT_TRAIN", 96])
    torch.manual_seed(0)
    model = KeypointDetector(cfg)
    gi.name_hashed_init(model)
    model.train()
    images, targets = gi.model_inputs()
    ref_targets = to_ref_targets(targets)
    feats = model.backbone(images)
    pred = model.heads.predictor(feats, ref_targets)
    out = dict(feat_mean=T(feats.mean()), feat_abs=T(feats.abs().mean()), feat_slice=T(feats[:, :4, ::6, ::16]),
               cls_slice=T(pred["cls"][:, :, ::4, ::8]), reg_slice=T(pred["reg"][:, ::25, ::6, ::16]),
               reg_abs=T(pred["reg"].abs().mean((0, 2, 3))))
    gi.name_hashed_init(model)       # fresh BN running stats for the full step
    model.zero_grad()
    loss_dict, log = model(images, ref_targets)
    sum(loss_dict.values()).backward()
    out.update({"loss_" + k: T(v) for k, v in loss_dict.items()})
    names, norms = [], []
    for n, p in model.named_parameters():
        names.append(n)
        norms.append(0.0 if p.grad is None else float(p.grad.double().norm()))
    out["param_names"] = np.array(names)
    out["grad_norms"] = np.array(norms)
    out["state_keys"] = np.array(list(model.state_dict().keys()))
    out["bn_running_mean_sample"] = T(model.backbone.base.base_layer[1].running_mean)
    # eval decode (PostProcessor) on image 0, lowered threshold so detections exist at random init
    model.eval()
    model.heads.post_processor.det_threshold = 0.0
    orig = torch.cuda.FloatTensor
    torch.cuda.FloatTensor = torch.FloatTensor
    try:
        with torch.no_grad():
            result, eval_utils, _ = model(images[:1], ref_targets[:1])
    finally:
        torch.cuda.FloatTensor = orig
    out["eval_result"] = T(result)
    out["eval_vis_scores"] = T(eval_utils["vis_scores"])
    # the same eval pass with TEST.GENERATE_GMW: K-normalised keypoints for GMW inference (detector_infer.py:227-243)
    model.heads.post_processor.generate_data = True
    torch.cuda.FloatTensor = torch.FloatTensor
    try:
        with torch.no_grad():
            result_g, _, vis_g = model(images[:1], ref_targets[:1])
    finally:
        torch.cuda.FloatTensor = orig
    out["gen_result"] = T(result_g)
    out["gen_kpts_2d"] = T(vis_g["gen_pred_extra_kpts_2d"])
    out["gen_kpts_3d"] = T(vis_g["gen_pred_extra_kpts_3d"])
    save("model_96x320", **out)


def golden_model_f64():
    """Ground truth for the whole-model tolerance: the SAME reference model and inputs as golden_model, run
    in float64 -- backbone (stock convs + the f64 build of the C oracle as `_ext`) and predictor in double; the predictor
    returns float32 maps by construction (detector_predictor.py:203), so the loss arithmetic stays float32 on f64-accurate
    inputs.  Lets the tests measure how far each fp32 run (the reference's, ours on the GPU) sits from the exact result."""
    from model.detector import KeypointDetector
    cfg = ref_cfg(["INPUT.WIDTH_TRAIN", 320, "INPUT.HEIGHT_TRAIN", 96])
    torch.manual_seed(0)
    model = KeypointDetector(cfg)
    gi.name_hashed_init(model)
    model = model.double().train()
    images, targets = gi.model_inputs()
    images = images.double()
    ref_targets = to_ref_targets(targets)
    feats = model.backbone(images)
    pred = model.heads.predictor(feats, ref_targets)
    out = dict(feat_slice=T(feats[:, :4, ::6, ::16]), cls_slice=T(pred["cls"][:, :, ::4, ::8]),
               reg_slice=T(pred["reg"][:, ::25, ::6, ::16]), reg_abs=T(pred["reg"].double().abs().mean((0, 2, 3))))
    gi.name_hashed_init(model)
    model.zero_grad()
    loss_dict, _ = model(images, ref_targets)
    sum(loss_dict.values()).backward()
    out.update({"loss_" + k: T(v) for k, v in loss_dict.items()})
    out["param_names"] = np.array([n for n, _ in model.named_parameters()])
    out["grad_norms"] = np.array([0.0 if p.grad is None else float(p.grad.norm()) for _, p in model.named_parameters()])
    save("model_96x320_f64", **out)


def main():
    install_stubs()
    sys.path.insert(0, REF)
    os.chdir(REF)
    torch.set_num_threads(8)
    golden_edge_depth()
    golden_losses()
    golden_anno_encoder()
    golden_decode()
    golden_loss_computation()
    golden_gen_data()
    golden_post_processor()
    golden_solver()
    golden_config()
    golden_model()
    golden_model_f64()


if __name__ == "__main__":
    main()
