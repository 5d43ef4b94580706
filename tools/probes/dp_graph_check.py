import os, sys, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist
from dcd_amd.config import get_cfg
from dcd_amd.data.synthetic import make_batch
from dcd_amd.engine.trainer import GraphedTrainStep, build_optimizer, init_like_trained, train_step, prepare_data_parallel
from dcd_amd.model.detector import KeypointDetector
cuda = torch.device("cuda:0")
with socket.socket() as s_:
    s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=cuda)
cfg = get_cfg(opts=["MODEL.PRETRAIN", False, "MODEL.DEVICE", str(cuda), "MODEL.USE_SYNC_BN", True, "INPUT.WIDTH_TRAIN", 320, "INPUT.HEIGHT_TRAIN", 96])
batches = [make_batch(2, seed=s, n_objects=n, input_size=(320, 96), device=cuda) for s, n in ((3, 3), (4, 5), (7, 2), (3, 3))]
batches[1][1][0].get_field('calib').f_u *= 1.01
order = sys.argv[1:] or ["eager", "eager", "graph", "graph_capture_on_2"]
for mode in order:
    torch.manual_seed(0)
    model = KeypointDetector(cfg).to(cuda).train(); init_like_trained(model)
    opt = build_optimizer(model, cfg); prepare_data_parallel(model, cfg)
    for g_ in opt.param_groups: g_["lr"].fill_(0.0)
    step = GraphedTrainStep(model, opt, cfg.SOLVER.GRAD_NORM_CLIP, distributed=True) if mode.startswith("graph") else None
    seq = [2, 0, 1, 2] if mode == "graph_capture_on_2" else [0, 1, 2, 0]
    out = []
    for bi in seq:
        images, targets = batches[bi]
        ld, _ = step(images, targets) if step else train_step(model, opt, images, targets, cfg.SOLVER.GRAD_NORM_CLIP)
        out.append((bi, {k[:9]: round(float(v.detach()), 6) for k, v in ld.items()} if bi == 2 else None))
    print(mode, out)
dist.destroy_process_group()
