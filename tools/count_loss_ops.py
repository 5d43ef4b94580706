"""ATen ops (= device launches, roughly) of the loss section by the Python function that issued them (eager GPU run with the loss graph off, dispatch-mode
counter keyed by the innermost frame inside dcd_amd/): where a fused kernel would remove the most launches."""
import os, sys, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode


class Counter(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.by_fn = collections.Counter()
        self.by_op = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        base = name.split(".")[1] if name.startswith("aten.") else name
        views = {"view", "reshape", "expand", "expand_as", "slice", "select", "unsqueeze", "squeeze", "detach", "alias", "permute",
                 "transpose", "t", "unbind", "as_strided", "_unsafe_view", "split", "split_with_sizes", "empty", "empty_like",
                 "empty_strided", "_local_scalar_dense", "unfold", "_reshape_alias", "lift_fresh", "new_empty", "view_as",
                 "is_same_size", "sym_size", "sym_stride", "sym_numel", "_to_copy_noop"}
        if base not in views:
            tag = "?"
            for fr in reversed(traceback.extract_stack()[:-1]):
                if "/dcd_amd/" in fr.filename:
                    tag = "%s:%s:%d" % (os.path.basename(fr.filename), fr.name, fr.lineno) if os.environ.get("BY_LINE") else \
                        "%s:%s" % (os.path.basename(fr.filename), fr.name)
                    break
            self.by_fn[tag] += 1
            self.by_op[name] += 1
        return func(*args, **(kwargs or {}))


def main():
    from dcd_amd.config import get_cfg
    from dcd_amd.data.synthetic import make_batch
    from dcd_amd.model.head import detector_loss
    os.environ["DCD_LOSS_GRAPH"] = "0"
    dev = torch.device("cuda:0")
    cfg = get_cfg(opts=["MODEL.PRETRAIN", False, "MODEL.DEVICE", "cuda"])
    ev = detector_loss.make_loss_evaluator(cfg)
    B = 8
    images, targets = make_batch(B, seed=1, n_objects=6, device=dev)
    M = targets[0].get_field("reg_mask").shape[0]
    C = sum(cfg.MODEL.HEAD.REGRESSION_CHANNELS[i][j] for i in range(len(cfg.MODEL.HEAD.REGRESSION_CHANNELS))
            for j in range(len(cfg.MODEL.HEAD.REGRESSION_CHANNELS[i]))) if hasattr(cfg.MODEL.HEAD, "REGRESSION_CHANNELS") else 415
    g = torch.Generator().manual_seed(0)
    preds = {"cls": torch.rand(B, 1, 96, 320, generator=g).clamp(1e-4, 1 - 1e-4).to(dev).requires_grad_(),
             "reg": None, "reg_pois": (0.1 * torch.randn(B, M, C, generator=g)).to(dev).requires_grad_()}
    with Counter() as c:
        loss_dict, _ = ev(preds, targets)
        total = sum(loss_dict.values())
    fwd_fn, fwd_op = c.by_fn, c.by_op
    with Counter() as c2:
        total.backward()
    print("forward: %d ops" % sum(fwd_fn.values()))
    for k, v in fwd_fn.most_common(int(os.environ.get('TOP', '30'))):
        print("  %4d  %s" % (v, k))
    print("backward: %d ops" % sum(c2.by_op.values()))
    for k, v in c2.by_op.most_common(15):
        print("  %4d  %s" % (v, k))


if __name__ == "__main__":
    main()
