"""ATen ops of one whole train step by the Python line that issued them (forward) and by op name (backward): where the stock
elementwise / copy / fill / reduce launches of the step come from.  python tools/count_step_ops.py [--batch 8]"""
import argparse, collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode

VIEWS = {"view", "reshape", "expand", "expand_as", "slice", "select", "unsqueeze", "squeeze", "detach", "alias", "permute",
         "transpose", "t", "unbind", "as_strided", "_unsafe_view", "split", "split_with_sizes", "empty", "empty_like",
         "empty_strided", "_local_scalar_dense", "unfold", "_reshape_alias", "lift_fresh", "new_empty", "view_as",
         "is_same_size", "sym_size", "sym_stride", "sym_numel", "_to_copy_noop", "new_empty_strided", "result_type", "is_pinned",
         "set_", "stride", "size", "numel", "dim", "storage_offset", "is_contiguous", "_has_compatible_shallow_copy_type"}


class Counter(TorchDispatchMode):
    def __init__(self, by_line=True):
        super().__init__()
        self.by_fn = collections.Counter()
        self.by_op = collections.Counter()
        self.by_line = by_line

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        base = name.split(".")[1] if name.startswith("aten.") else name
        if base not in VIEWS:
            tag = "(no python frame)"
            for fr in reversed(traceback.extract_stack()[:-1]):
                if "/dcd_amd/" in fr.filename:
                    tag = "%s:%s:%d" % (os.path.basename(fr.filename), fr.name, fr.lineno)
                    break
            self.by_fn[tag + "  " + base] += 1
            self.by_op[base] += 1
        return func(*args, **(kwargs or {}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--objects", type=int, default=6)
    ap.add_argument("--precision", default="f32")
    ap.add_argument("--scaling", default="weak")
    ap.add_argument("--amp", action="store_true")
    ap.add_argument("--top", type=int, default=80)
    args = ap.parse_args()
    import bench
    from dcd_amd.engine import trainer
    device = torch.device("cuda", 0)
    cfg, model, optimizer, images, targets = bench.build_everything(args, device, 1, 0)[:5]
    for _ in range(2):
        trainer.train_step(model, optimizer, images, targets)
    torch.cuda.synchronize()
    with Counter() as c:
        trainer.train_step(model, optimizer, images, targets)
    print("one step: %d non-view ATen calls" % sum(c.by_op.values()))
    print("-- by op")
    for k, v in c.by_op.most_common(40):
        print("  %4d  %s" % (v, k))
    print("-- by issuing line (forward and python-side backward) / '(no python frame)' = autograd engine")
    for k, v in c.by_fn.most_common(args.top):
        print("  %4d  %s" % (v, k))


if __name__ == "__main__":
    main()
