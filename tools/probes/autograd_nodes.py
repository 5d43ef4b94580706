"""Node types of the autograd graph of one train step's loss, with the input shapes of the view-backward nodes (select / slice /
unfold / index: each is a zero fill + a copy / scatter in the backward) and the fan-in of every node input (> 1 = additions)."""
import argparse, collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--objects", type=int, default=6)
    ap.add_argument("--precision", default="f32")
    ap.add_argument("--scaling", default="weak")
    ap.add_argument("--amp", action="store_true")
    args = ap.parse_args()
    import bench
    device = torch.device("cuda", 0)
    cfg, model, optimizer, images, targets = bench.build_everything(args, device, 1, 0)[:5]
    loss_dict, _ = model(images, targets)
    total = getattr(loss_dict, "total", None)
    if total is None:
        total = sum(loss_dict.values())
    seen, stack = set(), [total.grad_fn]
    types = collections.Counter()
    fanin = collections.Counter()
    views = collections.Counter()
    consumers = collections.defaultdict(list)
    while stack:
        n = stack.pop()
        if n is None or n in seen:
            continue
        seen.add(n)
        name = type(n).__name__
        types[name] += 1
        if any(k in name for k in ("Select", "Slice", "Unfold", "Index", "Unbind", "Expand", "Sum", "Cat", "Stack", "Clone", "Copy")):
            shape = getattr(n, "_saved_self_sym_sizes", None)
            views[(name, tuple(shape) if shape is not None else None)] += 1
        for nxt, idx in n.next_functions:
            if nxt is not None:
                fanin[(nxt, idx)] += 1
                consumers[(nxt, idx)].append(name)
                stack.append(nxt)
    print("nodes:", sum(types.values()))
    for k, v in types.most_common(60):
        print("  %4d  %s" % (v, k))
    print("view-like backward nodes by input shape:")
    for (k, sh), v in sorted(views.items(), key=lambda x: -x[1])[:50]:
        print("  %4d  %s %s" % (v, k, sh))
    multi = collections.Counter()
    for (n, idx), c in fanin.items():
        if c > 1:
            multi[(type(n).__name__, c)] += 1
    pat = collections.Counter()
    for (n, idx), c in fanin.items():
        if c > 1:
            pat[(type(n).__name__, tuple(sorted(consumers[(n, idx)])))] += 1
    print("producer <- consumers (count):")
    for (k, cons), v in sorted(pat.items(), key=lambda x: -x[1]):
        print("  %3d  %s <- %s" % (v, k, ", ".join(cons)))
    print("inputs fed by more than one consumer (each extra consumer = one addition in the backward):")
    for (k, c), v in sorted(multi.items(), key=lambda x: -x[1]):
        print("  %4d  %s x%d" % (v, k, c))


if __name__ == "__main__":
    main()
