"""argv: reference file, then other files: per file the worst / median relative gradient error against the reference and the
parameters furthest off."""
import sys, torch
ref = torch.load(sys.argv[1])
for f in sys.argv[2:]:
    d = torch.load(f)
    rel = []
    for n, g in ref["grads"].items():
        o = d["grads"][n]
        rel.append((((o - g).norm() / g.norm().clamp_min(1e-20)).item(), n, g.norm().item()))
    rel.sort(reverse=True)
    print("%s: loss %.6f (ref %.6f)  worst %.3e  median %.3e" % (f, d["loss"], ref["loss"], rel[0][0], rel[len(rel) // 2][0]))
    for v, n, gn in rel[:12]:
        print("      %.3e  %-75s |g| %.3e" % (v, n, gn))
    offs = [(v, n) for v, n, _ in rel if "conv_offset_mask.weight" in n]
    print("      offset convs: " + " ".join("%.2e" % v for v, _ in offs))
