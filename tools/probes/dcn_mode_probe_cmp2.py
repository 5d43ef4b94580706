"""never vs always per layer (no oracle); prints the layers that differ by more than 1e-5 and writes their indices to /tmp/dcn_diff_layers.txt"""
import torch
o = {m: torch.load("/tmp/dcn_out_%s.pt" % m) for m in ("never", "always")}
names = ["out", "g_in", "g_w", "g_b", "g_off", "g_mask"]
def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))
bad = []
for i in range(len(o["never"])):
    for prec in ("f32", "bf16x3"):
        d = [(rel(o["never"][i][prec][k], o["always"][i][prec][k]), names[k]) for k in range(6)]
        w = max(d)
        print("layer %2d %-6s never vs always: %s" % (i, prec, "  ".join("%s %.1e" % (n, v) for v, n in d)))
        if w[0] > 1e-5 and i not in bad:
            bad.append(i)
open("/tmp/dcn_diff_layers.txt", "w").write(" ".join(str(i) for i in bad))
print("layers that differ:", bad)
