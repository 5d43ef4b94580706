import torch
ref = torch.load("/tmp/dcn_out_oracle.pt")
o = {m: torch.load("/tmp/dcn_out_%s.pt" % m) for m in ("never", "always")}
names = ["out", "g_in", "g_w", "g_b", "g_off", "g_mask"]
def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))
for i, r in enumerate(ref):
    if r is None: continue
    for prec in ("f32", "bf16x3"):
        for m in ("never", "always"):
            print("layer %2d %-6s %-6s vs oracle: %s" % (i, prec, m, "  ".join("%s %.1e" % (names[k], rel(o[m][i][prec][k], r[k])) for k in range(6))))
