import torch
ref = torch.load("/tmp/dcn_out_oracle.pt")
o = {m: torch.load("/tmp/dcn_out_%s.pt" % m) for m in ("never", "always", "auto")}
names = ["out", "g_in", "g_w", "g_b", "g_off", "g_mask"]
def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))
for i in range(len(ref)):
    for prec in ("f32", "bf16x3"):
        line = "layer %2d %-6s" % (i, prec)
        for m in ("never", "always", "auto"):
            worst = max((rel(o[m][i][prec][k], ref[i][k]), names[k]) for k in range(len(ref[i])))
            line += " | %s vs oracle %.1e (%s)" % (m, worst[0], worst[1])
        line += " | never vs always %.1e" % max(rel(o["never"][i][prec][k], o["always"][i][prec][k]) for k in range(len(ref[i])))
        print(line)
