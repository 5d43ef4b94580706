import torch
c = torch.load("/tmp/twin_catch.pt"); n = c["name"]
g = {m: torch.load("/tmp/twin_grads_%s.pt" % m)[0] for m in ("auto", "always", "never")}
def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-20))
print("caught:", n, "replay", c["it"], "rel %.2e" % c["rel"])
print("always vs never %.2e | auto vs never %.2e | graph vs never %.2e | twin vs never %.2e | graph vs always %.2e | twin vs always %.2e" % (
    rel(g["always"][n], g["never"][n]), rel(g["auto"][n], g["never"][n]), rel(c["graph"][n], g["never"][n]), rel(c["twin"][n], g["never"][n]),
    rel(c["graph"][n], g["always"][n]), rel(c["twin"][n], g["always"][n])))
def worst(a, b):
    r = sorted(((rel(a[k], b[k]), k) for k in a if k in b and not k.endswith("conv.bias") and float(b[k].abs().max()) >= 1e-7), reverse=True)
    return "%.2e (%s), median %.2e" % (r[0][0], r[0][1], r[len(r) // 2][0])
print("over all parameters: always vs never", worst(g["always"], g["never"]))
print("                     auto vs never  ", worst(g["auto"], g["never"]))
print("                     graph vs never ", worst(c["graph"], g["never"]))
print("                     twin vs never  ", worst(c["twin"], g["never"]))
print("                     graph vs always", worst(c["graph"], g["always"]))
print("                     twin vs always ", worst(c["twin"], g["always"]))
