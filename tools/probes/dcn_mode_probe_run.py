"""Each captured layer's DCNv2 forward + backward (fp32 and split-bf16) under the DCD_DCN_HANDOVER mode of the environment (argv: tag)
-> /tmp/dcn_out_<tag>.pt; with ORACLE=1 also the CPU oracle's fp32 results -> /tmp/dcn_out_oracle.pt."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from dcd_amd import _ext
dev = torch.device("cuda:0")
tag = sys.argv[1]
layers = torch.load("/tmp/dcn_layers.pt")
outs = []
for l in layers:
    x, w, b, off, m = (l[k].to(dev) for k in ("x", "w", "b", "off", "m"))
    g = torch.Generator(device=dev).manual_seed(7)
    res = {}
    for prec in ("f32", "bf16x3"):
        kw = {} if prec == "f32" else {"precision": prec}
        # twice: the launch policy of the second call knows the first call's far count (the "auto" mode's state)
        for _ in range(2):
            y = _ext.dcn_v2_forward(x, w, b, off, m, *l["geom"], **kw)
            gy = torch.randn(y.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
            gr = _ext.dcn_v2_backward(x, w, b, off, m, gy, *l["geom"], **kw)
        res[prec] = [y.cpu()] + [t.cpu() for t in gr]
    outs.append(res)
torch.save(outs, "/tmp/dcn_out_%s.pt" % tag)
if os.environ.get("ORACLE") == "1":
    from oracle import dcn_oracle
    dcn_oracle.build()
    ref = []
    only = [int(v) for v in os.environ.get("ORACLE_LAYERS", "").split()] if os.environ.get("ORACLE_LAYERS") else None
    for li, l in enumerate(layers):
        if only is not None and li not in only:
            ref.append(None)
            continue
        y = dcn_oracle.dcn_v2_forward(l["x"], l["w"], l["b"], l["off"], l["m"], *l["geom"])
        gy = torch.randn(y.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(7)).cpu()
        gr = dcn_oracle.dcn_v2_backward(l["x"], l["w"], l["b"], l["off"], l["m"], gy, *l["geom"])
        ref.append([y] + list(gr))
    torch.save(ref, "/tmp/dcn_out_oracle.pt")
print("done", tag)
