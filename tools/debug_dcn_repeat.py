import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dcd_amd import _ext
from oracle import dcn_oracle
dev = torch.device("cuda:0")
a = (3, 3, 1, 1, 1, 1, 1, 1, 1)
names = ("grad_input", "grad_offset", "grad_mask", "grad_weight", "grad_bias")
for (B, C, Co, H, W, osc) in [(2, 512, 256, 3, 10, 0.1), (2, 256, 256, 6, 20, 0.1), (2, 256, 128, 6, 20, 0.5), (2, 128, 128, 12, 40, 0.1), (2, 64, 64, 24, 80, 0.1), (2, 512, 256, 3, 10, 2.0)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, C, H, W, generator=g); off = torch.randn(B, 18, H, W, generator=g) * osc
    m = torch.sigmoid(torch.randn(B, 9, H, W, generator=g)); w = torch.randn(Co, C, 3, 3, generator=g) / (C * 9) ** 0.5
    b = torch.randn(Co, generator=g); gy = torch.randn(B, Co, H, W, generator=g)
    ref = dcn_oracle.dcn_v2_backward(x, w, b, off, m, gy, *a)
    xd, wd, bd, od, md, gd = (t.to(dev) for t in (x, w, b, off, m, gy))
    worst = [0.0] * 5; spread = [0.0] * 5
    first = None
    for it in range(12):
        junk = torch.randn(1 << (16 + it % 5), device=dev) * 1e6       # perturb allocator / leave garbage behind
        del junk
        got = _ext.dcn_v2_backward(xd, wd, bd, od, md, gd, *a)
        got = [t.cpu() for t in got]
        if first is None: first = got
        for i in range(5):
            s = ref[i].abs().max().item() + 1e-12
            worst[i] = max(worst[i], (got[i] - ref[i]).abs().max().item() / s)
            spread[i] = max(spread[i], (got[i] - first[i]).abs().max().item() / s)
    print((B, C, Co, H, W, osc), "err vs oracle", " ".join("%s %.1e" % (n[5:], e) for n, e in zip(names, worst)), "| run-to-run", " ".join("%.1e" % e for e in spread))
