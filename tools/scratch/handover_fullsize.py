"""Full-size DCN layers at bs 8 with near / far-heavy offsets under a pinned hand-over policy (DCD_DCN_HANDOVER=never | always, one
process each): saves the outputs, or compares with the other process's.  usage: handover_fullsize.py save|cmp <dir>"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from dcd_amd import _ext

mode, d = sys.argv[1], sys.argv[2]
os.makedirs(d, exist_ok=True)
dev = torch.device("cuda:0")
LAYERS = [(64, 64, 96, 320), (128, 64, 48, 160), (128, 128, 48, 160), (256, 64, 24, 80), (256, 256, 24, 80), (512, 256, 12, 40)]
a = (3, 3, 1, 1, 1, 1, 1, 1, 1)
for prec in ("f32", "bf16"):
    for li, (C, Co, H, W) in enumerate(LAYERS):
        for osc in (0.5, 2.5, 6.0):
            g = torch.Generator().manual_seed(li * 10 + int(osc * 2))
            B = 8
            x = torch.randn(B, C, H, W, generator=g).to(dev)
            off = (torch.randn(B, 18, H, W, generator=g) * osc).to(dev)
            m = torch.sigmoid(torch.randn(B, 9, H, W, generator=g)).to(dev)
            w = (torch.randn(Co, C, 3, 3, generator=g) / (C * 9) ** 0.5).to(dev)
            b = torch.zeros(Co, device=dev)
            gy = torch.randn(B, Co, H, W, generator=g).to(dev)
            # two near calls first so that an "auto" policy would have settled; pinned policies ignore it
            y = _ext.dcn_v2_forward(x, w, b, off, m, *a, precision=prec)
            outs = [y] + list(_ext.dcn_v2_backward(x, w, b, off, m, gy, *a, precision=prec))
            torch.cuda.synchronize()
            path = os.path.join(d, "%s_%d_%g.pt" % (prec, li, osc))
            if mode == "save":
                torch.save([t.cpu() for t in outs], path)
            else:
                ref = torch.load(path)
                rel = [((t.cpu() - r).abs().max() / r.abs().max().clamp_min(1e-12)).item() for t, r in zip(outs, ref)]
                print("%-5s %3d->%3d@%3dx%3d off %.1f  [y gi goff gmask gw gb] %s" % (prec, C, Co, H, W, osc, " ".join("%.1e" % v for v in rel)), flush=True)
