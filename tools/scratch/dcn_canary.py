"""Out-of-bounds writes of the DCNv2 forward / backward: every output and the workspace sit between canary zones (1 M floats each
side, a bit pattern) inside one allocation; offsets with a far tail and a few extreme values; the canaries must survive."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from dcd_amd import _lib, _ext

dev = torch.device("cuda:0")
L = _lib.lib()
PAT = 0x7FC0DEAD
G = 1 << 20
def carve(sizes):
    total = sum(s + G for s in sizes) + G
    buf = torch.full((total,), PAT, dtype=torch.int32, device=dev)
    views, o = [], G
    for s in sizes:
        views.append((o, s)); o += s + G
    return buf, views
def check(buf, views, names):
    m = torch.ones(buf.numel(), dtype=torch.bool, device=dev)
    for o, s in views:
        m[o:o + s] = False
    bad = ((buf != PAT) & m).nonzero().flatten()
    if bad.numel():
        hits = {}
        for i in bad.tolist()[:2000]:
            # nearest view
            k = min(range(len(views)), key=lambda j: min(abs(i - views[j][0]), abs(i - (views[j][0] + views[j][1]))))
            side = "before" if i < views[k][0] else "after"
            d = views[k][0] - i if side == "before" else i - (views[k][0] + views[k][1]) + 1
            hits.setdefault((names[k], side), []).append(d)
        return "%d canary words overwritten: %s" % (bad.numel(), {k: (len(v), min(v), max(v)) for k, v in hits.items()})
    return "ok"
torch.manual_seed(0)
for prec in (0, 2, 1):
  for (C, Co, H, W) in ((64, 64, 96, 320), (128, 64, 48, 160), (128, 128, 48, 160), (256, 64, 24, 80), (256, 256, 24, 80), (512, 256, 12, 40)):
    for variant in ("near", "far", "extreme"):
        B = 8
        x = torch.randn(B, C, H, W, device=dev)
        off = torch.randn(B, 18, H, W, device=dev) * 0.5
        if variant != "near":
            far = torch.rand(B, 18, H, W, device=dev) < 0.004
            off = torch.where(far, off * 20, off)
        if variant == "extreme":
            ex = torch.rand(B, 18, H, W, device=dev) < 0.0005
            off = torch.where(ex, torch.randn(B, 18, H, W, device=dev) * 300, off)
            off[0, 0, 0, 0] = 1e6; off[1, 3, 5, 7] = -1e6; off[2, 4, 1, 1] = float("inf"); off[3, 5, 2, 2] = float("nan")
        m = torch.sigmoid(torch.randn(B, 9, H, W, device=dev))
        w = torch.randn(Co, C, 3, 3, device=dev) / (C * 9) ** 0.5
        b = torch.zeros(Co, device=dev)
        gy = torch.randn(B, Co, H, W, device=dev)
        geom = (B, C, H, W, Co, 3, 3, 1, 1, 1, 1, 1, 1, 1)
        nws = L.dcd_dcn_v2_workspace_bytes(*geom)
        sizes = [B * Co * H * W, B * C * H * W, B * 18 * H * W, B * 9 * H * W, Co * C * 9, Co, (nws + 3) // 4]
        names = ["output", "grad_input", "grad_offset", "grad_mask", "grad_weight", "grad_bias", "workspace"]
        buf, views = carve(sizes)
        fl = buf.view(torch.float32)
        ptr = lambda k: buf.data_ptr() + 4 * views[k][0]
        st = _lib.stream_of(x)
        for rep in range(3):
            r1 = L.dcd_dcn_v2_forward(st, x.data_ptr(), w.data_ptr(), b.data_ptr(), off.data_ptr(), m.data_ptr(), ptr(0), *geom, prec, ptr(6), nws)
            r2 = L.dcd_dcn_v2_backward(st, x.data_ptr(), w.data_ptr(), b.data_ptr(), off.data_ptr(), m.data_ptr(), gy.data_ptr(), ptr(1), ptr(2), ptr(3),
                                       ptr(4), ptr(5), *geom, prec, ptr(6), nws)
            torch.cuda.synchronize()
        res = check(buf, views, names)
        if res != "ok" or variant == "near":
            print("prec %d %3d->%3d@%3dx%3d %-8s status %d %d : %s" % (prec, C, Co, H, W, variant, r1, r2, res), flush=True)
print("done")
