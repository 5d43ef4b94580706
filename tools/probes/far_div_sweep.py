"""DCN backward time per layer against the far-sample hand-over limit (DCD_FAR_DIV / DCD_FAR_DIV_WIDE: the one-pass kernel keeps a
call while at most 1 in DIV offset coordinates is displaced by 3 px or more) over offset scales: where the generic kernels
start to pay.  python tools/probes/far_div_sweep.py [B]"""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = """
import sys, torch
sys.path.insert(0, %r)
from dcd_amd import _ext
B = %d
dev = torch.device('cuda:0')
for C, Co, H, W in %r:
    res = []
    for osc in %r:
        g = torch.Generator(device=dev).manual_seed(1)
        x = torch.randn(B, C, H, W, device=dev, generator=g); off = torch.randn(B, 18, H, W, device=dev, generator=g) * osc
        m = torch.sigmoid(torch.randn(B, 9, H, W, device=dev, generator=g)); w = torch.randn(Co, C, 3, 3, device=dev, generator=g) / (C * 9) ** 0.5
        b = torch.zeros(Co, device=dev); gy = torch.randn(B, Co, H, W, device=dev, generator=g); a = (3, 3, 1, 1, 1, 1, 1, 1, 1)
        for _ in range(2):
            _ext.dcn_v2_backward(x, w, b, off, m, gy, *a); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(5): _ext.dcn_v2_backward(x, w, b, off, m, gy, *a)
        e1.record(); torch.cuda.synchronize()
        far = (off.abs() >= 3).float().mean().item()
        res.append('%%.3f (far %%.3f)' %% (e0.elapsed_time(e1) / 5, far))
    print('%%3d->%%3d @%%3dx%%3d ' %% (C, Co, H, W) + '  '.join(res))
"""
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
layers = [(64, 64, 96, 320), (128, 64, 48, 160), (128, 128, 48, 160), (256, 128, 24, 80)]
scales = [0.5, 1.0, 1.3, 1.6, 2.0, 3.0]
print("B =", B, "offset scales", scales)
for div in (64, 24, 12, 6, 3, 1):
    env = dict(os.environ, DCD_FAR_DIV=str(div), DCD_FAR_DIV_WIDE=str(div))
    out = subprocess.run([sys.executable, "-c", code % (R, B, layers, scales)], capture_output=True, text=True, env=env)
    print("== 1 in", div)
    print(out.stdout.strip(), out.stderr.strip()[-300:] if out.returncode else "")
