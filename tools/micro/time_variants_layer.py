"""A/B timing of library variants (tools/micro/libdcd_<name>.so, build_variants.sh) on ONE layer and precision:
   python tools/micro/time_variants_layer.py C Co H W prec [off_scale]   (each variant in its own process)"""
import sys, os, shutil, subprocess, glob
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
main = os.path.join(R, "dcd_amd", "libdcd_hip.so")
code = """
import sys, torch
sys.path.insert(0, %r)
from dcd_amd import _ext
C, Co, H, W = %s
prec, osc = %r, %s
dev = torch.device('cuda:0'); B = 8
x = torch.randn(B, C, H, W, device=dev); off = torch.randn(B, 18, H, W, device=dev) * osc
m = torch.sigmoid(torch.randn(B, 9, H, W, device=dev)); w = torch.randn(Co, C, 3, 3, device=dev) / (C * 9) ** 0.5
b = torch.zeros(Co, device=dev); gy = torch.randn(B, Co, H, W, device=dev); a = (3, 3, 1, 1, 1, 1, 1, 1, 1)
for _ in range(3):
    _ext.dcn_v2_backward(x, w, b, off, m, gy, *a, precision=prec); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(10): _ext.dcn_v2_backward(x, w, b, off, m, gy, *a, precision=prec)
e1.record(); torch.cuda.synchronize()
print('bwd %%.3f ms' %% (e0.elapsed_time(e1) / 10))
"""
geo = tuple(int(v) for v in sys.argv[1:5])
prec = sys.argv[5] if len(sys.argv) > 5 else "f32"
osc = float(sys.argv[6]) if len(sys.argv) > 6 else 0.5
shutil.copy(main, main + ".bak")
try:
    for f in [main + ".bak"] + sorted(glob.glob(os.path.join(R, "tools", "micro", "libdcd_*.so"))):
        if not f.endswith(".bak"):
            shutil.copy(f, main)
        out = subprocess.run([sys.executable, "-c", code % (R, geo, prec, osc)], capture_output=True, text=True)
        print(os.path.basename(f), geo, prec, osc, out.stdout.strip(), out.stderr.strip()[-200:] if out.returncode else "")
finally:
    shutil.copy(main + ".bak", main)
