"""Reads the shader-clock stamps of a -DSW_PROFILE build of the one-pass backward (tools/micro/libdcd_prof*.so)."""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
lib_path = sys.argv[1]
import dcd_amd._lib as L
L.LIB_PATH = lib_path
from dcd_amd import _ext
C, Co, H, W, B = 64, 64, 96, 320, 8
dev = torch.device("cuda:0")
x = torch.randn(B, C, H, W, device=dev); off = torch.randn(B, 18, H, W, device=dev) * 0.5
m = torch.sigmoid(torch.randn(B, 9, H, W, device=dev)); w = torch.randn(Co, C, 3, 3, device=dev) / 24.0
b = torch.zeros(Co, device=dev); gy = torch.randn(B, Co, H, W, device=dev)
a = (3, 3, 1, 1, 1, 1, 1, 1, 1)
for _ in range(3):
    _ext.dcn_v2_backward(x, w, b, off, m, gy, *a)
torch.cuda.synchronize()
lib = ctypes.CDLL(lib_path)
buf = (ctypes.c_ulonglong * 2048)()
assert lib.dcd_debug_sweep_profile(buf) == 0
v = list(buf)
for row in range(2, 6):
    s = v[row * 64: row * 64 + 64]
    base = s[0]
    print("row %d: state %d | taps %s | fixup..retire-start %d | retire %d | row total (next row start) %d" % (
        row, s[1] - s[0], [s[2 + t] - (s[1] if t == 0 else s[1 + t]) for t in range(9)] if s[2] else "-",
        s[12] - (s[10] if s[10] else s[1]), s[13] - s[12], v[(row + 1) * 64] - s[0]))
    if s[16]:
        print("   step stamps (front done) deltas:", [s[16 + k] - (s[1] if k == 0 else s[15 + k]) for k in range(36)])
