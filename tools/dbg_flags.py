import sys, os
sys.path.insert(0, "/root/repo")
import torch
from dcd_amd import _lib
L = _lib.lib()
dev = torch.device("cuda:0")
B, C, Co, H, W = 8, 128, 128, 48, 160
osc = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
x = torch.randn(B, C, H, W, device=dev)
off = torch.randn(B, 18, H, W, device=dev) * osc
m = torch.sigmoid(torch.randn(B, 9, H, W, device=dev))
w = torch.randn(Co, C, 3, 3, device=dev) / (C * 9) ** 0.5
b = torch.zeros(Co, device=dev)
geom = (B, C, H, W, Co, 3, 3, 1, 1, 1, 1, 1, 1, 1)
n = L.dcd_dcn_v2_workspace_bytes(*geom)
ws = torch.zeros(n, dtype=torch.uint8, device=dev)
out = torch.empty(B, Co, H, W, device=dev)
st = L.dcd_dcn_v2_forward(torch.cuda.current_stream().cuda_stream, x.data_ptr(), w.data_ptr(), b.data_ptr(), off.data_ptr(), m.data_ptr(), out.data_ptr(), *geom, 0, ws.data_ptr(), n)
torch.cuda.synchronize()
nw = 9 * C * Co          # Kp*Cop with cpgp = C (multiple of 32), Cop = Co
flags = ws[2 * nw * 4: 2 * nw * 4 + 8 * 5 * 12 + 64]
print("status", st, "flag bytes set:", int(flags.sum().item()), "of", 8 * 5 * 12, "(TR=4) or", 8 * 5 * 6, "(TR=8)")
print(flags[:64].tolist())
