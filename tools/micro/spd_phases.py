"""Phase clocks of spd_diag_block (profile build of csrc/spd.hip: hipcc -DSPD_PROFILE -> tools/micro/libspd_prof.so)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libspd_prof.so"))
lib.dcd_spd_solve_workspace_bytes.restype = ctypes.c_size_t
lib.dcd_spd_solve_workspace_bytes.argtypes = [ctypes.c_int, ctypes.c_int]
lib.dcd_spd_solve.restype = ctypes.c_int
lib.dcd_spd_solve.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 3 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
dev = torch.device("cuda:0")
b, n = 8, 2628
A = torch.randn(b, n, n, device=dev)
S = A @ A.transpose(1, 2) / n + 0.05 * torch.eye(n, device=dev)
aug = torch.empty(b, n + 4, n, device=dev)
aug[:, :n] = S
aug[:, n] = torch.randn(b, n, device=dev)
nb = lib.dcd_spd_solve_workspace_bytes(b, n)
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
y = torch.empty(b, n, device=dev)
info = torch.zeros(64, dtype=torch.int32, device=dev)
st = lib.dcd_spd_solve(torch.cuda.current_stream().cuda_stream, aug.data_ptr(), y.data_ptr(), b, n, n + 4, info.data_ptr(), ws.data_ptr(), nb)
torch.cuda.synchronize()
c = info[8:13].tolist()
print("status", st, "cycles: load %d | cholesky %d | store L %d | inverse %d | store D %d  (2.4 GHz: %.1f us total)" % (*c, sum(c) / 2400.0))
