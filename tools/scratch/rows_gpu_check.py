"""Column sums of the loss rows: kernel vs op by op on the GPU (debug aid)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
os.environ["DCD_LOSS_ROWS"] = "0"
import test_host_rows as T
from test_host_golden import small_cfg
from dcd_amd.model.head.detector_loss import Loss_Computation
host_loss, tv, pois = T._rows_inputs(False, False)
dev = torch.device("cuda:0")
tvd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in tv.items()}
l0 = Loss_Computation(small_cfg("cuda:0"))
os.environ["DCD_LOSS_ROWS"] = "1"
l1 = Loss_Computation(small_cfg("cuda:0"))
p0 = pois.to(dev).requires_grad_()
S0, ix = l0._rows({'reg_pois': p0, 'reg': None}, tvd)
p1 = pois.to(dev).requires_grad_()
S1, ix1 = l1._fused_rows({'reg_pois': p1}, tvd, 1.0)
names = {v: k for k, v in ix.items()}
for c in range(25):
    print("%-12s %14.6f %14.6f" % (names[c], float(S0[c]), float(S1[c])))
from dcd_amd import ops
orig = ops._lib.lib().dcd_edge_depth_forward
import ctypes
def spy(*args):
    st = orig(*args)
    torch.cuda.synchronize()
    print("solver args N K topk", args[6:9])
    return st
class L:
    def __getattr__(self, n):
        return spy if n == "dcd_edge_depth_forward" else getattr(ops._lib._LIB, n)
real_lib = ops._lib.lib
ops._lib.lib = lambda: L()
saved = {}
orig_save = torch.autograd.function.FunctionCtx.save_for_backward
S1, _ = l1._fused_rows({'reg_pois': p1}, tvd, 1.0)
fn = S1.grad_fn
t = fn.saved_tensors
names = "pois dim_mean kps_pred kps3d_pred rot P_rows depth pmask idx".split()
for n, x in zip(names, t[:9]):
    print(n, tuple(x.shape), x.dtype, float(x.float().sum()), x.data_ptr() % 256)
