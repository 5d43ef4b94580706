import sys, os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
import torch
from dcd_amd import _ext
from test_gpu_dcn import make_case
cuda=torch.device('cuda:0')
x, w, b, off, m, gy = (t.to(cuda) for t in make_case(2, 64, 64, 24, 40, off_scale=0.25, seed=5))
off.clamp_(-0.9, 0.9)
a = (3, 3, 1, 1, 1, 1, 1, 1, 1)
outs=[_ext.dcn_v2_backward(x, w, b, off, m, gy, *a) for _ in range(3)]
torch.cuda.synchronize()
for k,name in enumerate(("gin","goff","gmask","gw","gb")):
    d=(outs[0][k]-outs[1][k]).abs()
    print(name, float(d.max()), int((d>0).sum()), d.numel())
d=(outs[0][0]-outs[1][0]).abs()
nz=(d>0).nonzero()
print(nz[:10].tolist())
