"""Which backward-data kernel MIOpen picks for DLA's stride-2 3x3 layers at bs 8, with the solver switches set (a) by this script
before `import torch` (MODE=py), (b) in the shell (MODE=sh: the caller exports them), (c) not at all (MODE=none).  Run under
rocprofv3 --kernel-trace --stats."""
import os, sys
if os.environ.get("MODE") == "py":
    for v in ("MIOPEN_DEBUG_3D_CONV_IMPLICIT_GEMM_HIP_BWD_XDLOPS", "MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_HIP_GROUP_BWD_XDLOPS", "MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_HIP_BWD_XDLOPS"):
        os.environ[v] = "0"
import torch
dev = torch.device("cuda:0")
for C, K, H, W in [(64, 128, 96, 320), (128, 256, 48, 160), (256, 512, 24, 80), (32, 64, 192, 640)]:
    x = torch.randn(8, C, H, W, device=dev, requires_grad=True)
    w = torch.randn(K, C, 3, 3, device=dev, requires_grad=True)
    y = torch.nn.functional.conv2d(x, w, stride=2, padding=1)
    g = torch.randn_like(y)
    for _ in range(2):
        torch.autograd.grad(y, [x, w], g, retain_graph=True)
torch.cuda.synchronize()
print("done", os.environ.get("MODE"))
