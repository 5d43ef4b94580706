"""Advisor r2 (low): the trunk variances are w^T G w / n - mean^2 with G accumulated in fp32 pieces.  How large is the error of the
batch variance when the input has a strong DC component?  Compares against the dense convolution's statistics in float64."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from torch.nn import functional as F
from dcd_amd.model.head import trunk_moments as TM

dev = torch.device("cuda:0")
torch.manual_seed(0)
B, C, H, W, O = 8, 64, 96, 320, 256
w = (torch.randn(O, C, 3, 3) / (C * 9) ** 0.5).to(dev)
for ratio in (0.0, 1.0, 3.0, 10.0, 30.0):
    x = torch.relu(torch.randn(B, C, H, W, device=dev)) + ratio * 0.58          # relu(N(0,1)) has std 0.58
    y = F.conv2d(x.double(), w.double(), padding=1)
    var_ref = y.var(dim=(0, 2, 3), unbiased=False)
    mean_ref = y.mean(dim=(0, 2, 3))
    for mode in ("shift", "bmm"):
        os.environ["DCD_TRUNK_GRAM"] = mode
        S1, G, _ = TM.patch_moments(x)
        Wd = w.reshape(O, -1).double()
        n = B * H * W
        mean = (Wd @ S1) / n
        var = ((Wd @ G) * Wd).sum(-1) / n - mean * mean
        rel = ((var - var_ref).abs() / var_ref).max().item()
        print("input mean/std %5.1f  %-5s  max rel. error of the batch variance %.2e   (|mean|/std of the outputs up to %.1f)" % (
            x.mean().item() / x.std().item(), mode, rel, (mean_ref.abs() / var_ref.sqrt()).max().item()))
