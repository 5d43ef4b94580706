"""Pure-PyTorch restatement of DCNv2 forward (autograd supplies the backward).

TEST INFRASTRUCTURE.  A second, independent statement of the formula in
cuda/dcn_v2_im2col_cuda.cu:25-54,125-195 (CPU twin cpu/dcn_v2_im2col_cpu.cpp:27-56,127-196):
per tap k=(i,j): h = y*s-p+i*d+off[:,2k], w = x*s-p+j*d+off[:,2k+1]; the sample is zero unless
-1 < h < H and -1 < w < W; each of the 4 corners contributes only if it lies inside the image;
col = sample*mask[:,k]; out = einsum(W, col) + bias.  Used to cross-check the C oracle's
hand-written backward through autograd, on any dtype.  deformable_groups == 1 only.
"""
import torch


def dcn_v2_reference(x, offset, mask, weight, bias, stride=1, padding=1, dilation=1):
    B, C, H, W = x.shape
    Co, _, kh, kw = weight.shape
    Ho = (H + 2 * padding - (dilation * (kh - 1) + 1)) // stride + 1
    Wo = (W + 2 * padding - (dilation * (kw - 1) + 1)) // stride + 1
    ys = (torch.arange(Ho, dtype=x.dtype) * stride - padding).view(1, Ho, 1)
    xs = (torch.arange(Wo, dtype=x.dtype) * stride - padding).view(1, 1, Wo)
    xf = x.reshape(B, C, H * W)
    cols = []
    for i in range(kh):
        for j in range(kw):
            k = i * kw + j
            h = ys + i * dilation + offset[:, 2 * k]
            w = xs + j * dilation + offset[:, 2 * k + 1]
            valid = ((h > -1) & (w > -1) & (h < H) & (w < W)).to(x.dtype)
            h0, w0 = torch.floor(h).detach(), torch.floor(w).detach()
            lh, lw = h - h0, w - w0
            val = 0
            for dy, dx, wt in ((0, 0, (1 - lh) * (1 - lw)), (0, 1, (1 - lh) * lw),
                               (1, 0, lh * (1 - lw)), (1, 1, lh * lw)):
                hh, ww = h0 + dy, w0 + dx
                inside = ((hh >= 0) & (hh <= H - 1) & (ww >= 0) & (ww <= W - 1)).to(x.dtype)
                idx = (hh.clamp(0, H - 1) * W + ww.clamp(0, W - 1)).long().view(B, 1, Ho * Wo).expand(B, C, Ho * Wo)
                v = torch.gather(xf, 2, idx).view(B, C, Ho, Wo)
                val = val + v * (wt * inside).unsqueeze(1)
            cols.append(val * (valid * mask[:, k]).unsqueeze(1))
    col = torch.stack(cols, dim=2)  # B, C, kh*kw, Ho, Wo
    return torch.einsum("ock,bckhw->bohw", weight.reshape(Co, C, kh * kw), col) + bias.view(1, Co, 1, 1)
