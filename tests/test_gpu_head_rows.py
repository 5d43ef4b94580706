"""ops.head_rows (csrc/heads.hip, dcd_head_rows_*): the regression heads' 1x1 output layers at listed rows in one launch against
one F.linear per head, values and every gradient; and the predictor through it against the predictor through F.linear."""
import os
import sys

import numpy as np
import pytest
import torch
from torch.nn import functional as F

sys.path.insert(0, os.path.dirname(__file__))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("R,K", [(50, 256), (320, 256), (7, 64)])
def test_head_rows_equal_one_linear_per_head(cuda, R, K):
    from dcd_amd import ops
    g = torch.Generator().manual_seed(4)
    trunk_of, outs = [0, 0, 1, 2, 2, 2], [4, 2, 20, 3, 1, 219]
    T = 3
    feat = torch.randn(T, R, K, generator=g).to(cuda).requires_grad_()
    ws = [(torch.randn(o, K, 1, 1, generator=g) * 0.1).to(cuda).requires_grad_() for o in outs]
    bs = [torch.randn(o, generator=g).to(cuda).requires_grad_() for o in outs]
    y = ops.head_rows(feat, trunk_of, ws, bs)
    ref = torch.cat([F.linear(feat[t], w.view(w.shape[0], K), b) for t, w, b in zip(trunk_of, ws, bs)], dim=1)
    assert y.shape == ref.shape == (R, sum(outs))
    scale = ref.abs().max().item()
    assert (y - ref).abs().max().item() <= 2e-6 * scale
    gy = torch.randn(R, sum(outs), generator=g).to(cuda)
    got = torch.autograd.grad(y, [feat] + ws + bs, gy)
    want = torch.autograd.grad(ref, [feat] + ws + bs, gy)
    for a, b in zip(got, want):
        assert a.shape == b.shape
        assert (a - b).abs().max().item() <= 5e-6 * max(b.abs().max().item(), 1e-3)
    # reproducible: fixed summation order in all three kernels
    again = torch.autograd.grad(ops.head_rows(feat, trunk_of, ws, bs), [feat] + ws + bs, gy)
    assert all(torch.equal(a, b) for a, b in zip(got, again))


def test_head_rows_reject_heads_out_of_trunk_order(cuda):
    from dcd_amd import ops
    from dcd_amd._lib import DcdHipError
    feat = torch.zeros(2, 4, 64, device=cuda)
    ws = [torch.zeros(3, 64, device=cuda), torch.zeros(2, 64, device=cuda)]
    with pytest.raises(DcdHipError):
        ops.head_rows(feat, [1, 0], ws, [None, None])


def test_predictor_through_head_rows_equals_predictor_through_linears(cuda, monkeypatch):
    import golden_inputs as gi
    from test_host_golden import small_cfg
    from dcd_amd.model.head import detector_predictor as dp
    torch.manual_seed(0)
    cfg = small_cfg(str(cuda))
    pred = dp.make_predictor(cfg, 64).to(cuda).train()
    gi.name_hashed_init(pred)
    _, targets = gi.model_inputs()
    targets = [t.to(cuda) for t in targets]
    feats = torch.randn(2, 64, 24, 80, generator=torch.Generator().manual_seed(1)).to(cuda)
    res = {}
    for mode in (True, False):
        monkeypatch.setattr(dp, "_HEAD_ROWS", mode)
        gi.name_hashed_init(pred)                       # also resets the BatchNorm running estimates
        pred.zero_grad()
        x = feats.clone().requires_grad_()
        out = pred(x, targets)
        w = torch.randn(out['reg_pois'].shape, generator=torch.Generator().manual_seed(2)).to(cuda)
        ((out['reg_pois'] * w).sum() + out['cls'].sum()).backward()
        res[mode] = (out['reg_pois'].detach(), out['cls'].detach(), x.grad, {n: p.grad.clone() for n, p in pred.named_parameters()
                                                                            if p.grad is not None})
    a, b = res[True], res[False]
    assert (a[0] - b[0]).abs().max().item() <= 2e-6 * b[0].abs().max().item()
    assert (a[1] - b[1]).abs().max().item() <= 1e-6          # same kernels on both sides; the border scatter adds with atomics
    assert (a[2] - b[2]).abs().max().item() <= 1e-5 * b[2].abs().max().item()
    assert a[3].keys() == b[3].keys()
    top = max(v.abs().max().item() for v in b[3].values())
    for n in a[3]:                  # biases in front of a BatchNorm have a zero gradient: rounding noise on both sides
        assert (a[3][n] - b[3][n]).abs().max().item() <= 1e-5 * max(b[3][n].abs().max().item(), 1e-3 * top), n
