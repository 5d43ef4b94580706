"""dcd_amd/model/head/trunk_moments.py: the regression trunks (3x3 conv -> BatchNorm -> ReLU) evaluated at listed positions from
the Gram matrix of the input's 3x3 patches must equal the dense evaluation -- outputs, gradients w.r.t. the shared input / conv
weights / BN parameters, and the running estimates.  float64 on the host: round-off out of the picture, tolerance 1e-9.
(The model-level equality with the reference is in tests/test_host_golden.py / test_gpu_golden.py.)"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn
from torch.nn import functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_trunks(n, cin, cout, seed):
    from dcd_amd.model.layers.conv import Conv2d
    from dcd_amd.model.layers.norm import BatchNorm2d
    g = torch.Generator().manual_seed(seed)
    trunks = nn.ModuleList()
    for _ in range(n):
        t = nn.Sequential(Conv2d(cin, cout, kernel_size=3, padding=1, bias=False), BatchNorm2d(cout, fuse_relu=True), nn.Identity())
        with torch.no_grad():
            t[0].weight.copy_(torch.randn(t[0].weight.shape, generator=g) * 0.2)
            t[1].weight.copy_(torch.rand(cout, generator=g) + 0.5)
            t[1].bias.copy_(torch.randn(cout, generator=g) * 0.1)
        trunks.append(t)
    return trunks.double().train()


def dense_reference(x, trunks, centers, extra):
    outs = []
    for i, t in enumerate(trunks):
        y = F.conv2d(x, t[0].weight, None, 1, 1)
        z = torch.relu(F.batch_norm(y, None, None, t[1].weight, t[1].bias, True, 0.0, t[1].eps))
        pos = centers if extra is None or extra[0] != i else torch.cat((centers, extra[1]), 1)
        b, c = z.shape[0], z.shape[1]
        outs.append(z.flatten(2).gather(2, pos.unsqueeze(1).expand(b, c, pos.shape[1])).transpose(1, 2))
    return outs


def test_trunks_at_equal_dense_conv_bn_relu():
    from dcd_amd.model.head import trunk_moments as TM
    torch.manual_seed(0)
    B, C, H, W, O = 3, 5, 9, 14, 7
    trunks = make_trunks(3, C, O, 1)
    x0 = torch.randn(B, C, H, W, dtype=torch.float64)
    centers = torch.randint(0, H * W, (B, 6))
    centers[0, 0], centers[1, 1], centers[2, 2] = 0, H * W - 1, W - 1            # corners: the zero padding matters
    extra = (1, torch.randint(0, H * W, (B, 11)))
    wts = [torch.randn(B, 6 + (11 if i == 1 else 0), O, dtype=torch.float64) for i in range(3)]
    res = []
    for mode in ("moments", "dense"):
        for t in trunks:
            t[1].reset_running_stats()
        trunks.zero_grad()
        x = x0.clone().requires_grad_()
        assert TM.usable(trunks, x)
        outs = TM.trunks_at(x, trunks, centers, extra) if mode == "moments" else dense_reference(x, trunks, centers, extra)
        sum((o * w).sum() for o, w in zip(outs, wts)).backward()
        res.append(([o.detach() for o in outs], x.grad.clone(), [p.grad.clone() for p in trunks.parameters()],
                    [t[1].running_mean.clone() for t in trunks], [t[1].running_var.clone() for t in trunks],
                    [int(t[1].num_batches_tracked) for t in trunks]))
    (o0, gx0, gp0, rm0, rv0, nb0), (o1, gx1, gp1, rm1, rv1, nb1) = res
    for a, b in zip(o0, o1):
        assert (a - b).abs().max().item() <= 1e-9
    assert (gx0 - gx1).abs().max().item() <= 1e-9 * max(gx1.abs().max().item(), 1)
    for a, b in zip(gp0, gp1):
        assert (a - b).abs().max().item() <= 1e-9 * max(b.abs().max().item(), 1)
    assert nb0 == [1, 1, 1]
    # running estimates: momentum 0.1, unbiased variance -- against torch's own BatchNorm update
    for i, t in enumerate(trunks):
        bn = nn.BatchNorm2d(O).double().train()
        bn(F.conv2d(x0, t[0].weight, None, 1, 1))
        assert (rm0[i] - bn.running_mean).abs().max().item() <= 1e-12
        assert (rv0[i] - bn.running_var).abs().max().item() <= 1e-12


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _sync_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    from dcd_amd.model.head import trunk_moments as TM
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    B, C, H, W, O = 4, 5, 8, 10, 6
    x0 = torch.randn(B, C, H, W, dtype=torch.float64)
    centers = torch.randint(0, H * W, (B, 5))
    wts = torch.randn(B, 5, O, dtype=torch.float64)
    # single process, full batch
    full = make_trunks(2, C, O, 3)
    xf = x0.clone().requires_grad_()
    sum((o * wts).sum() for o in TM.trunks_at(xf, full, centers)).backward()
    # two ranks, half the batch each, per-channel sums exchanged over gloo
    half = make_trunks(2, C, O, 3)
    for t in half:
        t[1].sync_group = dist.group.WORLD
    sl = slice(2 * rank, 2 * rank + 2)
    xs = x0[sl].clone().requires_grad_()
    sum((o * wts[sl]).sum() for o in TM.trunks_at(xs, half, centers[sl])).backward()
    gw = [p.grad.clone() for p in half.parameters()]
    for g_ in gw:
        dist.all_reduce(g_)                    # the full-batch gradient is the SUM of the local ones (DDP would average)
    ok = torch.allclose(xs.grad, xf.grad[sl], atol=1e-10)
    ok = ok and all(torch.allclose(a, p.grad, atol=1e-9) for a, p in zip(gw, full.parameters()))
    ok = ok and all(torch.allclose(a[1].running_var, b[1].running_var, atol=1e-12) for a, b in zip(half, full))
    torch.save({"ok": bool(ok)}, os.path.join(out_dir, "tm%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_trunks_at_with_synchronised_statistics_two_ranks(tmp_path):
    """MODEL.USE_SYNC_BN: the per-channel sums (not the Gram matrix) are all-reduced, differentiably; two ranks with half the
    batch each must reproduce the single-process full-batch gradients and running estimates."""
    port = _free_port()
    mp.spawn(_sync_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        assert torch.load(os.path.join(str(tmp_path), "tm%d.pt" % r))["ok"]


def test_affine_relu_and_patch_linear_nodes_equal_the_plain_tensor_operations():
    """`_AffineRelu` (relu(y scale + shift) with the scale / shift gradients summed in short steps, trunk_moments._sum_rows) and
    `_PatchLinear` (einsum('bkm,ok->bmo') as batched products) against the plain formulas differentiated by autograd, float64."""
    import torch
    from dcd_amd.model.head import trunk_moments as TM
    torch.manual_seed(0)
    # 67, 2 * 67 and 16 * 67: sizes whose only divisor up to 64 is 1 or 2 -- the first form of _sum_rows never returned for them
    for T, B, M, O in ((3, 2, 40, 8), (1, 2, 832, 4), (2, 3, 7, 5), (1, 1, 67, 2), (2, 2, 134, 3), (1, 2, 1072, 2)):
        y = torch.randn(T, B, M, O, dtype=torch.float64, requires_grad=True)
        sc = torch.randn(T, O, dtype=torch.float64, requires_grad=True)
        sh = torch.randn(T, O, dtype=torch.float64, requires_grad=True)
        g = torch.randn(T, B, M, O, dtype=torch.float64)
        ref = torch.relu(y * sc.view(T, 1, 1, O) + sh.view(T, 1, 1, O))
        ref.backward(g)
        want = (ref.detach().clone(), y.grad.clone(), sc.grad.clone(), sh.grad.clone())
        y.grad = sc.grad = sh.grad = None
        out = TM._AffineRelu.apply(y, sc, sh)
        out.backward(g)
        for a, b in zip((out.detach(), y.grad, sc.grad, sh.grad), want):
            assert torch.allclose(a, b, atol=1e-12)
        assert torch.allclose(TM._sum_rows(g), g.sum((1, 2)), atol=1e-12)
    X = torch.randn(2, 18, 11, dtype=torch.float64, requires_grad=True)
    W = torch.randn(5, 18, dtype=torch.float64, requires_grad=True)
    g = torch.randn(2, 11, 5, dtype=torch.float64)
    ref = torch.einsum('bkm,ok->bmo', X, W)
    ref.backward(g)
    want = (ref.detach().clone(), X.grad.clone(), W.grad.clone())
    X.grad = W.grad = None
    out = TM._PatchLinear.apply(X, W)
    out.backward(g)
    for a, b in zip((out.detach(), X.grad, W.grad), want):
        assert torch.allclose(a, b, atol=1e-12)
