"""Fused BN (+residual) (+ReLU) HIP kernels (csrc/norm.hip) against the stock torch ops the reference uses
(F.batch_norm + add + relu and their autograd), evaluated in fp64 on the same inputs.
Tolerance: 2e-5 of each tensor's max magnitude (fp32 arithmetic; the task's bound for floating point is 1e-3)."""
import pytest
import torch
from torch.nn import functional as F

pytestmark = pytest.mark.gpu

TOL = 2e-5

CASES = [  # B, C, H, W, residual, relu
    (2, 16, 48, 160, False, True),
    (8, 64, 24, 80, True, True),
    (2, 5, 7, 9, True, True),          # HW not a multiple of 4 -> scalar path
    (1, 1, 3, 5, False, False),
    (3, 33, 12, 40, True, False),
    (2, 512, 12, 40, False, True),     # small channels: one launch per direction (B*HW <= 16384, C >= 64)
    (8, 256, 24, 80, True, False),     # ... at the size of DLA level 4 (15360 values per channel)
    (1, 64, 128, 128, False, True),    # ... exactly 16384 values per channel
    (1, 64, 128, 132, True, True),     # just above: two-launch path
    (1, 16, 384, 1280, False, True),   # the full-resolution base layer (one image)
]


def _reference(x, res, w, b, relu, gy, eps=1e-5):
    xd = x.double().requires_grad_()
    rd = None if res is None else res.double().requires_grad_()
    wd, bd = w.double().requires_grad_(), b.double().requires_grad_()
    rm, rv = torch.zeros_like(wd), torch.ones_like(wd)
    y = F.batch_norm(xd, rm, rv, wd, bd, True, 0.1, eps)
    if rd is not None:
        y = y + rd
    if relu:
        y = F.relu(y)
    y.backward(gy.double())
    return y.detach(), xd.grad, None if rd is None else rd.grad, wd.grad, bd.grad, rm, rv


def _close(a, ref, what, tol=TOL):
    err = (a.double() - ref).abs().max().item()
    scale = max(ref.abs().max().item(), 1e-6)
    assert err <= tol * scale, "%s: max abs err %.3e vs scale %.3e" % (what, err, scale)


@pytest.mark.parametrize("B,C,H,W,use_res,relu", CASES)
def test_bn_act_train_matches_stock_ops(cuda, B, C, H, W, use_res, relu):
    from dcd_amd.model.layers.norm import BatchNorm2d
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + C)
    x = (torch.randn(B, C, H, W, generator=g) * 1.7 + 0.8).to(cuda)
    res = torch.randn(B, C, H, W, generator=g).to(cuda) if use_res else None
    gy = torch.randn(B, C, H, W, generator=g).to(cuda)
    bn = BatchNorm2d(C, fuse_relu=relu).to(cuda).train()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g) * 0.3)
    xg = x.clone().requires_grad_()
    rg = None if res is None else res.clone().requires_grad_()
    y = bn(xg, rg)
    y.backward(gy)
    ry, rgx, rgr, rgw, rgb, rm, rv = _reference(x, res, bn.weight.detach(), bn.bias.detach(), relu, gy)
    _close(y, ry, "y")
    _close(xg.grad, rgx, "grad_x")
    if use_res:
        _close(rg.grad, rgr, "grad_residual", 1e-7)
    _close(bn.weight.grad, rgw, "grad_weight")
    _close(bn.bias.grad, rgb, "grad_bias")
    _close(bn.running_mean, rm, "running_mean")
    _close(bn.running_var, rv, "running_var")
    assert int(bn.num_batches_tracked) == 1


def test_bn_act_is_deterministic_and_eval_matches(cuda):
    from dcd_amd.model.layers.norm import BatchNorm2d
    g = torch.Generator(device="cpu").manual_seed(3)
    x = torch.randn(4, 32, 24, 80, generator=g).to(cuda)
    res = torch.randn(4, 32, 24, 80, generator=g).to(cuda)
    bn = BatchNorm2d(32, fuse_relu=True).to(cuda).train()
    outs = []
    for _ in range(2):
        bn.reset_running_stats()
        xg = x.clone().requires_grad_()
        y = bn(xg, res)
        y.square().sum().backward()
        outs.append((y.detach().clone(), xg.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])   # fixed-order reductions
    bn.eval()
    with torch.no_grad():
        ye = bn(x, res)
        ref = F.relu(F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.1, bn.eps) + res)
    _close(ye, ref.double(), "eval y")


def test_bn_sync_group_world_size_1_equals_local(cuda, tmp_path):
    """The synchronised path (all-reduce of the fp64 sums over RCCL) with a one-rank group must equal the local path."""
    import torch.distributed as dist
    from dcd_amd.model.layers.norm import BatchNorm2d
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="file://" + str(tmp_path / "pg"), rank=0, world_size=1)
        created = True
    try:
        g = torch.Generator(device="cpu").manual_seed(11)
        # two sizes: one slice per channel (sums straight from the statistics kernels, no combine launch) and the two-stage form
        for shape in ((2, 8, 12, 40), (2, 8, 160, 128)):
            x = torch.randn(*shape, generator=g).to(cuda)
            a, b = BatchNorm2d(8, fuse_relu=True).to(cuda).train(), BatchNorm2d(8, fuse_relu=True).to(cuda).train()
            b.sync_group = dist.group.WORLD
            xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
            ya, yb = a(xa), b(xb)
            w = torch.randn(*shape, generator=g).to(cuda)
            (ya * w).sum().backward()
            (yb * w).sum().backward()
            # same sums up to their order (the local path reduces in one kernel, the synchronised one in slices)
            _close(yb, ya.double(), "forward %s" % (shape,), 1e-6)
            _close(xb.grad, xa.grad.double(), "grad_input %s" % (shape,), 1e-5)
            _close(b.weight.grad, a.weight.grad.double(), "grad_weight %s" % (shape,), 1e-5)
            _close(b.bias.grad, a.bias.grad.double(), "grad_bias %s" % (shape,), 1e-5)
            _close(b.running_var, a.running_var.double(), "running_var %s" % (shape,), 1e-6)
    finally:
        if created:
            dist.destroy_process_group()


def test_bn_bad_arguments_raise(cuda):
    from dcd_amd import _lib, ops
    x = torch.randn(2, 4, 3, 3)
    with pytest.raises(_lib.DcdHipError):
        ops.batch_norm_act(x, None, None, None, None, None, None, 0.1, 1e-5, True)      # CPU tensor: no CPU path in ops
    L = _lib.lib()
    xc = x.to(cuda)
    assert L.dcd_bn_stats(_lib.stream_of(xc), xc.data_ptr(), 2, 4, 9, None, None, 0) == 1
    st = torch.empty(8, dtype=torch.float64, device=cuda)
    assert L.dcd_bn_stats(_lib.stream_of(xc), xc.data_ptr(), 2, 4, 9, st.data_ptr(), None, 0) == 2


@pytest.mark.parametrize("B,C,H,W,N", [(2, 16, 12, 40, 7), (8, 256, 24, 80, 40), (1, 5, 7, 9, 3)])
def test_bn_relu_at_positions_matches_dense_then_gather(cuda, B, C, H, W, N):
    """BatchNorm2d.forward_at (BN + ReLU evaluated at listed positions, dense input gradient from sparse output gradients)
    against the stock ops in fp64: batch_norm -> relu -> gather, with repeated positions."""
    from dcd_amd.model.layers.norm import BatchNorm2d
    g = torch.Generator(device="cpu").manual_seed(B * 100 + C)
    x = (torch.randn(B, C, H, W, generator=g) * 1.3 + 0.4)
    pos = torch.randint(0, H * W, (B, N), generator=g)
    pos[:, -1] = pos[:, 0]                                           # a repeated position
    gout = torch.randn(B, N, C, generator=g)
    w0, b0 = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    xd, wd, bd = x.double().requires_grad_(), w0.double().requires_grad_(), b0.double().requires_grad_()
    rm, rv = torch.zeros(C, dtype=torch.float64), torch.ones(C, dtype=torch.float64)
    y = F.relu(F.batch_norm(xd, rm, rv, wd, bd, True, 0.1, 1e-5))
    ref = y.flatten(2).gather(2, pos.unsqueeze(1).expand(B, C, N)).transpose(1, 2)
    ref.backward(gout.double())
    bn = BatchNorm2d(C, fuse_relu=True).to(cuda).train()
    with torch.no_grad():
        bn.weight.copy_(w0)
        bn.bias.copy_(b0)
    xg = x.to(cuda).requires_grad_()
    out = bn.forward_at(xg, pos.to(cuda))
    out.backward(gout.to(cuda))
    _close(out.detach().cpu(), ref.detach(), "y_at")
    _close(xg.grad.cpu(), xd.grad, "grad_x")
    _close(bn.weight.grad.cpu(), wd.grad, "grad_weight")
    _close(bn.bias.grad.cpu(), bd.grad, "grad_bias")
    _close(bn.running_mean.cpu(), rm, "running_mean")
    _close(bn.running_var.cpu(), rv, "running_var")
    assert int(bn.num_batches_tracked) == 1


@pytest.mark.parametrize("shape", [(8, 27, 96, 320), (2, 27, 12, 40), (3, 5, 7, 9), (1, 300, 4, 4)])
def test_channel_sums_one_launch_equals_fp64_sum_and_repeats(cuda, shape):
    """Bias gradient of the DCN offset convolution (ops.channel_sums, one launch with a last-arriver combine): the fp64 sum rounded
    once, identical on every call (the arrival counters must be left at zero), also for two different tensors back to back."""
    from dcd_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(*shape, generator=g).to(cuda)
    y = (torch.randn(*shape, generator=g) * 3 + 1).to(cuda)
    ref_x = x.double().sum(dim=(0, 2, 3))
    ref_y = y.double().sum(dim=(0, 2, 3))
    first = ops.channel_sums(x)
    for _ in range(3):
        sx, sy = ops.channel_sums(x), ops.channel_sums(y)
        assert torch.equal(sx, first)
        assert (sx.double() - ref_x).abs().max().item() <= 1e-6 * max(ref_x.abs().max().item(), 1.0)
        assert (sy.double() - ref_y).abs().max().item() <= 1e-6 * max(ref_y.abs().max().item(), 1.0)


def test_channel_sums_on_two_streams_at_once(cuda):
    """Advisor r3: the arrival counters of the one-launch channel sums live in the call's workspace (one zeroed buffer per
    stream), not in a process-global array -- launches interleaved on two streams must not see each other's arrivals."""
    from dcd_amd import ops
    g = torch.Generator().manual_seed(4)
    xs = [torch.randn(8, 27, 96, 320, generator=g).to(cuda) for _ in range(2)]
    refs = [x.double().sum(dim=(0, 2, 3)) for x in xs]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    outs = [[], []]
    for _ in range(20):
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                outs[i].append(ops.channel_sums(xs[i]))
    torch.cuda.synchronize()
    for i in range(2):
        for o in outs[i]:
            assert torch.equal(o, outs[i][0])
            assert (o.double() - refs[i]).abs().max().item() <= 1e-6 * max(refs[i].abs().max().item(), 1.0)
