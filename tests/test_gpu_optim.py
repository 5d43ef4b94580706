"""csrc/optim.hip through `ClipAdamW.clip_and_step` against the library calls it replaces (torch.nn.utils.clip_grad_norm_ semantics as
`trainer.clip_grad_norm`, the non-finite guard, torch.optim.AdamW fused + capturable).  PARITY with the library at rounding level: the
step is `DGDE/engine/trainer.py:144-147` (clip at 15, AdamW of `DGDE/solver/__init__.py:37`)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(64, 64, 3, 3), (256,), (27, 64, 3, 3), (1,), (3, 5, 7), (2048,), (2049,), (512, 256, 3, 3), (4097, 3), (16, 3, 7, 7)]


def _pair(cuda, seed, n_extra=0, lr=3e-4, blr=6e-4, wd=1e-5):
    from dcd_amd.engine import trainer
    g = torch.Generator().manual_seed(seed)
    shapes = SHAPES + [(33,)] * n_extra
    make = lambda: [torch.nn.Parameter(torch.randn(*s, generator=torch.Generator().manual_seed(seed + i)).to(cuda)) for i, s in enumerate(shapes)]
    pa, pb = make(), make()
    def opt(cls, ps):
        w = [p for p in ps if p.dim() > 1]
        b = [p for p in ps if p.dim() <= 1]
        groups = [{"params": w, "lr": torch.tensor(lr, device=cuda)}, {"params": b, "lr": torch.tensor(blr, device=cuda)}]
        return cls(groups, lr=lr, weight_decay=wd, betas=(0.9, 0.99), fused=True, capturable=True)
    return pa, pb, opt(trainer.ClipAdamW, pa), opt(torch.optim.AdamW, pb), g


def _set_grads(pa, pb, g, scale, skip=()):
    for i, (a, b) in enumerate(zip(pa, pb)):
        if i in skip:
            a.grad = b.grad = None
            continue
        gr = (torch.randn(a.shape, generator=g) * scale).to(a.device)
        a.grad, b.grad = gr.clone(), gr.clone()


def _library_step(trainer, opt, params, clip):
    trainer.guard_nonfinite_step(opt, trainer.clip_grad_norm(params, clip))
    opt.step()


@pytest.mark.parametrize("scale,clip", [(1.0, 15.0), (1e-3, 15.0), (30.0, 15.0), (1.0, 1e9)])
def test_clip_and_step_equals_the_library_calls(cuda, scale, clip):
    """Six steps from the same state and gradients: norm, clipped gradients, parameters, both moments and the step counters -- with the
    clip active (large gradients), inactive (small), and parameters that get no gradient in some steps (they keep their own step
    count, so their bias corrections differ from the others')."""
    from dcd_amd.engine import trainer
    pa, pb, oa, ob, g = _pair(cuda, 1)
    assert oa.own_kernels_ok()
    gmax = [0.0] * len(pa)
    for it in range(6):
        skip = (3, 5) if it in (1, 2) else ()
        _set_grads(pa, pb, g, scale, skip)
        total = oa.clip_and_step(clip)
        gmax = [max(m, 0.0 if a.grad is None else a.grad.abs().max().item()) for m, a in zip(gmax, pa)]
        ref_total = trainer.clip_grad_norm(pb, clip)
        trainer.guard_nonfinite_step(ob, ref_total)
        ob.step()
        assert abs(float(total) - float(ref_total)) <= 2e-6 * float(ref_total)
        for i, (a, b) in enumerate(zip(pa, pb)):
            if i in skip:
                assert a.grad is None
                continue
            assert torch.allclose(a.grad, b.grad, rtol=2e-6, atol=0), "clipped gradient %d at step %d" % (i, it)
    for i, (a, b) in enumerate(zip(pa, pb)):
        sa, sb = oa.state[a], ob.state[b]
        assert float(sa["step"]) == float(sb["step"]) == (4.0 if i in (3, 5) else 6.0)
        # a step moves a parameter by ~lr: 2e-6 of the VALUE after six steps is 1e-2 of one update's rounding headroom
        assert torch.allclose(a, b, rtol=2e-6, atol=1e-8), "parameter %d: %.3e" % (i, (a - b).abs().max().item())
        # the moments are sums of terms of the (clipped) gradients' size that may cancel: rounding of the TERMS
        assert (sa["exp_avg"] - sb["exp_avg"]).abs().max().item() <= 2e-7 * gmax[i], i
        assert (sa["exp_avg_sq"] - sb["exp_avg_sq"]).abs().max().item() <= 2e-7 * gmax[i] ** 2, i


def test_single_step_update_matches_to_rounding(cuda):
    """The UPDATE itself (parameter after minus before), where a parameter-level tolerance would hide a wrong step size: first step
    (bias corrections 0.1 / 0.01) and a later one, relative to the largest update."""
    from dcd_amd.engine import trainer
    pa, pb, oa, ob, g = _pair(cuda, 2, wd=0.0)
    with torch.no_grad():                                        # small parameters: their own rounding (1 ulp) far below an update (~3e-4)
        for a, b in zip(pa, pb):
            a.mul_(1e-3)
            b.mul_(1e-3)
    for it in range(3):
        _set_grads(pa, pb, g, 1.0)
        before = [a.detach().clone() for a in pa]
        oa.clip_and_step(15.0)
        _library_step(trainer, ob, pb, 15.0)
        for a, b, a0 in zip(pa, pb, before):
            ua, ub = a - a0, b - a0
            assert (ua - ub).abs().max().item() <= 2e-4 * ub.abs().max().item() + 1e-12, (it, tuple(a.shape))


def test_non_finite_norm_freezes_everything(cuda):
    from dcd_amd.engine import trainer
    pa, pb, oa, ob, g = _pair(cuda, 3)
    _set_grads(pa, pb, g, 1.0)
    oa.clip_and_step(15.0)
    snap = [(a.detach().clone(), oa.state[a]["exp_avg"].clone(), oa.state[a]["exp_avg_sq"].clone(), float(oa.state[a]["step"])) for a in pa]
    _set_grads(pa, pb, g, 1.0)
    pa[2].grad[0, 0, 0, 0] = float("nan")
    total = oa.clip_and_step(15.0)
    assert not np.isfinite(float(total))
    for a, (p0, m0, v0, s0) in zip(pa, snap):
        assert torch.equal(a, p0) and torch.equal(oa.state[a]["exp_avg"], m0) and torch.equal(oa.state[a]["exp_avg_sq"], v0)
        assert float(oa.state[a]["step"]) == s0
    _set_grads(pa, pb, g, 1.0)
    oa.clip_and_step(15.0)                                       # and the next finite step moves again
    assert not torch.equal(pa[0], snap[0][0]) and float(oa.state[pa[0]]["step"]) == 2.0


def test_many_tensors_span_several_launches(cuda):
    """More tensors than one launch's argument table holds (64), an element count of one among them."""
    from dcd_amd.engine import trainer
    pa, pb, oa, ob, g = _pair(cuda, 4, n_extra=150)
    for it in range(2):
        _set_grads(pa, pb, g, 5.0)
        ta = oa.clip_and_step(15.0)
        tb = trainer.clip_grad_norm(pb, 15.0)
        trainer.guard_nonfinite_step(ob, tb)
        ob.step()
        assert abs(float(ta) - float(tb)) <= 2e-6 * float(tb)
    for a, b in zip(pa, pb):
        assert torch.allclose(a, b, rtol=2e-6, atol=1e-8)


def test_clip_and_step_inside_a_captured_graph(cuda):
    """The pointer tables travel in the kernel arguments: a replayed graph needs nothing from the host, and repeats the eager steps."""
    from dcd_amd.engine import trainer
    pa, pb, oa, ob, g = _pair(cuda, 5)
    _set_grads(pa, pb, g, 1.0)
    static = [a.grad for a in pa]
    oa.clip_and_step(15.0)                                       # state tensors exist before the capture
    _library_step(trainer, ob, pb, 15.0)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        total = oa.clip_and_step(15.0)
    for it in range(3):
        for s, b in zip(static, pb):
            gr = (torch.randn(s.shape, generator=g) * 20.0).to(cuda)
            s.copy_(gr)
            b.grad = gr.clone()
        graph.replay()
        ref = trainer.clip_grad_norm(pb, 15.0)
        trainer.guard_nonfinite_step(ob, ref)
        ob.step()
        assert abs(float(total) - float(ref)) <= 2e-6 * float(ref)
    # the capture itself did not execute: 1 eager + 3 replays = 4 steps on both sides
    assert float(oa.state[pa[0]]["step"]) == float(ob.state[pb[0]]["step"]) == 4.0
    for a, b in zip(pa, pb):
        assert torch.allclose(a, b, rtol=2e-6, atol=1e-8)


def test_argument_checks(cuda):
    import ctypes
    from dcd_amd import _lib
    L = _lib.lib()
    x = torch.zeros(8, device=cuda)
    n = (ctypes.c_int64 * 1)(8)
    ptr = (ctypes.c_void_p * 1)(x.data_ptr())
    scal = torch.zeros(4, device=cuda)
    nb = L.dcd_clip_adamw_workspace_bytes(1, n)
    ws = torch.empty(nb, dtype=torch.uint8, device=cuda)
    st = _lib.stream_of(x)
    assert L.dcd_clip_grad_norm_scalars(st, 1, ptr, n, 1.0, ws.data_ptr(), nb, scal.data_ptr()) == 0
    assert L.dcd_clip_grad_norm_scalars(st, 1, ptr, n, 1.0, ws.data_ptr(), nb - 1, scal.data_ptr()) == 2
    assert L.dcd_clip_grad_norm_scalars(st, 1, ptr, n, 1.0, None, nb, scal.data_ptr()) == 1
    assert L.dcd_adamw_apply(st, 1, ptr, ptr, ptr, ptr, None, n, scal.data_ptr(), 0.9, 0.99, 1e-8, 0.0, scal.data_ptr()) == 1
    assert L.dcd_clip_adamw_workspace_bytes(1, (ctypes.c_int64 * 1)(1 << 31)) == 0
