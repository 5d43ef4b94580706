"""The loss-row arithmetic of csrc/loss_rows_math.h, compiled for the HOST with a one-lane wave (tests/host/host_rows.cpp),
against the op-by-op evaluation of the same terms (Loss_Computation._rows on the patched CPU ops) and its autograd:
column sums and the gradient w.r.t. the head outputs.  The GPU build of the same header is checked in test_gpu_golden.py."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
import golden_inputs as gi  # noqa: E402
from test_host_golden import small_cfg  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def host_rows(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("host_rows") / "libhost_rows.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", os.path.join(HERE, "host", "host_rows.cpp"), "-o", out])
    return ctypes.CDLL(out)


def _rows_inputs(empty_image=False, mixed_flags=False, seed=0):
    from dcd_amd.model.head.detector_loss import Loss_Computation
    preds, targets = gi.loss_inputs()
    loss = Loss_Computation(small_cfg("cpu"))
    # (host tensors take the op-by-op rows whatever DCD_LOSS_ROWS says: the row kernel is chosen per call for device fp32 rows)
    _, tv = loss.prepare_targets(targets)
    if empty_image:                                          # image 0 owns no object: the calibration-rank quirk is exercised
        tv['reg_mask'] = tv['reg_mask'].clone()
        tv['reg_mask'][0] = 0
    if mixed_flags:
        # the fixture's objects are all visible, untruncated and found: flip flags object by object so that every masked
        # branch (truncated offset, invisible keypoint groups, invalid pairs, objects without dense keypoints, no
        # orientation, degenerate box) carries weight
        tv = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in tv.items()}
        slots = tv['reg_mask'].reshape(-1).nonzero().reshape(-1).tolist()
        B, M = tv['reg_mask'].shape

        def at(name, i):
            return tv[name].reshape(B * M, *tv[name].shape[2:])[slots[i]]
        at('trunc_mask', 0).fill_(1)
        at('trunc_mask', 3).fill_(1)
        at('keypoints_depth_mask', 1)[1] = 0
        at('keypoints_depth_mask', 2)[:] = 0
        at('extra_kpts_2d', 1)[::3, 2] = 0
        at('extra_kpts_2d', 4)[5:40, 2] = 0
        at('find_pcl', 2).fill_(False)
        at('ori_mask', 3).fill_(False)
        at('2d_bboxes' if '2d_bboxes' in tv else 'bboxes', 5)[2] = at('bboxes', 5)[0]
        at('keypoints', 0)[2:5, 2] = 0
    reg = torch.from_numpy(preds["reg"])
    rng = np.random.RandomState(seed)
    cx = tv["target_centers"].long()
    bidx = torch.arange(cx.shape[0]).view(-1, 1).expand(-1, cx.shape[1])
    pois = reg[bidx, :, cx[:, :, 1], cx[:, :, 0]].contiguous()          # B x M x 415, the heads at the object centres
    pois += torch.from_numpy(rng.normal(0, 0.05, tuple(pois.shape)).astype(np.float32))
    return loss, tv, pois


def _host_args(loss, tv, pois, keep):
    from dcd_amd import _lib, ops
    enc = loss.anno_encoder
    loss._rows_spec = None
    # build the spec exactly as the product does
    spec_holder = {}
    orig = ops.loss_rows
    ops.loss_rows = lambda p, spec, dim_mean, t: (spec_holder.update(spec=spec, dim_mean=dim_mean, tv=t), torch.zeros(25))[1]
    try:
        loss._fused_rows({'reg_pois': pois}, tv, 1.0)
    finally:
        ops.loss_rows = orig
    spec, dim_mean, tvc = spec_holder["spec"], spec_holder["dim_mean"], spec_holder["tv"]
    targets = tuple(ops._typed(tvc[name], dt) for name, dt in ops._ROWS_TARGETS)
    a = ops._LossRows._args(spec, pois, targets, dim_mean.contiguous())
    keep.extend(targets)
    keep.append(dim_mean)
    return a, spec


def _run_host(host_rows, loss, tv, pois, gsums):
    from oracle import torch_ops
    keep = []
    pois = pois.detach().contiguous()
    a, spec = _host_args(loss, tv, pois, keep)
    B, M, C = pois.shape
    BM, K, NP = B * M, spec["K"], spec["NP"]
    f32 = torch.float32
    kps = torch.zeros((2, BM, K, 2), dtype=f32)
    kps3d = torch.zeros((2, BM, K, 3), dtype=f32)
    rot = torch.zeros((2, BM), dtype=f32)
    P_rows = torch.zeros((2, BM, 3, 4), dtype=f32)
    kmask = torch.zeros((2, BM, K), dtype=torch.uint8)
    a.kps_pred, a.kps_tgt = kps[0].data_ptr(), kps[1].data_ptr()
    a.kps3d_pred, a.kps3d_tgt = kps3d[0].data_ptr(), kps3d[1].data_ptr()
    a.rot, a.P_rows, a.kmask = rot.data_ptr(), P_rows.data_ptr(), kmask.data_ptr()
    host_rows.host_rows_prepare(ctypes.byref(a))
    assert torch.equal(rot[0], rot[1]) and torch.equal(P_rows[0], P_rows[1]) and torch.equal(kmask[0], kmask[1])
    # the solver, from the oracle (differentiable): predicted keypoints -> depths, target keypoints -> pair mask
    kp, k3 = kps[0].clone().requires_grad_(), kps3d[0].clone().requires_grad_()
    km = kmask[0].bool()
    depth, _ = torch_ops.pairs_kpts_depth(kp, k3, rot[0].unsqueeze(-1), P_rows[0], training=True, kpts_2d_mask=km)
    with torch.no_grad():
        _, pmask = torch_ops.pairs_kpts_depth(kps[1], kps3d[1], rot[0].unsqueeze(-1), P_rows[0], training=True, kpts_2d_mask=km)
    depth_c, pmask_c = depth.detach().contiguous(), pmask.float().contiguous()
    cols = torch.zeros((25, BM), dtype=f32)
    corners = torch.zeros((2, BM, 8, 3), dtype=f32)
    iou3d = torch.zeros(BM, dtype=f32)
    sums = torch.zeros(25, dtype=f32)
    a.pair_depth, a.pair_mask = depth_c.data_ptr(), pmask_c.data_ptr()
    a.cols, a.corners_pred, a.corners_tgt = cols.data_ptr(), corners[0].data_ptr(), corners[1].data_ptr()
    a.iou3d, a.sums = iou3d.data_ptr(), sums.data_ptr()
    host_rows.host_rows_forward(ctypes.byref(a))
    gs = gsums.contiguous()
    gpois = torch.full((BM, C), float('nan'), dtype=f32)      # every element must be written
    gpair = torch.full((BM, NP), float('nan'), dtype=f32)
    a.grad_sums, a.grad_pois, a.grad_pair = gs.data_ptr(), gpois.data_ptr(), gpair.data_ptr()
    host_rows.host_rows_backward(ctypes.byref(a))
    assert torch.isfinite(gpois).all() and torch.isfinite(gpair).all()
    depth.backward(gpair)
    gk, gk3 = kp.grad.contiguous(), k3.grad.contiguous()
    a.grad_kps, a.grad_kps3d = gk.data_ptr(), gk3.data_ptr()
    host_rows.host_rows_finish(ctypes.byref(a))
    return sums, gpois.view(B, M, C), corners


@pytest.mark.parametrize("empty_image,mixed_flags", [(False, False), (True, False), (False, True)])
def test_row_kernel_arithmetic_equals_the_op_by_op_rows(cpu_backend, host_rows, empty_image, mixed_flags):
    loss, tv, pois = _rows_inputs(empty_image, mixed_flags)
    p = pois.clone().requires_grad_()
    S_ref, ix = loss._rows({'reg_pois': p, 'reg': None}, tv)
    assert len(ix) == 25 and sorted(ix.values()) == list(range(25))
    gs = torch.from_numpy(np.random.RandomState(5).uniform(0.5, 1.5, 25).astype(np.float32))
    (S_ref * gs).sum().backward()
    sums, gpois, corners = _run_host(host_rows, loss, tv, pois, gs)
    names = {v: k for k, v in ix.items()}
    for c in range(25):
        if names[c] == 'iou3d':
            continue
        ref = float(S_ref[c])
        assert abs(float(sums[c]) - ref) <= 2e-5 * max(abs(ref), 1.0), (names[c], float(sums[c]), ref)
    if mixed_flags:
        for name in ('trunc', 'invalid_l', 'kd_i'):
            assert float(S_ref[ix[name]]) > 0, name
        assert float(S_ref[ix['m2d']]) < float(S_ref[ix['m3d']]) and float(S_ref[ix['m2']]) < float(S_ref[ix['ov']])
    g_ref = p.grad
    scale = g_ref.abs().max().item()
    err = (gpois - g_ref).abs().max().item()
    assert err <= 2e-5 * scale, (err, scale, np.unravel_index((gpois - g_ref).abs().argmax().item(), tuple(g_ref.shape)))
    assert (gpois[~tv['reg_mask'].bool()] == 0).all()


def test_column_names_are_the_same_in_both_evaluations(cpu_backend):
    """`_losses_from_columns` addresses columns by name: the fused kernel's fixed order must be the op-by-op order."""
    from dcd_amd import ops
    loss, tv, pois = _rows_inputs()
    _, ix = loss._rows({'reg_pois': pois, 'reg': None}, tv)
    orig = ops.loss_rows
    ops.loss_rows = lambda p, spec, dim_mean, t: torch.zeros(25)
    try:
        _, ix_fused = loss._fused_rows({'reg_pois': pois}, tv, 1.0)
    finally:
        ops.loss_rows = orig
    assert ix == ix_fused


def test_batch_without_any_object_gives_zero_columns_and_zero_gradients(cpu_backend, host_rows):
    """No annotated object in the whole batch (the reference would index an empty list): every slot is empty, reads slot 0's
    all-zero targets (depth 0, zero-size box) and must still produce finite arithmetic -- all 25 columns 0, gradients 0."""
    loss, tv, pois = _rows_inputs()
    tv = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in tv.items()}
    tv['reg_mask'].zero_()
    gs = torch.ones(25)
    sums, gpois, _ = _run_host(host_rows, loss, tv, pois, gs)
    assert torch.isfinite(sums).all() and (sums == 0).all()
    assert (gpois == 0).all()
