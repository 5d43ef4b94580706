"""csrc/loss_rows.hip on the GPU: the per-object rows of the loss as one kernel against the same terms evaluated op by op
(Loss_Computation._rows on the HIP ops), values and gradients, and the whole loss through both."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
import golden_inputs as gi  # noqa: E402
from test_host_golden import small_cfg, LOSS_KEYS  # noqa: E402
from test_host_rows import _rows_inputs  # noqa: E402

pytestmark = pytest.mark.gpu


def _losses(cuda, monkeypatch, rows, tv_cpu, pois_cpu, cls_cpu, hm_cpu):
    from dcd_amd.model.head.detector_loss import Loss_Computation
    monkeypatch.setenv("DCD_LOSS_ROWS", rows)
    monkeypatch.setenv("DCD_LOSS_GRAPH", "0")
    loss = Loss_Computation(small_cfg(str(cuda)))
    assert loss.fused_rows is (rows == "1")
    tv = {k: (v.to(cuda) if torch.is_tensor(v) else v) for k, v in tv_cpu.items()}
    pois = pois_cpu.to(cuda).requires_grad_()
    cls = cls_cpu.to(cuda).requires_grad_()
    loss_dict, names, packed = loss._core({'cls': cls, 'reg': None, 'reg_pois': pois}, hm_cpu.to(cuda), tv)
    loss_dict.total.backward()
    return loss_dict, dict(zip(names, packed.tolist())), pois.grad, cls.grad


@pytest.mark.parametrize("empty_image,mixed_flags", [(False, False), (True, False), (False, True)])
def test_loss_through_the_row_kernel_equals_the_op_by_op_loss(cuda, monkeypatch, empty_image, mixed_flags):
    monkeypatch.setenv("DCD_LOSS_ROWS", "0")
    host_loss, tv, pois = _rows_inputs(empty_image, mixed_flags)
    preds, targets = gi.loss_inputs()
    hm, _ = host_loss.prepare_targets(targets)
    cls = torch.from_numpy(preds["cls"])
    a = _losses(cuda, monkeypatch, "0", tv, pois, cls, hm)
    b = _losses(cuda, monkeypatch, "1", tv, pois, cls, hm)
    assert list(a[0].keys()) == list(b[0].keys()) == LOSS_KEYS
    for k in LOSS_KEYS:
        ref = float(a[0][k])
        assert abs(float(b[0][k]) - ref) <= 2e-5 * max(abs(ref), 1e-3), (k, float(b[0][k]), ref)
    assert a[1].keys() == b[1].keys()
    for k, ref in a[1].items():
        if ref != ref:                                           # mean over an empty set: NaN in both (as the reference)
            assert b[1][k] != b[1][k], k
            continue
        assert abs(b[1][k] - ref) <= 2e-5 * max(abs(ref), 1e-3), ("log " + k, b[1][k], ref)
    scale = a[2].abs().max().item()
    assert (a[2] - b[2]).abs().max().item() <= 2e-5 * scale
    assert torch.equal(a[3], b[3])                               # the heat-map term does not pass through the rows
    assert (b[2][~tv['reg_mask'].bool()] == 0).all()


def test_row_kernel_rejects_a_channel_map_that_does_not_cover_the_heads(cuda):
    from dcd_amd import _lib
    a = _lib.LossRowsArgs()
    a.B, a.M, a.C, a.K, a.NP, a.num_classes = 1, 4, 415, 73, 1500, 3
    st = _lib.lib().dcd_loss_rows_forward(None, a)
    assert st == 1


def test_batch_without_any_object_trains_on_the_heat_map_only(cuda, monkeypatch):
    """No annotated object in the batch: the per-object losses are exactly 0 with zero gradient w.r.t. the head outputs, the
    heat-map loss and its gradient are unaffected, nothing is non-finite (the op-by-op form divides by the zero depth target of
    slot 0 in a logging column; the kernel never forms that quotient for an empty slot)."""
    monkeypatch.setenv("DCD_LOSS_ROWS", "0")
    host_loss, tv, pois = _rows_inputs()
    tv = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in tv.items()}
    tv['reg_mask'].zero_()
    preds, targets = gi.loss_inputs()
    hm, _ = host_loss.prepare_targets(targets)
    loss_dict, log, gp, gc = _losses(cuda, monkeypatch, "1", tv, pois, torch.from_numpy(preds["cls"]), hm)
    for k in LOSS_KEYS:
        v = float(loss_dict[k])
        assert v == v and abs(v) != float("inf"), k
        if k != "hm_loss":
            assert v == 0.0, (k, v)
    assert float(loss_dict["hm_loss"]) > 0
    assert (gp == 0).all() and torch.isfinite(gc).all() and gc.abs().max().item() > 0
