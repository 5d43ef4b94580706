import os
import sys


import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_dcn():
    """The C oracle bound with the reference's `_ext` signatures (test infrastructure)."""
    from oracle import dcn_oracle
    dcn_oracle.build()
    return dcn_oracle


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture()
def cpu_backend(monkeypatch, oracle_dcn):
    """Run the host-side model / loss code on the CPU by patching the ORACLE in for every HIP entry point.
    Test-only: the product modules themselves have no CPU path and raise on CPU tensors."""
    from dcd_amd import ops
    from dcd_amd.model.backbone.DCNv2 import dcn_v2
    from oracle import torch_ops
    for name in ("pairs_kpts_depth", "compute_z", "focal_loss", "giou_loss", "nms_hm", "select_topk",
                 "select_point_of_interest", "iou_3d"):
        monkeypatch.setattr(ops, name, getattr(torch_ops, name))
    monkeypatch.setattr(dcn_v2, "_backend", oracle_dcn)
    return torch_ops
