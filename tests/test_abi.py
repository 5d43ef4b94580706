"""The C-ABI library loads on a machine without a GPU and exports every symbol include/dcd_hip.h declares
(no compute calls here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "dcd_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dcd_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    syms = declared_symbols()
    for must in ("dcd_dcn_v2_forward", "dcd_dcn_v2_backward", "dcd_edge_depth_forward", "dcd_edge_depth_backward",
                 "dcd_focal_loss", "dcd_giou_loss", "dcd_heatmap_topk", "dcd_nms_hm", "dcd_poi_gather", "dcd_iou3d"):
        assert must in syms


def test_library_loads_and_exports_every_declared_symbol():
    from dcd_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(handle, name), "libdcd_hip.so does not export %s" % name
    assert set(_lib.SIGNATURES) == set(declared_symbols()), "ctypes table and header must list the same functions"
    handle.dcd_version.restype = ctypes.c_char_p
    assert handle.dcd_version().startswith(b"dcd_hip")


def test_workspace_query_is_pure_host_code():
    from dcd_amd import _lib
    L = _lib.lib()
    n = L.dcd_dcn_v2_workspace_bytes(8, 64, 96, 320, 64, 3, 3, 1, 1, 1, 1, 1, 1, 1)
    assert n >= 2 * 64 * 576 * 4
    assert L.dcd_dcn_v2_workspace_bytes(8, 64, 96, 320, 64, 3, 3, 0, 1, 1, 1, 1, 1, 1) == 0   # stride 0 is invalid


def test_header_lists_every_environment_variable_the_library_reads():
    """include/dcd_hip.h "Environment" == the names passed to dcd_env() in csrc/ (the only way the library reads its environment:
    no raw getenv), so a maintainer linking the library can see -- or compile out, -DDCD_NO_TUNING_ENV -- every switch."""
    import glob
    src = ""
    for p in glob.glob(os.path.join(ROOT, "dcd_amd", "csrc", "*")):
        if p.endswith((".hip", ".inc", ".h")):
            text = open(p).read()
            if not p.endswith("tuning_env.h"):
                assert not re.search(r"(?<![A-Za-z_])getenv\s*\(", text), "%s reads the environment directly" % p
            src += text
    used = set(re.findall(r'dcd_env\("([A-Z0-9_]+)"\)', src))
    header = open(os.path.join(ROOT, "include", "dcd_hip.h")).read()
    start = header.index(" * Environment.")
    env_section = header[start:header.index("*/", start)]
    listed = set(re.findall(r"\b(DCD_[A-Z0-9_]+)\b", env_section)) - {"DCD_NO_TUNING_ENV"}
    assert used and used == listed, (sorted(used - listed), sorted(listed - used))
