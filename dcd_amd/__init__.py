"""dcd_amd -- MI355X (gfx950) native implementation of the DGDE hot path of BraveGroup/DCD.

Layout mirrors the reference's DGDE package for the files on the hot path only:
  dcd_amd/csrc/            HIP kernels + C ABI (include/dcd_hip.h)  -> dcd_amd/libdcd_hip.so
  dcd_amd/_lib.py          ctypes loader of the C-ABI library (fails loudly when it is missing)
  dcd_amd/_ext.py          drop-in for the reference's `_ext` module (dcn_v2_forward/backward)
  dcd_amd/ops.py           autograd wrappers of the other kernels (edge depth, focal, GIoU, decode)
  dcd_amd/config/          cfg tree with the reference's key names (config/defaults.py, runs/DGDE.yaml)
  dcd_amd/model/           KeypointDetector = DLA-34+DCN backbone + heads (model/*.py of the reference)
  dcd_amd/structures/      ParamsList, to_image_list
  dcd_amd/data/synthetic.py  KITTI-shaped synthetic batches
  dcd_amd/engine/          train step / DDP helpers used by bench.py
"""
import os as _os

__version__ = "0.1.0"

# MIOpen's composable-kernel "grouped conv backward data" solver (picked for one of DLA's stride-2 3x3 layers) zero-fills its
# output with hipMemsetAsync and accumulates into it.  Inside a captured HIP graph that is a memset node, and memset nodes are
# not ordered reliably against their kernels on this stack (profiles/r02_graph_memset_hazard.txt; round 5: the graphed train
# step's ATen reductions with memset-cleared semaphores returned wrong sums in some replays).  With the solver off MIOpen takes
# its assembly Winograd kernel for that layer (same step time, 38.13 vs 38.21 ms); set the variable yourself to keep the solver.
# MIOpen reads it when the library is loaded, i.e. at `import torch`: a process that imports torch before this package must
# export it itself (bench.py, tests/conftest.py and __graft_entry__.py do).
_os.environ.setdefault("MIOPEN_DEBUG_GROUP_CONV_IMPLICIT_GEMM_HIP_BWD_XDLOPS", "0")
