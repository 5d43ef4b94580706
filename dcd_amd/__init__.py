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

# MIOpen's composable-kernel "grouped conv backward data" solver (its pick for the input gradient of DLA's stride-2 3x3 layers at
# bs 8) zero-fills its output with hipMemsetAsync and accumulates into it.  Inside a captured HIP graph that is a memset node, and
# memset nodes are not ordered reliably against their kernels on this stack (profiles/r02_graph_memset_hazard.txt; round 5: ATen
# reductions with memset-cleared semaphores returned wrong sums in some replays of the graphed train step).  Round 5 first tried to
# switch the solver off through MIOPEN_DEBUG_* variables: the name it exported is not read by the MIOpen build inside PyTorch, and
# the three names that build does read (..._3D_CONV_IMPLICIT_GEMM_HIP_BWD_XDLOPS, ..._HIP_GROUP_BWD_XDLOPS, ..._HIP_BWD_XDLOPS) did
# not keep the solver out either (tools/probes/miopen_env_probe.py; tools/check_graph_memsets.sh caught it at bs 8).  What does:
# a process that builds a whole-step graph runs these layers on our own kernels (ops.stride2_on_own_kernels, called by
# engine.trainer.GraphedTrainStep) -- no MIOpen convolution is left in the captured step.  No environment is touched here.
