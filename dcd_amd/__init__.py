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
__version__ = "0.1.0"
