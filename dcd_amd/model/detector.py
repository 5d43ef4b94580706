"""KeypointDetector = backbone + heads (DGDE/model/detector.py:12-44).

train: forward(images, targets) -> (loss_dict, log_loss_dict); eval: -> (result, eval_utils, visualize_preds).
`backbone` and `heads` are the reference's attribute names (state-dict prefixes)."""
import contextlib

import torch
from torch import nn

from dcd_amd.structures.image_list import to_image_list
from . import backbone as _backbone
from .head import detector_head


class KeypointDetector(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.fp16 = bool(cfg.MODEL.FP16)
        self.test = cfg.DATASETS.TEST_SPLIT == 'test'
        self.backbone = _backbone.build_backbone_DGDE(cfg)
        self.heads = detector_head.bulid_head(cfg, self.backbone.out_channels)

    def forward(self, images, targets=None):
        training = self.training
        if training and targets is None:
            raise ValueError("In training mode, targets should be passed")
        pixels = to_image_list(images).tensors
        # MODEL.FP16 (detector.py:34-36: autocast around the backbone).  On the device the mixed-precision region is a PRECISION
        # SCOPE of our own kernels, not torch.autocast: every 3x3 convolution and every DCNv2 contraction inside it rounds its
        # operands to bf16 and runs ONE product on the bf16 matrix cores with fp32 accumulation (DCD_PREC_BF16); activations,
        # normalisation statistics, sampling arithmetic and every sum stay fp32 -- no tensor changes type between our kernels, so none
        # of them falls back to a stock op (bfloat16 needs no loss scaling: BASELINE config 3).  Host tensors (the CPU tests) have
        # no kernels of ours: there the flag is torch's autocast, as in the reference.
        from dcd_amd import _ext
        if training and self.fp16:
            amp = (_ext.precision_scope("bf16") if pixels.is_cuda
                   else torch.autocast(device_type=pixels.device.type, dtype=torch.bfloat16))
        else:
            amp = contextlib.nullcontext()
        with amp:
            features = self.backbone(pixels)
        return self.heads(features, targets) if training else self.heads(features, targets, test=self.test)
