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
        # MODEL.FP16 (detector.py:34-36): autocast around the backbone.  bfloat16 on MI355X (no loss scaling needed; BASELINE
        # config 3); the DCN op stays an fp32 op at its boundary and takes the split-bf16 matrix path (DCNv2/dcn_v2.py).
        amp = (torch.autocast(device_type=pixels.device.type, dtype=torch.bfloat16) if (training and self.fp16)
               else contextlib.nullcontext())
        with amp:
            features = self.backbone(pixels)
        return self.heads(features, targets) if training else self.heads(features, targets, test=self.test)
