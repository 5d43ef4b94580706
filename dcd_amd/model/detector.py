"""KeypointDetector = backbone + heads (DGDE/model/detector.py:12-44).

train: forward(images, targets) -> (loss_dict, log_loss_dict); eval: -> (result, eval_utils, visualize_preds)."""
import torch
from torch import nn

from dcd_amd.structures.image_list import to_image_list
from .backbone import build_backbone_DGDE
from .head.detector_head import bulid_head


class KeypointDetector(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.backbone = build_backbone_DGDE(cfg)
        self.heads = bulid_head(cfg, self.backbone.out_channels)
        self.test = cfg.DATASETS.TEST_SPLIT == 'test'
        self.fp16 = cfg.MODEL.FP16

    def forward(self, images, targets=None):
        if self.training and targets is None:
            raise ValueError("In training mode, targets should be passed")
        images = to_image_list(images)
        if self.training and self.fp16:
            # The DCN op is fp32 (as in the reference, cuda/dcn_v2_cuda.cu:58); under autocast its inputs are cast back.
            with torch.autocast(device_type=images.tensors.device.type):
                features = self.backbone(images.tensors)
        else:
            features = self.backbone(images.tensors)
        if self.training:
            return self.heads(features, targets)
        return self.heads(features, targets, test=self.test)
