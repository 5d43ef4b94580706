"""Head = predictor + (loss | post-processor)  (DGDE/model/head/detector_head.py:10-34)."""
import torch
from torch import nn

from .detector_predictor import make_predictor
from .detector_loss import make_loss_evaluator
from .detector_infer import make_post_processor


class Detect_Head(nn.Module):
    def __init__(self, cfg, in_channels):
        super().__init__()
        self.predictor = make_predictor(cfg, in_channels)
        self.loss_evaluator = make_loss_evaluator(cfg)
        self.post_processor = make_post_processor(cfg)
        self.fp16 = cfg.MODEL.FP16

    def forward(self, features, targets=None, test=False):
        if self.fp16:
            with torch.autocast(device_type=features.device.type):
                x = self.predictor(features, targets)
        else:
            x = self.predictor(features, targets)
        if self.training:
            return self.loss_evaluator(x, targets)
        return self.post_processor(x, targets, test=test, features=features)


def bulid_head(cfg, in_channels):   # [sic] the reference's spelling is part of its API
    return Detect_Head(cfg, in_channels)
