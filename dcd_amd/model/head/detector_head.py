"""Head = predictor + (loss | post-processor)  (DGDE/model/head/detector_head.py:10-34).

The three sub-module names (`predictor`, `loss_evaluator`, `post_processor`) are state-dict prefixes and therefore the
reference's; `bulid_head` [sic] is its factory's spelling and part of the API."""
import contextlib

import torch
from torch import nn

from . import detector_infer, detector_loss, detector_predictor


def _maybe_autocast(enabled, tensor):
    """MODEL.FP16 around the predictor (detector_head.py:20-22).  Device tensors: the precision scope of our own kernels
    (operands of the 3x3 convolutions rounded to bf16, fp32 accumulate and storage; see model/detector.py); host tensors: autocast."""
    if not enabled:
        return contextlib.nullcontext()
    if tensor.is_cuda:
        from dcd_amd import _ext
        return _ext.precision_scope("bf16")
    return torch.autocast(device_type=tensor.device.type)


class Detect_Head(nn.Module):
    def __init__(self, cfg, in_channels):
        super().__init__()
        self.fp16 = bool(cfg.MODEL.FP16)
        self.predictor = detector_predictor.make_predictor(cfg, in_channels)
        self.loss_evaluator = detector_loss.make_loss_evaluator(cfg)
        self.post_processor = detector_infer.make_post_processor(cfg)

    def forward(self, features, targets=None, test=False):
        with _maybe_autocast(self.fp16, features):
            predictions = self.predictor(features, targets)
        if not self.training:
            return self.post_processor(predictions, targets, test=test, features=features)
        return self.loss_evaluator(predictions, targets)


def bulid_head(cfg, in_channels):
    return Detect_Head(cfg, in_channels)
