from .dla_dcn import build_backbone as build_backbone_DGDE  # same export name as DGDE/model/backbone/__init__.py

__all__ = ["build_backbone_DGDE"]
