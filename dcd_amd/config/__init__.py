"""`from dcd_amd.config import cfg` mirrors the reference's `from config import cfg` (DGDE/config/__init__.py)."""
import os

from .cfgnode import CfgNode
from .defaults import _C as cfg
from .dgde_run import dgde_overrides

# class-name -> id table the reference exports next to cfg (DGDE/config/__init__.py:16-28); data side only
TYPE_ID_CONVERSION = {"car": 0, "pedestrian": 1, "bicycle": 2, "motorcycle": 3, "barrier": 4, "bus": 5,
                      "construction_vehicle": 6, "traffic_cone": 7, "trailer": 8, "truck": 9, "DontCare": 10}


def get_cfg(yaml_file=None, opts=()):
    """A fresh, un-frozen copy of the defaults merged with the DGDE experiment (`dgde_run.dgde_overrides()`; or with
    `yaml_file`, e.g. the reference's own runs/DGDE.yaml, when given) and then with `opts` (list of key, value)."""
    c = cfg.clone()
    c.defrost()
    if yaml_file:
        c.merge_from_file(yaml_file)
    else:
        c.merge_from_list(dgde_overrides())
    if opts:
        c.merge_from_list(list(opts))
    return c


__all__ = ["cfg", "CfgNode", "get_cfg", "dgde_overrides", "TYPE_ID_CONVERSION"]
