from .params_3d import ParamsList
from .image_list import ImageList, to_image_list

__all__ = ["ParamsList", "ImageList", "to_image_list"]
