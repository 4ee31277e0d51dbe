"""AbstractMultiscale (reference: multiscale/multiscale.py:34-153)."""
from abc import ABCMeta, abstractmethod

import numpy as np

from .. import constants as cst


class AbstractMultiscale:
    __metaclass__ = ABCMeta

    multiscale_methods_avail = {}
    cfg = None

    def __new__(cls, _left_img=None, _right_img=None, **cfg):
        if cls is AbstractMultiscale:
            method = cfg.get("multiscale_method")
            if isinstance(method, str):
                try:
                    return super(AbstractMultiscale, cls).__new__(cls.multiscale_methods_avail[method])
                except KeyError:
                    raise KeyError("No multiscale method named {} supported".format(method))
            raise KeyError("No multiscale method named {} supported".format(method))
        return super(AbstractMultiscale, cls).__new__(cls)

    @classmethod
    def register_subclass(cls, short_name, *args):
        def decorator(subclass):
            cls.multiscale_methods_avail[short_name] = subclass
            for arg in args:
                cls.multiscale_methods_avail[arg] = subclass
            return subclass

        return decorator

    @abstractmethod
    def desc(self):
        """Describes the multiscale method"""

    @abstractmethod
    def disparity_range(self, disp, disp_min, disp_max):
        """Per-pixel disparity range of the next scale from the current disparity map."""

    @staticmethod
    def mask_invalid_disparities(disp):
        """multiscale.py:129-153: copy of the disparity map with every invalid pixel set to NaN."""
        out = np.array(disp["disparity_map"].data, copy=True)
        out[(np.asarray(disp["validity_mask"].data) & cst.PANDORA_MSK_PIXEL_INVALID) != 0] = np.nan
        return out
