"""AbstractInterpolation + the "mc-cnn" and "sgm" interpolations of rejected pixels (reference:
validation/interpolated_disparity.py:40-375).  Same registry mechanics and dataset protocol as the reference; the four
gather passes it takes from validation_cpp run on the GPU (pmx_interpolate_disparity)."""
import logging
from abc import ABCMeta, abstractmethod

import numpy as np

from .. import runtime
from ..criteria import mask_border


class AbstractInterpolation:
    """interpolated_disparity.py:40-150"""
    __metaclass__ = ABCMeta
    interpolation_methods_avail = {}

    def __new__(cls, **cfg):
        if cls is AbstractInterpolation:
            name = cfg["interpolated_disparity"]
            try:
                return super(AbstractInterpolation, cls).__new__(cls.interpolation_methods_avail[name])
            except (KeyError, TypeError):
                logging.error("No interpolation method named %s supported", name)
                raise KeyError
        return super(AbstractInterpolation, cls).__new__(cls)

    @classmethod
    def register_subclass(cls, short_name):
        def decorator(subclass):
            cls.interpolation_methods_avail[short_name] = subclass
            return subclass

        return decorator

    @abstractmethod
    def desc(self):
        print("Disparity interpolation method description for the validation step")

    @abstractmethod
    def interpolated_disparity(self, left, img_left=None, img_right=None, cv=None):
        """Updates left["disparity_map"] and left["validity_mask"] in place (occlusions / mismatches become
        FILLED_OCCLUSION / FILLED_MISMATCH)."""

    @staticmethod
    def _run(left, passes):
        disp, valid = runtime.get_engine().interpolate_disparity(left["disparity_map"].data, left["validity_mask"].data, passes)
        left["disparity_map"].data = disp
        left["validity_mask"].data = valid.astype(np.int32)  # the reference's masks come back as py::array_t<int>

    @staticmethod
    def _one(which, disp, valid):
        d, v = runtime.get_engine().interpolate_disparity(disp, valid, [which])
        return d, v.astype(np.int32)


@AbstractInterpolation.register_subclass("mc-cnn")
class McCnnInterpolation(AbstractInterpolation):
    """interpolated_disparity.py:153-262"""

    def __init__(self, **cfg):
        self.check_config(**cfg)

    def check_config(self, **cfg):
        """No optional configuration."""

    def desc(self):
        print("MC-CNN interpolation method")

    def interpolated_disparity(self, left, img_left=None, img_right=None, cv=None):
        self._run(left, ["occlusion_mc_cnn", "mismatch_mc_cnn"])
        left.attrs["interpolated_disparity"] = "mc-cnn"
        if left.attrs["offset_row_col"] > 0:
            left["validity_mask"] = mask_border(left)

    @staticmethod
    def interpolate_occlusion_mc_cnn(disp, valid):
        return AbstractInterpolation._one("occlusion_mc_cnn", disp, valid)

    @staticmethod
    def interpolate_mismatch_mc_cnn(disp, valid):
        return AbstractInterpolation._one("mismatch_mc_cnn", disp, valid)


@AbstractInterpolation.register_subclass("sgm")
class SgmInterpolation(AbstractInterpolation):
    """interpolated_disparity.py:265-375"""

    def __init__(self, **cfg):
        self.check_config(**cfg)

    def check_config(self, **cfg):
        """No optional configuration."""

    def desc(self):
        print("SGM interpolation method")

    def interpolated_disparity(self, left, img_left=None, img_right=None, cv=None):
        self._run(left, ["mismatch_sgm", "occlusion_sgm"])
        left.attrs["interpolated_disparity"] = "sgm"

    @staticmethod
    def interpolate_occlusion_sgm(disp, valid):
        return AbstractInterpolation._one("occlusion_sgm", disp, valid)

    @staticmethod
    def interpolate_mismatch_sgm(disp, valid):
        return AbstractInterpolation._one("mismatch_sgm", disp, valid)
