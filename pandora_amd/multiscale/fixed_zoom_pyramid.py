"""FixedZoomPyramid (reference: multiscale/fixed_zoom_pyramid.py:38-184); the window search runs on the device
(pmx_disparity_range), the order-0 zoom is scipy's, as in the reference."""
import numpy as np
from scipy.ndimage import zoom

from .. import runtime
from ..matching_cost.matching_cost import ConfigError
from . import multiscale


@multiscale.AbstractMultiscale.register_subclass("fixed_zoom_pyramid")
class FixedZoomPyramid(multiscale.AbstractMultiscale):
    _PYRAMID_NUM_SCALES = 2
    _PYRAMID_SCALE_FACTOR = 2
    _PYRAMID_MARGE = 1

    def __init__(self, left_img=None, right_img=None, **cfg):
        self.cfg = self.check_conf(left_img, right_img, **cfg)
        self._num_scales = self.cfg["num_scales"]
        self._scale_factor = self.cfg["scale_factor"]
        self._marge = self.cfg["marge"]

    def check_conf(self, left_img, right_img, **cfg):
        """fixed_zoom_pyramid.py:60-98"""
        cfg.setdefault("num_scales", self._PYRAMID_NUM_SCALES)
        cfg.setdefault("scale_factor", self._PYRAMID_SCALE_FACTOR)
        cfg.setdefault("marge", self._PYRAMID_MARGE)
        for img in (left_img, right_img):  # input disparities cannot be grids
            if img is not None and isinstance(img.attrs.get("disparity_source"), str):
                raise TypeError("Multiscale processing does not accept input disparity grids.")
        if cfg.get("multiscale_method") != "fixed_zoom_pyramid":
            raise ConfigError("multiscale_method must be fixed_zoom_pyramid")
        for key, ok in (("num_scales", lambda x: x > 1), ("scale_factor", lambda x: x > 1), ("marge", lambda x: x >= 0)):
            if isinstance(cfg[key], bool) or not isinstance(cfg[key], int) or not ok(cfg[key]):
                raise ConfigError(f"{key}: bad value {cfg[key]!r}")
        for key in cfg:
            if key not in ("multiscale_method", "num_scales", "scale_factor", "marge"):
                raise ConfigError(f"unknown multiscale key {key!r}")
        return cfg

    def desc(self):
        print("FixedZoomPyramid method")

    def disparity_range(self, disp, disp_min, disp_max):
        """fixed_zoom_pyramid.py:106-184: window min / max of the valid disparities -/+ marge, the full range for invalid
        pixels and the frame, then an order-0 zoom to the next scale."""
        gmin, gmax = int(np.nanmin(disp_min)), int(np.nanmax(disp_max))
        eng = runtime.get_engine()
        lo, hi = eng.disparity_range(np.asarray(disp["disparity_map"].data), np.asarray(disp["validity_mask"].data),
                                     disp.attrs["window_size"], self._marge, gmin, gmax)
        if self._scale_factor == 1:
            return lo, hi
        return zoom(lo, self._scale_factor, order=0), zoom(hi, self._scale_factor, order=0)
