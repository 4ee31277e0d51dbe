"""BilateralFilter (reference: filter/bilateral.py:40-255), computed on the device (pmx_bilateral_filter_disparity)."""
import numpy as np

from .. import runtime
from ..matching_cost.matching_cost import ConfigError
from . import filter as _filter


@_filter.AbstractFilter.register_subclass("bilateral")
class BilateralFilter(_filter.AbstractFilter):
    _SIGMA_COLOR = 2.0
    _SIGMA_SPACE = 6.0

    def __init__(self, cfg=None, image_shape=None, step=1, **kwargs):
        self.cfg = self.check_conf(dict(cfg or {}))
        self._sigma_color = float(self.cfg["sigma_color"])
        self._sigma_space = float(self.cfg["sigma_space"])
        self._image_shape = [] if image_shape is None else image_shape
        self._step = step

    @property
    def margins(self):
        """bilateral.py:63-67"""
        from ..margins import uniform

        return uniform(min([*self._image_shape, int(3 * self._sigma_space + 1)]) * self._step)

    def check_conf(self, cfg):
        """bilateral.py:74-96"""
        cfg.setdefault("sigma_color", self._SIGMA_COLOR)
        cfg.setdefault("sigma_space", self._SIGMA_SPACE)
        if cfg.get("filter_method") != "bilateral":
            raise ConfigError("filter_method must be bilateral")
        for key in ("sigma_color", "sigma_space"):
            if not isinstance(cfg[key], float) or not cfg[key] > 0:
                raise ConfigError(f"{key} must be a float > 0")
        for key in cfg:
            if key not in ("filter_method", "sigma_color", "sigma_space"):
                raise ConfigError(f"unknown filter key {key!r}")
        return cfg

    def desc(self):
        print("Bilateral filter description")

    def filter_disparity(self, disp, img_left=None, img_right=None, cv=None):
        """bilateral.py:100-140: weighted mean over the valid pixels of the window, in place."""
        eng = runtime.get_engine()
        disp["disparity_map"].data = eng.bilateral_filter_disparity(np.asarray(disp["disparity_map"].data),
                                                                    np.asarray(disp["validity_mask"].data),
                                                                    self._sigma_color, self._sigma_space)
        disp.attrs["filter"] = "bilateral"
