"""DisparityDenoiser (reference: filter/disparity_denoiser.py:51-313): a bilateral filter of the distance to the local tangent
plane of the disparity map.  The gradient of the blurred map is 2-D host work done with the reference's own expressions
(scipy.ndimage.gaussian_filter + np.gradient, disparity_denoiser.py:138-149); the windowed part - filter_size^2 neighbours per
pixel, three gaussians each - runs on the device (pmx_denoise_disparity)."""
import numpy as np
from scipy.ndimage import gaussian_filter

from .. import runtime
from ..matching_cost.matching_cost import ConfigError
from . import filter as _filter

_KEYS = ("filter_method", "filter_size", "sigma_euclidian", "sigma_color", "sigma_planar", "sigma_grad", "band")


@_filter.AbstractFilter.register_subclass("disparity_denoiser")
class DisparityDenoiser(_filter.AbstractFilter):
    # defaults of disparity_denoiser.py:56-62
    _FILTER_SIZE = 11
    _SIGMA_EUCLIDIAN = 4.0
    _SIGMA_COLOR = 100.0
    _SIGMA_PLANAR = 12.0
    _SIGMA_GRAD = 1.5
    _BAND = None

    def __init__(self, *args, cfg=None, **kwargs):
        self.cfg = self.check_conf(dict(cfg or {}))
        self._filter_size = int(self.cfg["filter_size"])
        self._sigma_euclidian = float(self.cfg["sigma_euclidian"])
        self._sigma_color = float(self.cfg["sigma_color"])
        self._sigma_planar = float(self.cfg["sigma_planar"])
        self._sigma_grad = float(self.cfg["sigma_grad"])
        self._band = self.cfg["band"]
        if self._filter_size % 2 == 0:  # the reference asserts (disparity_denoiser.py:80)
            raise ConfigError("filter_size must be odd")

    def check_conf(self, cfg):
        """disparity_denoiser.py:94-130"""
        for key, default in (("filter_size", self._FILTER_SIZE), ("sigma_euclidian", self._SIGMA_EUCLIDIAN),
                             ("sigma_color", self._SIGMA_COLOR), ("sigma_planar", self._SIGMA_PLANAR),
                             ("sigma_grad", self._SIGMA_GRAD), ("band", self._BAND)):
            cfg.setdefault(key, default)
        if cfg.get("filter_method") != "disparity_denoiser":
            raise ConfigError("filter_method must be disparity_denoiser")
        if not isinstance(cfg["filter_size"], int) or isinstance(cfg["filter_size"], bool) or not cfg["filter_size"] > 0:
            raise ConfigError("filter_size must be an int > 0")
        for key in ("sigma_euclidian", "sigma_color", "sigma_planar"):
            if not isinstance(cfg[key], float) or not cfg[key] > 0:
                raise ConfigError(f"{key} must be a float > 0")
        if not isinstance(cfg["sigma_grad"], float) or not cfg["sigma_grad"] >= 0:
            raise ConfigError("sigma_grad must be a float >= 0")
        if cfg["band"] is not None and not isinstance(cfg["band"], str):
            raise ConfigError("band must be a band name or None")
        for key in cfg:
            if key not in _KEYS:
                raise ConfigError(f"unknown filter key {key!r}")
        return cfg

    def desc(self):
        print("Disparity denoiser filter description")

    def get_grad(self, disp):
        """disparity_denoiser.py:138-149: (d/drow, d/dcol) of the gaussian-blurred map"""
        return np.stack(np.gradient(gaussian_filter(disp, sigma=self._sigma_grad)), axis=0)

    def _color_band(self, img_left):
        """disparity_denoiser.py:246-254: the image itself, the SECOND band of a multiband image, or the named band"""
        im = np.asarray(img_left["im"].data)
        if self._band is None:
            return im if im.ndim < 3 else im[1]
        names = [str(b) for b in np.asarray(img_left.coords["band_im"])]
        if self._band not in names:
            raise ValueError(f"{self._band!r} is not in list")  # what list.index raises in the reference
        return im[names.index(self._band)]

    def filter_disparity(self, disp, img_left=None, img_right=None, cv=None):
        """disparity_denoiser.py:223-313, in place on the valid, finite pixels."""
        dmap = np.asarray(disp["disparity_map"].data)
        grad = self.get_grad(dmap)
        eng = runtime.get_engine()
        disp["disparity_map"].data = eng.denoise_disparity(dmap, np.asarray(disp["validity_mask"].data), self._color_band(img_left),
                                                           grad[0], grad[1], self._filter_size, self._sigma_euclidian,
                                                           self._sigma_color, self._sigma_planar)
        disp.attrs["filter"] = "disparity_denoiser"
