"""StdIntensity (reference: cost_volume_confidence/std_intensity.py:38-124): standard deviation of the left image in the
matching window.  2-D host work (float64 cumulative sums, img_tools.py:834-952), no volume involved."""
import numpy as np

from ..matching_cost.matching_cost import ConfigError
from . import cost_volume_confidence as _cvc


def compute_mean_raster(im, win_size):
    """img_tools.py:834-879: box mean by cumulative sums (float64)."""
    ny_, nx_ = im.shape
    r_mean = np.nancumsum(np.r_[np.zeros((1, nx_)), im], axis=0)
    r_mean = r_mean[win_size:, :] - r_mean[:-win_size, :]
    r_mean = np.cumsum(np.c_[np.zeros(ny_ - (win_size - 1)), r_mean], axis=1)
    r_mean = r_mean[:, win_size:] - r_mean[:, :-win_size]
    return r_mean / float(win_size * win_size)


def compute_std_raster(im, win_size):
    """img_tools.py:915-952: sqrt(E[x^2] - E[x]^2), tiny variances clipped to 0."""
    mean_ = compute_mean_raster(im, win_size)
    mean_power_two = compute_mean_raster(im ** 2, win_size)
    var = mean_power_two - mean_ ** 2
    var[np.where(var < (10 ** (-15) * abs(mean_power_two)))] = 0
    return np.sqrt(var)


@_cvc.AbstractCostVolumeConfidence.register_subclass("std_intensity")
class StdIntensity(_cvc.AbstractCostVolumeConfidence):
    _method = "intensity_std"

    def __init__(self, **cfg):
        self.cfg = self.check_conf(**cfg)
        self._indicator = self._method + self.cfg["indicator"]

    def check_conf(self, **cfg):
        cfg.setdefault("indicator", self._indicator)
        if cfg.get("confidence_method") != "std_intensity" or not isinstance(cfg["indicator"], str):
            raise ConfigError("confidence_method must be std_intensity and indicator a str")
        for key in cfg:
            if key not in ("confidence_method", "indicator"):
                raise ConfigError(f"unknown confidence key {key!r}")
        return cfg

    def desc(self):
        print("Intensity confidence method")

    def confidence_prediction(self, disp, img_left=None, img_right=None, cv=None):
        """std_intensity.py:80-124"""
        nb_row, nb_col = img_left.sizes["row"], img_left.sizes["col"]
        window_size = cv.attrs["window_size"]
        conf = np.full((nb_row, nb_col), np.nan, dtype=np.float32)
        off = int((window_size - 1) / 2)
        from .. import runtime

        std = compute_std_raster(np.asarray(runtime.select_band(img_left, cv.attrs.get("band_correl"))), window_size)  # std_intensity.py:110-121
        if off != 0:
            conf[off:-off, off:-off] = std
        else:
            conf = std
        return self.allocate_confidence_map(self._indicator, conf, disp, cv)
