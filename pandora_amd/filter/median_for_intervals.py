"""MedianForIntervalsFilter (reference: filter/median_for_intervals.py:38-212): the median filter applied to the two
interval-bound layers of the confidence measure (on the device, the NaN-ignoring median of pmx_median_filter_disparity), then the
optional regularisation of the bounds in ambiguous zones (host side, ..interval_tools)."""
import numpy as np

from .. import constants as cst
from .. import runtime
from ..matching_cost.matching_cost import ConfigError
from . import filter as _filter


@_filter.AbstractFilter.register_subclass("median_for_intervals")
class MedianForIntervalsFilter(_filter.AbstractFilter):
    _FILTER_SIZE = 3
    _AMBIGUITY_THRESHOLD = 0.6
    _AMBIGUITY_KERNEL_SIZE = 5
    _VERTICAL_DEPTH = 0
    _QUANTILE_REGULARIZATION = 1.0

    def __init__(self, *args, cfg=None, step=1, **kwargs):
        self.cfg = self.check_conf(dict(cfg or {}))
        self._filter_size = int(self.cfg["filter_size"])
        self._interval_indicator = str(self.cfg["interval_indicator"])
        self._regularization = bool(self.cfg["regularization"])
        self._vertical_depth = int(self.cfg["vertical_depth"])
        self._quantile_regularization = float(self.cfg["quantile_regularization"])
        self._ambiguity_indicator = str(self.cfg["ambiguity_indicator"])
        self._ambiguity_threshold = float(self.cfg["ambiguity_threshold"])
        self._ambiguity_kernel_size = int(self.cfg["ambiguity_kernel_size"])
        self._step = step

    def check_conf(self, cfg):
        """median_for_intervals.py:73-115"""
        for key, default in (("filter_size", self._FILTER_SIZE), ("interval_indicator", ""), ("regularization", False),
                             ("vertical_depth", self._VERTICAL_DEPTH), ("quantile_regularization", self._QUANTILE_REGULARIZATION),
                             ("ambiguity_indicator", ""), ("ambiguity_threshold", self._AMBIGUITY_THRESHOLD),
                             ("ambiguity_kernel_size", self._AMBIGUITY_KERNEL_SIZE)):
            cfg.setdefault(key, default)
        if cfg.get("filter_method") != "median_for_intervals":
            raise ConfigError("filter_method must be median_for_intervals")

        def is_int(v):
            return isinstance(v, int) and not isinstance(v, bool)

        if not is_int(cfg["filter_size"]) or cfg["filter_size"] < 1 or cfg["filter_size"] % 2 == 0:
            raise ConfigError("filter_size must be an odd integer >= 1")
        if cfg["filter_size"] > 15:  # pmx_median_filter_disparity sorts the window in registers
            raise ConfigError("pandora_amd supports median filter_size up to 15")
        if not isinstance(cfg["interval_indicator"], str) or not isinstance(cfg["ambiguity_indicator"], str):
            raise ConfigError("interval_indicator and ambiguity_indicator must be str")
        if not isinstance(cfg["regularization"], bool):
            raise ConfigError("regularization must be a bool")
        for key in ("ambiguity_threshold", "quantile_regularization"):
            if not isinstance(cfg[key], float) or not 0 <= cfg[key] <= 1:
                raise ConfigError(f"{key} must be a float in [0, 1]")
        if not is_int(cfg["ambiguity_kernel_size"]) or cfg["ambiguity_kernel_size"] <= 0 or cfg["ambiguity_kernel_size"] % 2 != 1:
            raise ConfigError("ambiguity_kernel_size must be an odd int > 0")
        if not is_int(cfg["vertical_depth"]) or cfg["vertical_depth"] < 0:
            raise ConfigError("vertical_depth must be an int >= 0")
        for key in cfg:
            if key not in ("filter_method", "filter_size", "interval_indicator", "regularization", "ambiguity_indicator",
                           "ambiguity_threshold", "ambiguity_kernel_size", "vertical_depth", "quantile_regularization"):
                raise ConfigError(f"unknown filter key {key!r}")
        return cfg

    def desc(self):
        print("Median filter for intervals description")

    @property
    def margins(self):
        """median_for_intervals.py:123-126"""
        from ..margins import uniform

        return uniform(self._filter_size * self._step)

    def _layer(self, disp, name):
        return list(disp.coords["indicator"]).index(name)

    def filter_disparity(self, disp, img_left=None, img_right=None, cv=None):
        """median_for_intervals.py:128-212: both bound layers are median-filtered (NaN bounds ignored inside a window and left as
        they are); with ``regularization`` the bounds of ambiguous segments are replaced and those pixels get the
        PANDORA_MSK_PIXEL_INTERVAL_REGULARIZED bit."""
        suffix = "" if self._interval_indicator == "" else "." + self._interval_indicator
        k_inf = self._layer(disp, "confidence_from_interval_bounds_inf" + suffix)
        k_sup = self._layer(disp, "confidence_from_interval_bounds_sup" + suffix)
        conf = disp["confidence_measure"].data
        eng = runtime.get_engine()
        no_flags = np.zeros(conf.shape[:2], np.int64)
        for k in (k_inf, k_sup):
            conf[:, :, k] = eng.median_filter_disparity(np.ascontiguousarray(conf[:, :, k], np.float32), no_flags, self._filter_size)
        if self._regularization:
            from ..interval_tools import interval_regularization

            amb_name = "confidence_from_ambiguity" + ("" if self._ambiguity_indicator == "" else "." + self._ambiguity_indicator)
            lo, hi, mask = interval_regularization(conf[:, :, k_inf].copy(), conf[:, :, k_sup].copy(), conf[:, :, self._layer(disp, amb_name)],
                                                   self._ambiguity_threshold, self._ambiguity_kernel_size, self._vertical_depth,
                                                   self._quantile_regularization)
            # the filter may run several times: set the bit, do not add it (median_for_intervals.py:186-188)
            disp["validity_mask"].data[mask] |= cst.PANDORA_MSK_PIXEL_INTERVAL_REGULARIZED
            conf[:, :, k_inf] = lo
            conf[:, :, k_sup] = hi
