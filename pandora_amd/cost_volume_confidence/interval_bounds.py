"""Interval bounds (reference: cost_volume_confidence/interval_bounds.py:36-231): the disparity interval whose possibility reaches
a threshold, computed on the device-resident cost volume (pmx_interval_bounds).  The optional regularisation of the two maps
(interval_tools.py:36-96, a graph of ambiguous segments) is host-side work on 2-D maps and lives in ..interval_tools."""
from ..matching_cost.matching_cost import ConfigError
from . import cost_volume_confidence as _cvc
from .risk import _device_volume_and_grids


@_cvc.AbstractCostVolumeConfidence.register_subclass("interval_bounds")
class IntervalBounds(_cvc.AbstractCostVolumeConfidence):
    _POSSIBILITY_THRESHOLD = 0.9
    _AMBIGUITY_THRESHOLD = 0.6
    _AMBIGUITY_KERNEL_SIZE = 5
    _VERTICAL_DEPTH = 0
    _QUANTILE_REGULARIZATION = 1.0
    _method = "interval_bounds"

    def __init__(self, **cfg):
        self.cfg = self.check_conf(**cfg)
        self._possibility_threshold = float(self.cfg["possibility_threshold"])
        self._ambiguity_indicator = str(self.cfg["ambiguity_indicator"])
        self._ambiguity_threshold = float(self.cfg["ambiguity_threshold"])
        self._ambiguity_kernel_size = int(self.cfg["ambiguity_kernel_size"])
        self._regularization = bool(self.cfg["regularization"])
        self._vertical_depth = int(self.cfg["vertical_depth"])
        self._quantile_regularization = float(self.cfg["quantile_regularization"])
        self._indicator = self._method + str(self.cfg["indicator"])
        self._indicator_inf = self._method + "_inf" + str(self.cfg["indicator"])
        self._indicator_sup = self._method + "_sup" + str(self.cfg["indicator"])

    def check_conf(self, **cfg):
        """interval_bounds.py:84-120"""
        for key, default in (("possibility_threshold", self._POSSIBILITY_THRESHOLD), ("regularization", False), ("ambiguity_indicator", ""),
                             ("ambiguity_threshold", self._AMBIGUITY_THRESHOLD), ("ambiguity_kernel_size", self._AMBIGUITY_KERNEL_SIZE),
                             ("vertical_depth", self._VERTICAL_DEPTH), ("quantile_regularization", self._QUANTILE_REGULARIZATION),
                             ("indicator", self._indicator)):
            cfg.setdefault(key, default)
        if cfg.get("confidence_method") != "interval_bounds":
            raise ConfigError("confidence_method must be interval_bounds")
        for key in ("possibility_threshold", "ambiguity_threshold", "quantile_regularization"):
            if not isinstance(cfg[key], float) or not 0 <= cfg[key] <= 1:
                raise ConfigError(f"{key} must be a float in [0, 1]")
        if not isinstance(cfg["regularization"], bool):
            raise ConfigError("regularization must be a bool")
        if not isinstance(cfg["ambiguity_indicator"], str) or not isinstance(cfg["indicator"], str):
            raise ConfigError("ambiguity_indicator and indicator must be str")
        k = cfg["ambiguity_kernel_size"]
        if not isinstance(k, int) or isinstance(k, bool) or k <= 0 or k % 2 != 1:
            raise ConfigError("ambiguity_kernel_size must be an odd int > 0")
        if not isinstance(cfg["vertical_depth"], int) or isinstance(cfg["vertical_depth"], bool) or cfg["vertical_depth"] < 0:
            raise ConfigError("vertical_depth must be an int >= 0")
        for key in cfg:
            if key not in ("confidence_method", "possibility_threshold", "regularization", "ambiguity_indicator", "ambiguity_threshold",
                           "ambiguity_kernel_size", "vertical_depth", "quantile_regularization", "indicator"):
                raise ConfigError(f"unknown confidence key {key!r}")
        return cfg

    def desc(self):
        print("Interval bounds confidence method with regularization")

    def confidence_prediction(self, disp, img_left=None, img_right=None, cv=None):
        """interval_bounds.py:129-196: layers interval_bounds_inf and interval_bounds_sup."""
        type_factor = -1.0 if cv.attrs["type_measure"] == "min" else 1.0
        dcv, gmin, gmax = _device_volume_and_grids(cv, img_left)
        inf, sup = dcv.engine.interval_bounds(dcv, self._possibility_threshold, type_factor, gmin, gmax)
        if self._regularization:
            from ..interval_tools import interval_regularization

            indicator = "confidence_from_ambiguity" if self._ambiguity_indicator == "" else "confidence_from_ambiguity." + self._ambiguity_indicator
            amb = cv["confidence_measure"].sel({"indicator": indicator}).data  # interval_bounds.py:177-181
            inf, sup, _ = interval_regularization(inf, sup, amb, self._ambiguity_threshold, self._ambiguity_kernel_size,
                                                  self._vertical_depth, self._quantile_regularization)
        disp, cv = self.allocate_confidence_map(self._indicator_inf, inf, disp, cv)
        disp, cv = self.allocate_confidence_map(self._indicator_sup, sup, disp, cv)
        return disp, cv
