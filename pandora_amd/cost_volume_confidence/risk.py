"""Risk (reference: cost_volume_confidence/risk.py:38-233): mean span of the disparities within eta of the minimum cost, with the
matching bounds; the reductions over D run on the device-resident cost volume (pmx_risk)."""
import numpy as np

from ..matching_cost.matching_cost import ConfigError
from . import cost_volume_confidence as _cvc


def _device_volume_and_grids(cv, img_left):
    """The resident volume and the per-pixel [min, max] grids cropped to it; (None, None) when both grids are constant and equal
    to the volume's own range - the device then searches the whole range and two int64 maps need not be uploaded."""
    arr = cv["cost_volume"]
    if not hasattr(arr, "device_cv"):
        raise TypeError("confidence_prediction needs a device-resident cost volume (pandora_amd has no CPU path)")
    dcv = arr.device_cv
    ny_, nx_, _ = dcv.shape
    gmin = np.asarray(img_left["disparity"].sel(band_disp="min").data)[:ny_, :nx_]
    gmax = np.asarray(img_left["disparity"].sel(band_disp="max").data)[:ny_, :nx_]
    disp = np.asarray(cv.coords["disp"])
    if gmin.flat[0] == disp[0] and gmax.flat[0] == disp[-1]:
        if gmin.dtype.kind in "iu" and gmax.dtype.kind in "iu":  # (one threaded pass per grid: matching_cost.grid_extrema)
            from ..matching_cost.matching_cost import grid_extrema

            constant = grid_extrema(gmin) == (disp[0], disp[0]) and grid_extrema(gmax) == (disp[-1], disp[-1])
        else:
            constant = (gmin == disp[0]).all() and (gmax == disp[-1]).all()
        if constant:
            return dcv, None, None
    return dcv, gmin.astype(np.int64), gmax.astype(np.int64)


@_cvc.AbstractCostVolumeConfidence.register_subclass("risk")
class Risk(_cvc.AbstractCostVolumeConfidence):
    _ETA_MIN = 0.0
    _ETA_MAX = 0.7
    _ETA_STEP = 0.01
    _PERCENTILE = 1.0
    _method_max = "risk_max"
    _method_min = "risk_min"
    _method_disp_inf = "disp_inf_from_risk"
    _method_disp_sup = "disp_sup_from_risk"

    def __init__(self, **cfg):
        self.cfg = self.check_conf(**cfg)
        self._eta_min = self._ETA_MIN
        self._percentile = self._PERCENTILE
        self._eta_step = float(self.cfg["eta_step"])
        self._eta_max = float(self.cfg["eta_max"])
        self._indicator_max = self._method_max + str(self.cfg["indicator"])
        self._indicator_min = self._method_min + str(self.cfg["indicator"])
        self._indicator_disp_sup = self._method_disp_sup + str(self.cfg["indicator"])
        self._indicator_disp_inf = self._method_disp_inf + str(self.cfg["indicator"])
        self._etas = np.arange(self._eta_min, self._eta_max, self._eta_step)
        self._nbr_etas = self._etas.shape[0]

    def check_conf(self, **cfg):
        """risk.py:76-104"""
        cfg.setdefault("eta_max", self._ETA_MAX)
        cfg.setdefault("eta_step", self._ETA_STEP)
        cfg.setdefault("indicator", self._indicator)
        if cfg.get("confidence_method") != "risk":
            raise ConfigError("confidence_method must be risk")
        for key in ("eta_max", "eta_step"):
            if not isinstance(cfg[key], float) or not 0 < cfg[key] < 1:
                raise ConfigError(f"{key} must be a float in (0, 1)")
        # the device kernels keep the eta table in LDS (1024 entries); refuse at configuration time, not mid-pipeline
        if len(np.arange(0.0, cfg["eta_max"], cfg["eta_step"])) > 1024:
            raise ConfigError("pandora_amd supports at most 1024 etas: raise eta_step or lower eta_max")
        if not isinstance(cfg["indicator"], str):
            raise ConfigError("indicator must be a str")
        for key in cfg:
            if key not in ("confidence_method", "eta_max", "eta_step", "indicator"):
                raise ConfigError(f"unknown confidence key {key!r}")
        return cfg

    def desc(self):
        print("Risk method")

    def confidence_prediction(self, disp, img_left=None, img_right=None, cv=None):
        """risk.py:113-166: layers risk_max, risk_min, disp_sup_from_risk, disp_inf_from_risk (in this order)."""
        dcv, gmin, gmax = _device_volume_and_grids(cv, img_left)
        risk_max, risk_min, disp_sup, disp_inf = dcv.engine.risk(dcv, self._etas, gmin, gmax, negate=cv.attrs["type_measure"] == "max")
        disp, cv = self.allocate_confidence_map(self._indicator_max, risk_max, disp, cv)
        disp, cv = self.allocate_confidence_map(self._indicator_min, risk_min, disp, cv)
        disp, cv = self.allocate_confidence_map(self._indicator_disp_sup, disp_sup, disp, cv)
        disp, cv = self.allocate_confidence_map(self._indicator_disp_inf, disp_inf, disp, cv)
        return disp, cv
