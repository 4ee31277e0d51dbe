"""Ambiguity (reference: cost_volume_confidence/ambiguity.py:38-248); the integral over etas runs on the device-resident
cost volume (pmx_ambiguity), normalisation and bookkeeping stay on the host."""
import logging

import numpy as np

from ..matching_cost.matching_cost import ConfigError
from . import cost_volume_confidence as _cvc
from .risk import _device_volume_and_grids


def percentiles(values, qs):
    """np.percentile(values, q) for every q of ``qs`` without numpy's partition of the whole map (30 ms per call at 4 Mpx): the two
    order statistics around each percentile's virtual index come from the device (pmx_order_statistics, radix selection - exact,
    they are elements of the map), the interpolation between them is numpy's own, on the two-element array (same dtype rules,
    same _lerp, same gamma).  A map with NaNs, or a small one, goes to np.percentile as it is."""
    from .. import runtime

    a = np.asarray(values)
    n = a.size
    if a.dtype != np.float32 or n < (1 << 16):
        return [np.percentile(a, q) for q in qs]
    # numpy's "linear" method: virtual index (n - 1) * q / 100.  Under numpy 2 (NEP 50: python scalars are weak) np.percentile
    # divides q by a.dtype.type(100) and the index of a float32 map is itself a float32 (a 4 Mpx map's gamma is good to 1/4
    # only); under numpy 1's value-based casting it is a float64.  Mirrored - whatever the installed numpy does, so that small
    # maps (np.percentile itself) and large ones agree and the reference's normalisation bounds come out on its numpy too.
    if int(np.__version__.split(".")[0]) >= 2:
        virtual = [(n - 1) * np.true_divide(q, a.dtype.type(100)) for q in qs]
    else:
        virtual = [(n - 1) * (np.float64(q) / 100.0) for q in qs]
    lows = [int(np.floor(v)) for v in virtual]
    ranks = sorted({r for lo in lows for r in (lo, min(lo + 1, n - 1))} | {n - 1})
    stats = dict(zip(ranks, runtime.get_engine().order_statistics(a, ranks)))
    if np.isnan(stats[n - 1]):  # NaNs sort last: numpy's answer is NaN then
        return [np.percentile(a, q) for q in qs]
    return [np.quantile(np.array([stats[lo], stats[min(lo + 1, n - 1)]], np.float32), v - lo) for v, lo in zip(virtual, lows)]


@_cvc.AbstractCostVolumeConfidence.register_subclass("ambiguity")
class Ambiguity(_cvc.AbstractCostVolumeConfidence):
    _ETA_MIN = 0.0
    _ETA_MAX = 0.7
    _ETA_STEP = 0.01
    _PERCENTILE = 1.0
    _NORMALIZATION = True
    _method = "ambiguity"

    def __init__(self, **cfg):
        self.cfg = self.check_conf(**cfg)
        self._eta_min = self._ETA_MIN
        self._percentile = self._PERCENTILE
        self._normalization = self.cfg["normalization"]
        self._eta_max = float(self.cfg["eta_max"])
        self._eta_step = float(self.cfg["eta_step"])
        self._indicator = self._method + str(self.cfg["indicator"])
        self._etas = np.arange(self._eta_min, self._eta_max, self._eta_step)
        self._nbr_etas = self._etas.shape[0]

    def check_conf(self, **cfg):
        """ambiguity.py:76-104"""
        cfg.setdefault("eta_max", self._ETA_MAX)
        cfg.setdefault("eta_step", self._ETA_STEP)
        cfg.setdefault("indicator", self._indicator)
        cfg.setdefault("normalization", self._NORMALIZATION)
        if cfg.get("confidence_method") != "ambiguity":
            raise ConfigError("confidence_method must be ambiguity")
        for key in ("eta_max", "eta_step"):
            if not isinstance(cfg[key], float) or not 0 < cfg[key] < 1:
                raise ConfigError(f"{key} must be a float in (0, 1)")
        # the device kernels keep the eta table in LDS (1024 entries); refuse at configuration time, not mid-pipeline
        if len(np.arange(0.0, cfg["eta_max"], cfg["eta_step"])) > 1024:
            raise ConfigError("pandora_amd supports at most 1024 etas: raise eta_step or lower eta_max")
        if not isinstance(cfg["normalization"], bool) or not isinstance(cfg["indicator"], str):
            raise ConfigError("normalization must be a bool and indicator a str")
        for key in cfg:
            if key not in ("confidence_method", "eta_max", "eta_step", "indicator", "normalization"):
                raise ConfigError(f"unknown confidence key {key!r}")
        return cfg

    def desc(self):
        print("Ambiguity confidence method")

    def confidence_prediction(self, disp, img_left=None, img_right=None, cv=None):
        """ambiguity.py:113-166"""
        dcv, gmin, gmax = _device_volume_and_grids(cv, img_left)
        ambiguity = dcv.engine.ambiguity(dcv, self._etas, gmin, gmax, negate=cv.attrs["type_measure"] == "max")
        if self._normalization:
            if "global_disparity" in img_left.attrs:
                ambiguity = self.normalize_with_extremum(ambiguity, img_left, self._nbr_etas, cv.attrs["subpixel"])
                logging.info("You are not using ambiguity normalization by percentile; \\n"
                             "you are in a specific case with the instantiation of global_disparity.")
            elif img_right is not None and "global_disparity" in img_right.attrs:
                ambiguity = self.normalize_with_extremum(ambiguity, img_right, self._nbr_etas, cv.attrs["subpixel"])
            else:
                ambiguity = self.normalize_with_percentile(ambiguity)
        ambiguity = 1 - ambiguity
        return self.allocate_confidence_map(self._indicator, ambiguity, disp, cv)

    def normalize_with_percentile(self, ambiguity):
        """ambiguity.py:168-184"""
        norm_amb = np.copy(ambiguity)
        perc_min, perc_max = percentiles(norm_amb, (self._percentile, 100 - self._percentile))
        np.clip(norm_amb, perc_min, perc_max, out=norm_amb)
        return (norm_amb - np.min(norm_amb)) / (np.max(norm_amb) - np.min(norm_amb))
