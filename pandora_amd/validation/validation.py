"""AbstractValidation + CrossCheckingAccurate (reference: validation/validation.py:40-371).

Same registry mechanics, configuration keys and dataset protocol as the reference; the consistency check itself
runs on the GPU (pmx_cross_checking).  The interpolation of rejected pixels lives in interpolated_disparity.py.
"""
from abc import ABCMeta, abstractmethod

import numpy as np

from .. import runtime
from ..criteria import mask_border
from ..dataset import DataArray
from ..engine import DeviceMapArray
from ..matching_cost.matching_cost import ConfigError


def allocate_confidence_map(name_confidence_measure, confidence_map, disp, cv):
    """cost_volume_confidence.py:141-246 (AbstractCostVolumeConfidence.allocate_confidence_map): append one
    indicator layer to ``confidence_measure`` of the cost volume and/or the disparity dataset."""
    if "disp_min" not in name_confidence_measure and "disp_max" not in name_confidence_measure:
        name_confidence_measure = "confidence_from_" + name_confidence_measure
    layer = np.asarray(confidence_map, np.float32)

    def extend(ds):
        if "confidence_measure" in ds.data_vars:
            old = np.asarray(ds["confidence_measure"].data)
            data = np.full(old.shape[:2] + (old.shape[2] + 1,), np.nan, np.float32)
            data[:, :, :-1] = old
            data[:, :, -1] = layer
            indicator = np.append(np.copy(ds.coords["indicator"]), name_confidence_measure)
        else:
            data = layer[:, :, np.newaxis]  # (a view: the layer's array is the indicator's storage, no 16 MB copy at 4 Mpx)
            indicator = np.array([name_confidence_measure])
        ds.coords["indicator"] = indicator
        coords = {k: ds.coords[k] for k in ("row", "col") if k in ds.coords}  # (the machine's pre-disparity dataset has none)
        coords["indicator"] = indicator
        ds.data_vars["confidence_measure"] = DataArray(data, ("row", "col", "indicator"), coords)

    if cv is not None:
        extend(cv)
    if disp is not None:
        if "confidence_measure" in disp.data_vars or cv is None:
            extend(disp)
        else:
            disp.coords["indicator"] = cv.coords["indicator"]
            disp.data_vars["confidence_measure"] = cv["confidence_measure"]
    return disp, cv


class AbstractValidation:
    __metaclass__ = ABCMeta

    validation_methods_avail = {}
    cfg = None

    def __new__(cls, **cfg):
        if cls is AbstractValidation:
            if isinstance(cfg.get("validation_method"), str):
                try:
                    return super(AbstractValidation, cls).__new__(cls.validation_methods_avail[cfg["validation_method"]])
                except KeyError:
                    raise KeyError("No validation method named {} supported".format(cfg["validation_method"]))
            raise KeyError("No validation method named {} supported".format(cfg.get("validation_method")))
        return super(AbstractValidation, cls).__new__(cls)

    @classmethod
    def register_subclass(cls, short_name):
        def decorator(subclass):
            cls.validation_methods_avail[short_name] = subclass
            return subclass

        return decorator

    @abstractmethod
    def desc(self):
        """Describes the validation method"""

    @abstractmethod
    def disparity_checking(self, dataset_left, dataset_right, img_left=None, img_right=None, cv=None):
        """Occlusions and false matches by a consistency check on valid pixels; updates the validity mask."""


@AbstractValidation.register_subclass("cross_checking_accurate")
@AbstractValidation.register_subclass("cross_checking_fast")
class CrossCheckingAccurate(AbstractValidation):
    _THRESHOLD = 1.0

    def __init__(self, **cfg):
        self.cfg = self.check_conf(**cfg)
        self._threshold = self.cfg["cross_checking_threshold"]
        self._method = self.cfg["validation_method"]

    def check_conf(self, **cfg):
        """validation.py:196-217"""
        if "cross_checking_threshold" not in cfg:
            cfg["cross_checking_threshold"] = self._THRESHOLD
        if cfg.get("validation_method") not in ("cross_checking_accurate", "cross_checking_fast"):
            raise ConfigError("validation_method must be cross_checking_accurate or cross_checking_fast")
        if isinstance(cfg["cross_checking_threshold"], bool) or not isinstance(cfg["cross_checking_threshold"], (int, float)):
            raise ConfigError("cross_checking_threshold must be a number")
        if "interpolated_disparity" in cfg and cfg["interpolated_disparity"] not in ("mc-cnn", "sgm"):
            raise ConfigError("interpolated_disparity must be mc-cnn or sgm")
        for key in cfg:
            if key not in ("validation_method", "cross_checking_threshold", "interpolated_disparity"):
                raise ConfigError(f"unknown validation key {key!r}")
        return cfg

    def desc(self):
        print("Cross-checking method")

    def disparity_checking(self, dataset_left, dataset_right, img_left=None, img_right=None, cv=None):
        """validation.py:226-371.  Updates dataset_left["validity_mask"] (PANDORA_MSK_PIXEL_OCCLUSION /
        _MISMATCH), appends the left-right distance as "confidence_from_left_right_consistency"."""
        interval = np.asarray(dataset_left["disparity_interval"].data)
        dmin, dmax = int(interval[0]), int(interval[1])  # np.arange(disparity_min, disparity_max + 1), disparity.py:334-347
        eng = runtime.get_engine()
        maps = (dataset_left["disparity_map"], dataset_left["validity_mask"], dataset_right["disparity_map"])
        snaps = [m.device_snapshot() if isinstance(m, DeviceMapArray) and m.engine is eng and m.shape == (eng.H, eng.W) else None
                 for m in maps]
        if all(s is not None for s in snaps):
            # both sides' maps are still in HBM: checked there, the left mask updated (and framed, criteria.mask_border) in its own
            # snapshot; only the left-right distance comes down, into the confidence_measure array the reference stacks it on
            border = int(dataset_left.attrs.get("offset_row_col", 0))
            conf_snap = eng.cross_checking_maps(snaps[0], snaps[1], snaps[2], dmin, dmax, float(self._threshold), border)
            try:
                conf = eng.read_snapshot(conf_snap, "conf", maps[0].shape)
            finally:
                eng.free_snapshot(conf_snap)
            dataset_left.attrs["validation"] = self._method
            dataset_left, _ = allocate_confidence_map("left_right_consistency", conf, dataset_left, cv)
            return dataset_left
        validity, conf = eng.cross_checking(dataset_left["disparity_map"].data, dataset_left["validity_mask"].data,
                                            dataset_right["disparity_map"].data, dmin, dmax, float(self._threshold))
        dataset_left["validity_mask"].data = validity.astype(np.asarray(dataset_left["validity_mask"].data).dtype, copy=False)
        dataset_left.attrs["validation"] = self._method
        dataset_left, _ = allocate_confidence_map("left_right_consistency", conf, dataset_left, cv)
        if dataset_left.attrs.get("offset_row_col", 0) > 0:
            dataset_left["validity_mask"] = mask_border(dataset_left)
        return dataset_left
