"""AbstractDisparity + WinnerTakesAll (reference: disparity/disparity.py:42-553)."""
from abc import ABCMeta, abstractmethod

import numpy as np

from ..dataset import DataArray, Dataset
from ..criteria import LazyValidity
from ..engine import DeviceMapArray
from ..matching_cost.matching_cost import ConfigError


class AbstractDisparity:
    __metaclass__ = ABCMeta

    disparity_methods_avail = {}

    @property
    def margins(self):
        """NullMargins (the reference's default for this step)"""
        from ..margins import uniform

        return uniform(0)

    cfg = None

    def __new__(cls, **cfg):
        if cls is AbstractDisparity:
            if isinstance(cfg.get("disparity_method"), str):
                try:
                    return super(AbstractDisparity, cls).__new__(cls.disparity_methods_avail[cfg["disparity_method"]])
                except KeyError:
                    raise KeyError("No disparity method named {} supported".format(cfg["disparity_method"]))
            raise KeyError("No disparity method named {} supported".format(cfg.get("disparity_method")))
        return super(AbstractDisparity, cls).__new__(cls)

    @classmethod
    def register_subclass(cls, short_name):
        def decorator(subclass):
            cls.disparity_methods_avail[short_name] = subclass
            return subclass

        return decorator

    @abstractmethod
    def desc(self):
        """Describes the disparity method"""

    @staticmethod
    def coefficient_map(cv):
        """disparity.py:143-164: the cost of the winner of every pixel, cv["cost_volume"].sel(disp=cv["disp_indices"])."""
        vol = np.asarray(cv["cost_volume"].data)
        disp = np.asarray(cv.coords["disp"])
        idx = np.searchsorted(disp, np.asarray(cv["disp_indices"].data))
        idx = np.clip(idx, 0, len(disp) - 1)
        out = np.take_along_axis(vol, idx[:, :, None], axis=2)[:, :, 0].astype(np.float32)
        da = DataArray(out, ("row", "col"), {"row": cv.coords["row"], "col": cv.coords["col"]})
        da.name = "Coefficient Map"
        da.attrs = cv.attrs
        return da

    @abstractmethod
    def to_disp(self, cv, img_left=None, img_right=None):
        """Disparity computation and validity mask; returns the disparity dataset."""


def extract_disparity_interval_from_cost_volume(cost_volume):
    """disparity.py:301-315: DataArray [min, max] of the cost volume's disparity coordinates."""
    d = np.asarray(cost_volume.coords["disp"])
    return DataArray(np.array([d[0], d[-1]]), ("disparity",), {"disparity": ["min", "max"]})


def extract_interval_from_disparity_map(disparity_map):
    """disparity.py:318-331 -> (int min, int max)"""
    disparity_min, disparity_max = np.asarray(disparity_map["disparity_interval"].data)
    return int(disparity_min), int(disparity_max)


def extract_disparity_range_from_disparity_map(disparity_map):
    """disparity.py:334-348: np.arange(min, max + 1)"""
    disparity_min, disparity_max = extract_interval_from_disparity_map(disparity_map)
    return np.arange(disparity_min, disparity_max + 1)


@AbstractDisparity.register_subclass("wta")
class WinnerTakesAll(AbstractDisparity):
    _INVALID_DISPARITY = -9999

    def __init__(self, **cfg):
        self.cfg = self.check_conf(**cfg)
        self._invalid_disparity = self.cfg["invalid_disparity"]

    def check_conf(self, **cfg):
        if "invalid_disparity" not in cfg:
            cfg["invalid_disparity"] = self._INVALID_DISPARITY
        elif cfg["invalid_disparity"] == "NaN":
            cfg["invalid_disparity"] = np.nan
        if cfg.get("disparity_method") != "wta":
            raise ConfigError("disparity_method must be wta")
        if not isinstance(cfg["invalid_disparity"], (int, float)):
            raise ConfigError("invalid_disparity must be a number or 'NaN'")
        return cfg

    def desc(self):
        print("Winner takes all method")

    @staticmethod
    def _extremum_split(cost_volume, is_max):
        arr = cost_volume["cost_volume"]
        if not hasattr(arr, "device_cv"):
            raise TypeError("a device-resident cost volume is needed (pandora_amd has no CPU path)")
        dcv = arr.device_cv
        dcv.engine.set_validity(None)
        dcv.engine.wta(dcv, is_max, float("nan"))
        return dcv.engine.get_disparity()[0]

    @staticmethod
    def argmin_split(cost_volume):
        """disparity.py:482-516: disparity of the first minimum over D for every pixel (NaN costs never win; the reference
        expects them replaced by +inf beforehand, which is how the device kernel treats them anyway)."""
        return WinnerTakesAll._extremum_split(cost_volume, False)

    @staticmethod
    def argmax_split(cost_volume):
        """disparity.py:518-553"""
        return WinnerTakesAll._extremum_split(cost_volume, True)

    def to_disp(self, cv, img_left=None, img_right=None):
        """disparity.py:399-480: first arg-extremum over D with NaN -> +/-inf; pixels NaN for every d
        get invalid_disparity and PANDORA_MSK_PIXEL_INVALID."""
        arr = cv["cost_volume"]
        if not hasattr(arr, "device_cv"):
            raise TypeError("to_disp needs a device-resident cost volume (pandora_amd has no CPU path)")
        dcv = arr.device_cv
        eng = dcv.engine
        is_max = cv.attrs["type_measure"] == "max"
        vm = cv["validity_mask"] if "validity_mask" in cv.data_vars else None
        recipe = vm.recipe(eng) if isinstance(vm, LazyValidity) else None
        if recipe is not None:  # nobody has looked at the volume's mask: the device puts it together where the WTA needs it
            eng.compose_validity(*recipe)
        else:
            eng.set_validity(None if vm is None else vm.data)
        eng.wta(dcv, is_max, float(self._invalid_disparity))
        # the maps stay on the GPU until somebody reads them (engine.DeviceMapArray): a refinement step that follows works on them
        # where they are, a filter or the caller gets them with one download into pinned memory
        coords = {"row": cv.coords["row"], "col": cv.coords["col"]}
        disp_map = Dataset(coords=coords)
        disp_map["disparity_map"] = DeviceMapArray(eng, "disp", coords=coords)
        disp_map["disparity_interval"] = extract_disparity_interval_from_cost_volume(cv)
        cv["disp_indices"] = DeviceMapArray(eng, "disp", coords=coords)  # (the reference keeps a copy of the WTA result, disparity.py:459)
        disp_map.attrs = dict(cv.attrs)
        if "confidence_measure" in cv.data_vars:
            disp_map.coords["indicator"] = cv.coords["indicator"]
            disp_map["confidence_measure"] = cv["confidence_measure"]
        disp_map["validity_mask"] = DeviceMapArray(eng, "validity", coords=coords)
        disp_map.attrs["_device_cv"] = dcv  # lets the refinement step stay on the device
        return disp_map
