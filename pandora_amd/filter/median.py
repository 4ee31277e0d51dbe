"""MedianFilter (reference: filter/median.py:40-179), computed on the device (pmx_median_filter_disparity)."""
import numpy as np

from .. import runtime
from ..engine import DeviceMapArray
from ..matching_cost.matching_cost import ConfigError
from . import filter as _filter


@_filter.AbstractFilter.register_subclass("median")
class MedianFilter(_filter.AbstractFilter):
    _FILTER_SIZE = 3

    def __init__(self, *args, cfg=None, step=1, **kwargs):
        self.cfg = self.check_conf(dict(cfg or {}))
        self._filter_size = int(self.cfg["filter_size"])
        self._step = step

    @property
    def margins(self):
        """median.py:61-64"""
        from ..margins import uniform

        return uniform(self._filter_size * self._step)

    def check_conf(self, cfg):
        """median.py:68-90"""
        if "filter_size" not in cfg:
            cfg["filter_size"] = self._FILTER_SIZE
        if cfg.get("filter_method") != "median":
            raise ConfigError("filter_method must be median")
        fs = cfg["filter_size"]
        if isinstance(fs, bool) or not isinstance(fs, int) or fs < 1 or fs % 2 == 0:
            raise ConfigError("filter_size must be an odd integer >= 1")
        if fs > 15:  # pmx_median_filter_disparity sorts the window in registers
            raise ConfigError("pandora_amd supports median filter_size up to 15")
        for key in cfg:
            if key not in ("filter_method", "filter_size"):
                raise ConfigError(f"unknown filter key {key!r}")
        return cfg

    def desc(self):
        print("Median filter description")

    def filter_disparity(self, disp, img_left=None, img_right=None, cv=None):
        """median.py:94-131: median over the valid pixels, invalid neighbours ignored, in place."""
        eng = runtime.get_engine()
        dm, vm = disp["disparity_map"], disp["validity_mask"]
        snaps = [m.device_snapshot() if isinstance(m, DeviceMapArray) and m.engine is eng and m.shape == (eng.H, eng.W) else None
                 for m in (dm, vm)]
        if snaps[0] is not None and snaps[1] is not None:  # the maps never left the GPU: neither does the filtered one
            out = eng.median_filter_maps(snaps[0], snaps[1], self._filter_size)
            disp["disparity_map"] = DeviceMapArray.from_snapshot(eng, "disp", out, coords=dm.coords, dims=dm.dims)
        else:
            dm.data = eng.median_filter_disparity(np.asarray(dm.data), np.asarray(vm.data), self._filter_size)
        disp.attrs["filter"] = "median"
