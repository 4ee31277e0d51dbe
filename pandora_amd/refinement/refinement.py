"""AbstractRefinement + vfit / quadratic (reference: refinement/refinement.py:38-260, vfit.py,
quadratic.py and refinement/cpp/src/*.cpp)."""
from abc import ABCMeta

import numpy as np

from ..dataset import DataArray
from ..engine import DeviceMapArray
from ..matching_cost.matching_cost import ConfigError


class AbstractRefinement:
    __metaclass__ = ABCMeta

    subpixel_methods_avail = {}

    @property
    def margins(self):
        """NullMargins (the reference's default for this step)"""
        from ..margins import uniform

        return uniform(0)

    cfg = None
    _refinement_method_name = None

    def __new__(cls, **cfg):
        if cls is AbstractRefinement:
            if isinstance(cfg.get("refinement_method"), str):
                try:
                    return super(AbstractRefinement, cls).__new__(cls.subpixel_methods_avail[cfg["refinement_method"]])
                except KeyError:
                    raise KeyError("No refinement method named {} supported".format(cfg["refinement_method"]))
            raise KeyError("No refinement method named {} supported".format(cfg.get("refinement_method")))
        return super(AbstractRefinement, cls).__new__(cls)

    @classmethod
    def register_subclass(cls, short_name):
        def decorator(subclass):
            cls.subpixel_methods_avail[short_name] = subclass
            return subclass

        return decorator

    def desc(self):
        print(f"{self._refinement_method_name} refinement method")

    def subpixel_refinement(self, cv, disp):
        """refinement.py:77-122: in place on disp["disparity_map"], disp["validity_mask"]; adds
        disp["interpolated_coeff"]."""
        arr = cv["cost_volume"]
        if not hasattr(arr, "device_cv"):
            raise TypeError("subpixel_refinement needs a device-resident cost volume (pandora_amd has no CPU path)")
        dcv = arr.device_cv
        eng = dcv.engine
        is_max = cv.attrs["type_measure"] == "max"
        dm, vm = disp["disparity_map"], disp["validity_mask"]
        coords = {k: disp.coords[k] for k in ("row", "col") if k in disp.coords}
        if isinstance(dm, DeviceMapArray) and isinstance(vm, DeviceMapArray) and dm.engine is eng and dm.on_device() and vm.on_device():
            # straight after WTA: the maps never left the GPU and this step updates exactly these two variables in place
            eng.refine(dcv, self._refinement_method_name, is_max, superseded=(dm, vm))
            dm.rebind()
            vm.rebind()
        else:
            snaps = [m.device_snapshot() if isinstance(m, DeviceMapArray) and m.engine is eng else None for m in (dm, vm)]
            if snaps[0] is not None and snaps[1] is not None:
                # the maps left the engine's buffers for snapshots of their own (the other side's WTA came in between): they go
                # back device to device
                eng.maps_restore(snaps[0], snaps[1])
            else:
                # the host copies may have been edited since WTA (filters): they are the source of truth
                eng.set_disparity(np.asarray(dm.data, np.float32), np.asarray(vm.data, np.int64))
            eng.refine(dcv, self._refinement_method_name, is_max)
            disp["disparity_map"] = DeviceMapArray(eng, "disp", coords=coords)
            disp["validity_mask"] = DeviceMapArray(eng, "validity", coords=coords)
        disp.attrs["refinement"] = self._refinement_method_name
        disp["interpolated_coeff"] = DeviceMapArray(eng, "itp", coords=coords)


    def approximate_subpixel_refinement(self, cv_left, disp_right):
        """refinement.py:124-158: the right map of the approximate ("fast") right-side route - a diagonal search of the LEFT cost
        volume - refined on that volume's diagonals (refinement_cpp.loop_approximate_refinement).  In place on
        disp_right["disparity_map"] / ["validity_mask"]; adds ["interpolated_coeff"]."""
        arr = cv_left["cost_volume"]
        if not hasattr(arr, "device_cv"):
            raise TypeError("approximate_subpixel_refinement needs a device-resident cost volume (pandora_amd has no CPU path)")
        dcv = arr.device_cv
        eng = dcv.engine
        is_max = cv_left.attrs["type_measure"] == "max"
        dm, vm = disp_right["disparity_map"], disp_right["validity_mask"]
        coords = {k: disp_right.coords[k] for k in ("row", "col") if k in disp_right.coords}
        snaps = [m.device_snapshot() if isinstance(m, DeviceMapArray) and m.engine is eng else None for m in (dm, vm)]
        if snaps[0] is not None and snaps[1] is not None:
            eng.maps_restore(snaps[0], snaps[1])  # (device-resident right maps go back device to device, as in subpixel_refinement)
        else:
            eng.set_disparity(np.asarray(dm.data, np.float32), np.asarray(vm.data, np.int64))
        eng.refine_approximate(dcv, self._refinement_method_name, is_max)
        disp_right["disparity_map"] = DeviceMapArray(eng, "disp", coords=coords)
        disp_right["validity_mask"] = DeviceMapArray(eng, "validity", coords=coords)
        disp_right.attrs["refinement"] = self._refinement_method_name
        disp_right["interpolated_coeff"] = DeviceMapArray(eng, "itp", coords=coords)
        return disp_right


def _simple_conf(name):
    def check_conf(**cfg):
        if cfg.get("refinement_method") != name:
            raise ConfigError(f"refinement_method must be {name}")
        return cfg

    return staticmethod(check_conf)


@AbstractRefinement.register_subclass("vfit")
class Vfit(AbstractRefinement):
    check_conf = _simple_conf("vfit")

    def __init__(self, **cfg):
        self.cfg = self.check_conf(**cfg)
        self._refinement_method_name = str(self.cfg["refinement_method"])


@AbstractRefinement.register_subclass("quadratic")
class Quadratic(AbstractRefinement):
    check_conf = _simple_conf("quadratic")

    def __init__(self, **cfg):
        self.cfg = self.check_conf(**cfg)
        self._refinement_method_name = str(self.cfg["refinement_method"])
