"""AbstractMatchingCost - the reference's matching-cost plugin base re-stated for the MI355X engine.

Same registry mechanics, method names, argument meaning and error behaviour as
/root/reference/src/pandora/matching_cost/matching_cost.py:45-950 (register_subclass :109-131,
__new__ dispatch :80-107, check_conf :158-184, allocate_cost_volume :377-407, cv_masked :770-872),
but the cost volume lives in HBM (dataset.DeviceVolumeArray) and every O(H*W*D) operation is a HIP
kernel behind include/pandora_amd.h.  There is no CPU path.
"""
from abc import ABCMeta, abstractmethod

import numpy as np

from .. import criteria, runtime
from ..dataset import DataArray, Dataset, DeviceVolumeArray


class ConfigError(ValueError):
    """Raised where the reference's json_checker schema would reject a configuration."""


def grid_extrema(grid, memo=None):
    """(min, max) of an integer disparity grid; int64 C-contiguous grids (what create_dataset_from_inputs makes of a [min, max]
    list) in one pass on a few host threads (pmx_host_minmax_i64), anything else with numpy.  ``memo``: a dict the caller keeps
    for as long as it KNOWS the grids cannot change (PandoraMachine: from matching_cost_prepare to the end of matching_cost_run,
    one trigger, no user code in between) - a grid seen before (same memory, shape, strides) is not scanned again."""
    if memo is not None:
        key = (grid.__array_interface__["data"][0], grid.shape, grid.strides, grid.dtype.str)
        if key not in memo:
            memo[key] = (grid_extrema(grid), grid)  # (the reference to the grid keeps its memory from being handed to another array)
        return memo[key][0]
    if grid.dtype == np.int64 and grid.flags["C_CONTIGUOUS"] and grid.size:
        import ctypes as C

        from .. import _lib

        lo, hi = C.c_int64(), C.c_int64()
        rc = _lib.lib().pmx_host_minmax_i64(grid.ctypes.data_as(_lib.c_i64_p), grid.size, C.byref(lo), C.byref(hi))
        if rc == 0:
            return lo.value, hi.value
    return int(grid.min()), int(grid.max())


class AbstractMatchingCost:
    __metaclass__ = ABCMeta

    matching_cost_methods_avail = {}
    cfg = None
    _WINDOW_SIZE = 5
    _SUBPIX = 1
    _BAND = None
    _STEP_COL = 1
    _SPLINE_ORDER = 1

    def __new__(cls, **cfg):
        if cls is AbstractMatchingCost:
            if isinstance(cfg.get("matching_cost_method"), str):
                try:
                    return super(AbstractMatchingCost, cls).__new__(cls.matching_cost_methods_avail[cfg["matching_cost_method"]])
                except KeyError:
                    raise KeyError("No matching cost method named {} supported".format(cfg["matching_cost_method"]))
            raise KeyError("No matching cost method named {} supported".format(cfg.get("matching_cost_method")))
        return super(AbstractMatchingCost, cls).__new__(cls)

    @classmethod
    def register_subclass(cls, short_name, *args):
        def decorator(subclass):
            cls.matching_cost_methods_avail[short_name] = subclass
            for arg in args:
                cls.matching_cost_methods_avail[arg] = subclass
            return subclass

        return decorator

    def desc(self):
        print(f"{self._method} similarity measure")

    # -- configuration (matching_cost.py:140-184) ----------------------------------------------
    def instantiate_class(self, **cfg):
        self.cfg = self.check_conf(**cfg)
        self._window_size = int(self.cfg["window_size"])
        self._subpix = int(self.cfg["subpix"])
        self._band = self.cfg["band"]
        self._step_col = int(self.cfg["step"])
        self._method = str(self.cfg["matching_cost_method"])
        self._spline_order = int(self.cfg["spline_order"])
        del self.cfg["spline_order"]

    def check_conf(self, **cfg):
        cfg.setdefault("window_size", self._WINDOW_SIZE)
        cfg.setdefault("subpix", self._SUBPIX)
        cfg.setdefault("band", self._BAND)
        if "step" in cfg and cfg["step"] != 1:
            raise ValueError("Step parameter cannot be different from 1")
        cfg.setdefault("step", self._STEP_COL)
        cfg.setdefault("spline_order", self._SPLINE_ORDER)
        if not isinstance(cfg["subpix"], int) or cfg["subpix"] not in (1, 2, 4):
            raise ConfigError("subpix must be 1, 2 or 4")
        if not (cfg["band"] is None or isinstance(cfg["band"], str)):
            raise ConfigError("band must be a string or None")
        if not isinstance(cfg["spline_order"], int) or not 1 <= cfg["spline_order"] <= 5:
            raise ConfigError("spline_order must be an int in [1, 5]")
        return cfg

    def check_band_input_mc(self, img_left, img_right):
        """matching_cost.py:186-231: the "band" parameter against the bands of the two images."""
        def bands(ds):
            return list(ds.coords["band_im"]) if "band_im" in ds.coords else None

        left, right = bands(img_left), bands(img_right)
        if self._band is not None:
            if right is None:
                raise AttributeError(f"Right dataset is monoband: {self._band} band cannot be selected")
            if left is None:
                raise AttributeError(f"Left dataset is monoband: {self._band} band cannot be selected")
            if self._band not in right or self._band not in left:
                raise AttributeError(f"Wrong band instantiate : {self._band} not in img_left or img_right")
        elif left is not None and right is not None:
            raise AttributeError("Band must be instantiated in matching cost step")

    @property
    def margins(self):
        """HalfWindowMargins (matching_cost.py:76, margins/descriptors.py:88-114)"""
        from ..margins import uniform

        return uniform(int((self._window_size - 1) / 2))

    @property
    def margins_value(self):
        return self.margins.astuple()

    # -- geometry (matching_cost.py:330-427, 604-616) ------------------------------------------
    @staticmethod
    def get_min_max_from_grid(disp_min, disp_max, memo=None):
        if disp_min.dtype == np.int64 and disp_max.dtype == np.int64 and disp_min.size and disp_max.size:
            return grid_extrema(disp_min, memo)[0], grid_extrema(disp_max, memo)[1]  # (integers hold no NaN)
        return int(np.nanmin(disp_min)), int(np.nanmax(disp_max))

    @staticmethod
    def get_disparity_range(disparity_min, disparity_max, subpix):
        if subpix == 1:
            return np.arange(disparity_min, disparity_max + 1)
        rng = np.arange(disparity_min, disparity_max, 1 / float(subpix), dtype=np.float64)
        return np.append(rng, [disparity_max])

    @staticmethod
    def find_nearest_multiple_of_step(value, step):
        """matching_cost.py:618-632: the nearest multiple of step that is >= value"""
        return -(-value // step) * step

    @staticmethod
    def get_coordinates(margin, img_coordinates, step):
        """matching_cost.py:269-328: the columns to compute for a ROI read with `margin` columns on its left: every step-th column,
        phased so that the ROI's first column is one of them."""
        first, last = img_coordinates[0], img_coordinates[-1]
        if margin % step == 0:
            start = 0
        elif margin < step:
            start = margin
        else:
            start = step - (AbstractMatchingCost.find_nearest_multiple_of_step(margin, step) - margin)
        return np.arange(first + start, last + 1, step)

    def grid_estimation(self, img, cfg, disparity_grids):
        """matching_cost.py:330-375: the dataset (coords row / col / disp, the image's attrs, sampling_interval,
        col_to_compute) that will hold the cost volume."""
        c_col = np.asarray(img.coords["col"])
        if cfg and "ROI" in cfg:  # matching_cost.py:353-354: start so that the ROI's first column is computed
            index_compute_col = self.get_coordinates(cfg["ROI"]["margins"][0], c_col, self._step_col)
        else:
            index_compute_col = np.arange(c_col[0], c_col[-1] + 1, self._step_col)
        grids = [np.asarray(g.data if hasattr(g, "data") and not isinstance(g, np.ndarray) else g) for g in disparity_grids]
        disparity_min, disparity_max = self.get_min_max_from_grid(grids[0], grids[1], getattr(self, "_grid_memo", None))
        disparity_range = self.get_disparity_range(disparity_min, disparity_max, self._subpix)
        grid = Dataset(coords={"row": img.coords["row"], "col": index_compute_col, "disp": disparity_range}, attrs=dict(img.attrs))
        grid.attrs["sampling_interval"] = self._step_col
        grid.attrs["col_to_compute"] = index_compute_col
        return grid

    def allocate_cost_volume(self, image, disparity_grids, cfg=None):
        """matching_cost.py:377-407: grid_estimation + the attrs later steps read.  The reference fills a NaN float32
        (row, col, disp) array here; this build allocates the volume in HBM when compute_cost_volume binds it to the pair."""
        cv = self.grid_estimation(image, cfg, disparity_grids)
        cv.attrs.update({"window_size": self._window_size, "subpixel": self._subpix, "band_correl": self._band,
                         "offset_row_col": int((self._window_size - 1) / 2), "measure": self._method})
        disparity_range = np.asarray(cv.coords["disp"])
        cv.attrs["_d0"] = int(disparity_range[0])
        cv.attrs["_D"] = len(disparity_range)
        return cv

    def prefetch(self, img_left, img_right):
        """The machine's head start (no reference counterpart): make the pair resident BEFORE the cost volume is sized from the
        disparity grids, so that the transfer of the images (queued, pmx_set_images) runs while the host scans the grids.  The
        compute_cost_volume call that follows recognises the arrays it was given and does not fingerprint them again."""
        self.check_band_input_mc(img_left, img_right)
        eng = runtime.ensure_pair(img_left, img_right, self._subpix, band=self._band, spline_order=self._spline_order)
        self._prefetched = (runtime.resident_token(eng), id(img_left["im"].data), id(img_right["im"].data))

    def _bind_device_volume(self, img_left, img_right, cost_volume):
        self.check_band_input_mc(img_left, img_right)
        token, self._prefetched = getattr(self, "_prefetched", None), None  # one shot
        eng = runtime.get_engine()
        if not (token and token[0] is not None and runtime.resident_token(eng) is token[0]
                and (id(img_left["im"].data), id(img_right["im"].data)) == token[1:]):
            eng = runtime.ensure_pair(img_left, img_right, self._subpix, band=self._band, spline_order=self._spline_order)
        dcv = eng.alloc_cv(cost_volume.attrs["_D"], cost_volume.attrs["_d0"])
        cost_volume.attrs["_pair_token"] = runtime.resident_token(eng)
        cost_volume.data_vars["cost_volume"] = DeviceVolumeArray(dcv, {k: cost_volume.coords[k] for k in ("row", "col", "disp")})
        return eng, dcv

    # -- right-side helpers of the validation step ---------------------------------
    @staticmethod
    def reverse_disp_range(left_min, left_max):
        """matching_cost_cpp.reverse_disp_range (matching_cost.cpp:59-132) on the device."""
        return runtime.get_engine().reverse_disp_range(np.asarray(left_min, np.float32), np.asarray(left_max, np.float32))

    @abstractmethod
    def point_interval(self, img_left, img_right, disp):
        """matching_cost.py:429-482: the column ranges ((p0, p1), (q0, q1)) of the left / right image over which the measure is
        applied for the disparity ``disp`` (floating disparities round away from the image, an out-of-image disparity gives the
        empty range (nx, nx)).  The kernels apply the same overlap rule per cell; this is the host-side statement of it."""
        from math import ceil, floor

        nx_left, nx_right = int(img_left.sizes["col"]), int(img_right.sizes["col"])
        point_p = (nx_left, nx_left) if abs(disp) > nx_left else (max(0 - disp, 0), min(nx_left - disp, nx_left))
        point_q = (nx_right, nx_right) if abs(disp) > nx_right else (max(0 + disp, 0), min(nx_right + disp, nx_right))
        rnd = ceil if disp < 0 else floor
        return (int(rnd(point_p[0])), int(rnd(point_p[1]))), (int(rnd(point_q[0])), int(rnd(point_q[1])))

    def compute_cost_volume(self, img_left, img_right, cost_volume):
        """Fill cost_volume["cost_volume"] for the pair; returns the dataset."""

    # -- masking (matching_cost.py:770-872) ----------------------------------------------------
    def cv_masked(self, img_left, img_right, cost_volume, disp_min, disp_max):
        """In place: NaN for invalid / (dilated) no-data pixels and for disparities outside the
        per-pixel [disp_min, disp_max]; then the validity-mask updates of criteria.py:291-353."""
        eng = runtime.pair_engine(cost_volume, img_left, img_right, self._subpix, band=self._band, spline_order=self._spline_order)
        dcv = cost_volume["cost_volume"].device_cv
        for side in (img_left, img_right):
            if "msk" in side.data_vars and (side.attrs.get("valid_pixels", 0) != img_left.attrs.get("valid_pixels", 0)
                                            or side.attrs.get("no_data_mask", 1) != img_left.attrs.get("no_data_mask", 1)):
                raise ConfigError("left and right images must share one mask convention (valid_pixels / no_data_mask)")
        disp_min = np.asarray(disp_min.data if hasattr(disp_min, "data") and not isinstance(disp_min, np.ndarray) else disp_min)
        disp_max = np.asarray(disp_max.data if hasattr(disp_max, "data") and not isinstance(disp_max, np.ndarray) else disp_max)
        ny_, nx_, _ = dcv.shape
        disp_min, disp_max = disp_min[:ny_, :nx_], disp_max[:ny_, :nx_]
        coords = np.asarray(cost_volume.coords["disp"])
        # constant grids that cover the whole volume need no per-pixel range test on the device; integer grids (the usual
        # case) cannot hold NaN, so plain min / max do
        if disp_min.dtype.kind in "iu" and disp_max.dtype.kind in "iu":
            memo = getattr(self, "_grid_memo", None)
            (lo0, lo1), (hi0, hi1) = grid_extrema(disp_min, memo), grid_extrema(disp_max, memo)
            uniform = lo0 == lo1 and hi0 == hi1 and lo0 <= coords[0] and hi0 >= coords[-1]
        else:
            uniform = (np.nanmin(disp_min) == np.nanmax(disp_min) and np.nanmin(disp_max) == np.nanmax(disp_max)
                       and disp_min.flat[0] <= coords[0] and disp_max.flat[0] >= coords[-1] and not np.isnan(disp_min).any())
        eng.set_disparity_grids(None, None) if uniform else eng.set_disparity_grids(disp_min, disp_max)
        eng.cv_masked(dcv, self._window_size)
        if "validity_mask" in cost_volume.data_vars:
            criteria.mask_invalid_variable_disparity_range(cost_volume)
            if cost_volume.attrs["offset_row_col"] > 0:
                criteria.mask_border(cost_volume)

    @staticmethod
    def reverse_cost_volume(left_cv, disp_min):
        """matching_cost.py:920-934 -> matching_cost_cpp.reverse_cost_volume (matching_cost.cpp:26-56):
        right(i, j, d) = left(i, j + d + disp_min, D-1-d).  left_cv is the device-resident cost-volume VARIABLE
        (cv["cost_volume"]); the result is a new one."""
        if not hasattr(left_cv, "device_cv"):
            raise TypeError("reverse_cost_volume needs a device-resident cost volume (pandora_amd has no CPU path)")
        dcv = left_cv.device_cv
        return DeviceVolumeArray(dcv.engine.reverse_cost_volume(dcv, int(disp_min)), dict(left_cv.coords))
