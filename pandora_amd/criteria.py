"""Validity-mask criteria - host-side (numpy) mirror of the reference's criteria.py.

These are O(H*W) bookkeeping around the hot path (the O(H*W*D) work they depend on - which
pixels are NaN for every disparity - is computed on the GPU, Engine.nan_pixels).
Citations: /root/reference/src/pandora/criteria.py.
"""
import numpy as np

from . import constants as cst
from .dataset import DataArray


class LazyValidity:
    """``cv["validity_mask"]`` (criteria.py:66-158) for as long as nobody looks at it.  The machine only ever hands this mask to
    the disparity step, after mask_invalid_variable_disparity_range and mask_border have updated it; building it (a 4 Mpx int64
    map is 32 MB), fetching the all-NaN pixels from the GPU to OR them in and uploading the result is host work the device does in
    one small kernel (pmx_compose_validity).  So the mask is kept as its recipe - the base (one line of flags for every row when no
    input mask takes part, else the full map) plus the deferred updates - and ``.data`` carries the recipe out on the host, with
    the same functions as before, the first time anybody reads it; from then on it is a plain array."""
    dims = ("row", "col")

    def __init__(self, base, shape, coords=None):
        self._base = base
        self._shape = tuple(shape)
        self.coords = dict(coords or {})
        self._ops = []  # ("missing", device volume) | ("border", offset), in the order they were asked for
        self._host = None

    @property
    def pending(self):
        return self._host is None

    def defer(self, op, arg):
        self._ops.append((op, arg))

    def recipe(self, engine):
        """(base, volume whose missing-range snapshot is ORed in | None, border) when the deferred updates are the machine's
        sequence - at most one of each, in that order, on this engine -, else None (the caller reads ``.data``)."""
        if self._host is not None:
            return None
        ops, dcv, border = list(self._ops), None, 0
        if ops and ops[0][0] == "missing":
            dcv = ops.pop(0)[1]
        if ops and ops[0][0] == "border":
            border = ops.pop(0)[1]
        if ops or (dcv is not None and dcv.engine is not engine):
            return None
        return self._base, dcv, border

    @property
    def data(self):
        if self._host is None:
            if self._base.shape == self._shape:
                vm = self._base
            else:
                vm = np.empty(self._shape, np.int64)
                vm[:] = self._base
            ops, self._ops, self._base, self._host = self._ops, [], None, vm
            for op, arg in ops:
                if op == "missing":
                    _or_missing(vm, arg.engine.get_missing(arg))
                else:
                    _frame(vm, arg)
        return self._host

    @data.setter
    def data(self, value):
        self._host, self._base, self._ops = value, None, []

    values = data

    @property
    def shape(self):
        return self._shape if self._host is None else self._host.shape

    def sel(self, indexers=None, **kw):
        return DataArray(self.data, self.dims, self.coords).sel(indexers, **kw)

    def copy(self, deep=True):
        return DataArray(np.array(self.data, copy=True) if deep else self.data, self.dims, dict(self.coords))


def _or_missing(vm, missing):
    # "+= where the bit is not set yet" (criteria.py:317-322) is an OR
    np.bitwise_or(vm, cst.PANDORA_MSK_PIXEL_RIGHT_NODATA_OR_DISPARITY_RANGE_MISSING, out=vm, where=np.asarray(missing, bool))


def _frame(vm, offset):
    vm[:offset, :] = cst.PANDORA_MSK_PIXEL_LEFT_NODATA_OR_BORDER
    vm[-offset:, :] = cst.PANDORA_MSK_PIXEL_LEFT_NODATA_OR_BORDER
    vm[offset:-offset, :offset] = cst.PANDORA_MSK_PIXEL_LEFT_NODATA_OR_BORDER
    vm[offset:-offset, -offset:] = cst.PANDORA_MSK_PIXEL_LEFT_NODATA_OR_BORDER


def _dilate(mask, window_size):
    """scipy.ndimage.binary_dilation(mask, ones((w, w))) (criteria.py:37-62) without scipy."""
    o = window_size // 2
    if o == 0:
        return mask.copy()
    H, W = mask.shape
    pad = np.zeros((H + 2 * o, W + 2 * o), bool)
    pad[o:o + H, o:o + W] = mask
    out = np.zeros((H, W), bool)
    for i in range(window_size):
        for j in range(window_size):
            out |= pad[i:i + H, j:j + W]
    return out


def binary_dilation_msk(img, window_size):
    return _dilate(img["msk"].data == img.attrs["no_data_mask"], window_size)


def validity_mask(img_left, img_right, cv):
    """criteria.py:66-158: allocate cv["validity_mask"] (int64, row x col) from the disparity range,
    the image borders and the input masks."""
    H, W = cv.sizes["row"], cv.sizes["col"]
    line = np.zeros(W, dtype=np.int64)  # the range tests depend on the column only: build one line, then replicate it
    disp = np.asarray(cv.coords["disp"])
    d_min, d_max = disp[0], disp[-1]
    col = np.asarray(cv.coords["col"])
    offset = cv.attrs["offset_row_col"]
    if d_max < 0:  # criteria.py:114-120
        bit_1 = np.where((col + d_max) < (col[0] + offset))
        sel = np.where(((col + d_max) >= (col[0] + offset)) & ((col + d_min) < (col[0] + offset)))
    elif d_min > 0:  # criteria.py:123-129
        bit_1 = np.where((col + d_min) > (col[-1] - offset))
        sel = np.where(((col + d_min) <= (col[-1] - offset)) & ((col + d_max) > (col[-1] - offset)))
    else:  # criteria.py:132-138
        bit_1 = (np.array([], dtype=np.int64),)
        sel = np.where(((col + d_min) < (col[0] + offset)) | (col + d_max > (col[-1]) - offset))
    line[sel[0]] += cst.PANDORA_MSK_PIXEL_RIGHT_INCOMPLETE_DISPARITY_RANGE
    line[bit_1[0]] += cst.PANDORA_MSK_PIXEL_RIGHT_NODATA_OR_DISPARITY_RANGE_MISSING
    coords = {k: cv.coords[k] for k in ("row", "col")}
    if "msk" not in img_left.data_vars and "msk" not in img_right.data_vars:
        cv["validity_mask"] = LazyValidity(line, (H, W), coords)  # every row is this line
        return cv
    vm = np.empty((H, W), np.int64)
    vm[:] = line
    cv["validity_mask"] = DataArray(vm, ("row", "col"))
    if "msk" in img_left.data_vars:
        allocate_left_mask(cv, img_left)
    if "msk" in img_right.data_vars:
        allocate_right_mask(cv, img_right, bit_1)
        if "disparity" in img_left.data_vars:
            mask_partially_missing_variable_ranges(cv, img_left, img_right)
    cv["validity_mask"] = LazyValidity(cv["validity_mask"].data, (H, W), coords)  # (the later updates can still be deferred)
    return cv


def allocate_left_mask(cv, img_left):
    """criteria.py:182-216"""
    vm = cv["validity_mask"].data
    msk = img_left["msk"].data
    dil = binary_dilation_msk(img_left, cv.attrs["window_size"])
    vm += dil.astype(vm.dtype) * vm.dtype.type(cst.PANDORA_MSK_PIXEL_LEFT_NODATA_OR_BORDER)  # whatever integer type the mask has
    vm += np.where((msk != img_left.attrs["no_data_mask"]) & (msk != img_left.attrs["valid_pixels"]),
                   cst.PANDORA_MSK_PIXEL_IN_VALIDITY_MASK_LEFT, 0).astype(vm.dtype)


def allocate_right_mask(cv, img_right, bit_1):
    """criteria.py:219-288: a pixel gets bit 7 (resp. bit 1) when EVERY right position of its
    disparity range is masked invalid (resp. no-data / outside the image)."""
    offset = cv.attrs["offset_row_col"]
    vm = cv["validity_mask"].data
    H, W = vm.shape
    msk = img_right["msk"].data
    disp = np.asarray(cv.coords["disp"])
    d_min, d_max = int(disp[0]), int(disp[-1])  # .astype(int) truncation
    dil = binary_dilation_msk(img_right, cv.attrs["window_size"])
    r_mask = ((msk != img_right.attrs["no_data_mask"]) & (msk != img_right.attrs["valid_pixels"])).astype(np.int64)
    b_2_7 = np.zeros((H, W), np.int64)
    no_data_right = np.zeros((H, W), np.int64)
    col_range = np.arange(W)
    n = len(range(d_min, d_max + 1))
    for dsp in range(d_min, d_max + 1):
        col_d = col_range + dsp
        valid = (col_d >= col_range[0] + offset) & (col_d <= col_range[-1] - offset)
        b_2_7[:, col_range[valid]] += r_mask[:, col_d[valid]]
        b_2_7[:, col_range[~valid]] += 1
        no_data_right[:, col_range[valid]] += dil[:, col_d[valid]]
        no_data_right[:, col_range[~valid]] += 1
        b_2_7[:, bit_1[0]] = 0
        no_data_right[:, bit_1[0]] = 0
        # the reference adds the flag inside the loop (criteria.py:279-288); the counters only reach
        # n at the last iteration, so adding once here is identical
    vm[b_2_7 == n] += vm.dtype.type(cst.PANDORA_MSK_PIXEL_IN_VALIDITY_MASK_RIGHT)
    vm[no_data_right == n] += vm.dtype.type(cst.PANDORA_MSK_PIXEL_RIGHT_NODATA_OR_DISPARITY_RANGE_MISSING)


def partially_missing_variable_ranges(disps, img_mask):
    """cpp/src/criteria.cpp:27-102: True where the pixel's [dmin, dmax] range is not fully inside
    one run of valid right-image columns."""
    H, W = img_mask.shape
    out = np.ones((H, W), bool)
    cols = np.arange(W)
    dmin = disps[0].astype(np.float32).astype(np.int64)  # static_cast<int>(float)
    dmax = disps[1].astype(np.float32).astype(np.int64)
    for r in range(H):
        valid = ~img_mask[r]
        # run id of every valid column, -1 on masked ones
        change = np.diff(np.concatenate([[False], valid]).astype(np.int8)) == 1
        run_id = np.where(valid, np.cumsum(change), -1)
        lo, hi = cols + dmin[r], cols + dmax[r]
        ok = (lo >= 0) & (hi < W) & (lo <= hi)
        lo_c, hi_c = np.clip(lo, 0, W - 1), np.clip(hi, 0, W - 1)
        same = (run_id[lo_c] >= 0) & (run_id[lo_c] == run_id[hi_c])
        out[r] = ~(ok & same)
    return out


def mask_partially_missing_variable_ranges(cv, img_left, img_right):
    """criteria.py:161-179"""
    mask = partially_missing_variable_ranges(np.asarray(img_left["disparity"].data),
                                             img_right["msk"].data != img_right.attrs["valid_pixels"])
    cv["validity_mask"].data[mask] |= cst.PANDORA_MSK_PIXEL_INCOMPLETE_VARIABLE_DISPARITY_RANGE


def mask_invalid_variable_disparity_range(cv, missing_disparity_range=None):
    """criteria.py:291-322.  ``missing_disparity_range``: bool (row, col), True where the cost is NaN
    for every disparity; computed on the GPU when the volume is device resident."""
    arr, lazy = cv.data_vars.get("cost_volume"), cv["validity_mask"]
    if missing_disparity_range is None and isinstance(lazy, LazyValidity) and lazy.pending and hasattr(arr, "device_cv"):
        arr.device_cv.engine.mark_missing(arr.device_cv)  # the snapshot is taken now, it joins the mask when the mask is needed
        lazy.defer("missing", arr.device_cv)
        return
    if missing_disparity_range is None:
        if hasattr(arr, "device_cv"):
            missing_disparity_range = arr.device_cv.engine.nan_pixels(arr.device_cv)
        else:
            missing_disparity_range = np.min(np.isnan(arr.data), axis=2)
    _or_missing(cv["validity_mask"].data, missing_disparity_range)


def mask_border(dataset):
    """criteria.py:325-353"""
    offset = dataset.attrs["offset_row_col"]
    lazy = dataset["validity_mask"]
    if offset > 0:
        if isinstance(lazy, LazyValidity) and lazy.pending:
            lazy.defer("border", int(offset))
        else:
            _frame(lazy.data, offset)
    return dataset["validity_mask"]
