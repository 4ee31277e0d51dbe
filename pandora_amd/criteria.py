"""Validity-mask criteria - host-side (numpy) mirror of the reference's criteria.py.

These are O(H*W) bookkeeping around the hot path (the O(H*W*D) work they depend on - which
pixels are NaN for every disparity - is computed on the GPU, Engine.nan_pixels).
Citations: /root/reference/src/pandora/criteria.py.
"""
import numpy as np

from . import constants as cst
from .dataset import DataArray


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
    vm = np.empty((H, W), np.int64)
    vm[:] = line
    cv["validity_mask"] = DataArray(vm, ("row", "col"))
    if "msk" in img_left.data_vars:
        allocate_left_mask(cv, img_left)
    if "msk" in img_right.data_vars:
        allocate_right_mask(cv, img_right, bit_1)
        if "disparity" in img_left.data_vars:
            mask_partially_missing_variable_ranges(cv, img_left, img_right)
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
    if missing_disparity_range is None:
        arr = cv["cost_volume"]
        if hasattr(arr, "device_cv"):
            missing_disparity_range = arr.device_cv.engine.nan_pixels(arr.device_cv)
        else:
            missing_disparity_range = np.min(np.isnan(arr.data), axis=2)
    vm = cv["validity_mask"].data
    # "+= where the bit is not set yet" (criteria.py:317-322) is an OR
    np.bitwise_or(vm, cst.PANDORA_MSK_PIXEL_RIGHT_NODATA_OR_DISPARITY_RANGE_MISSING, out=vm,
                  where=np.asarray(missing_disparity_range, bool))


def mask_border(dataset):
    """criteria.py:325-353"""
    offset = dataset.attrs["offset_row_col"]
    vm = dataset["validity_mask"].data
    if offset > 0:
        vm[:offset, :] = cst.PANDORA_MSK_PIXEL_LEFT_NODATA_OR_BORDER
        vm[-offset:, :] = cst.PANDORA_MSK_PIXEL_LEFT_NODATA_OR_BORDER
        vm[offset:-offset, :offset] = cst.PANDORA_MSK_PIXEL_LEFT_NODATA_OR_BORDER
        vm[offset:-offset, -offset:] = cst.PANDORA_MSK_PIXEL_LEFT_NODATA_OR_BORDER
    return dataset["validity_mask"]
