"""The reference's own parametrised cases for the criteria sub-functions (tests/test_criteria.py:51-735), replayed through
pandora_amd.criteria with the same call sequence as the reference tests.  Host logic only (no GPU)."""
import json
import os

import numpy as np
import pytest

from pandora_amd import criteria, matching_cost
from pandora_amd.dataset import DataArray, make_image

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "criteria_cases.json")) as f:
    CASES = json.load(f)["tests"]


def _arr(v, dtype=np.float64):
    return np.array([[np.nan if x is None else x for x in row] for row in v], dtype)


def _left(case, disparity=(-1, 1), grids=None):
    return make_image(_arr(case["left_data"], np.float32), disparity=None if grids else list(disparity), msk=np.array(case["left_msk"]),
                      valid_pixels=case["left_attrs"]["valid_pixels"], no_data_mask=case["left_attrs"]["no_data_mask"],
                      disparity_grids=grids)


def _grid(left, window_size):
    m = matching_cost.AbstractMatchingCost(matching_cost_method="sad", window_size=window_size, subpix=1)
    cv = m.allocate_cost_volume(left, (left["disparity"].sel(band_disp="min"), left["disparity"].sel(band_disp="max")))
    cv["validity_mask"] = DataArray(np.zeros((cv.sizes["row"], cv.sizes["col"]), np.uint16), ("row", "col"))
    return cv


@pytest.mark.parametrize("case", CASES["test_binary_dilation_msk"], ids=lambda c: c["id"])
def test_binary_dilation_msk(case):  # test_criteria.py:51-105
    np.testing.assert_array_equal(criteria.binary_dilation_msk(_left(case), case["window_size"]), np.array(case["gt_dil"], bool))


@pytest.mark.parametrize("case", CASES["test_mask_border"], ids=lambda c: c["id"])
def test_mask_border(case):  # test_criteria.py:107-190
    cv = _grid(_left(case), case["window_size"])
    criteria.mask_border(cv)
    np.testing.assert_array_equal(cv["validity_mask"].data, np.array(case["gt_mask"]))


@pytest.mark.parametrize("case", CASES["test_allocate_left_mask"], ids=lambda c: c["id"])
def test_allocate_left_mask(case):  # test_criteria.py:627-735
    left = _left(case)
    cv = _grid(left, case["window_size"])
    criteria.allocate_left_mask(cv, left)
    np.testing.assert_array_equal(cv["validity_mask"].data, np.array(case["gt_mask"]))


@pytest.mark.parametrize("case", CASES["test_allocate_right_mask"], ids=lambda c: c["id"])
def test_allocate_right_mask(case):  # test_criteria.py:345-625
    left = _left(case, case["disparity"])
    right = make_image(_arr(case["right_data"], np.float32), msk=np.array(case["right_msk"]),
                       valid_pixels=case["right_attrs"]["valid_pixels"], no_data_mask=case["right_attrs"]["no_data_mask"])
    cv = _grid(left, case["window_size"])
    criteria.allocate_right_mask(cv, right, tuple(np.array(b, np.int64) for b in case["bit_1"]))
    np.testing.assert_array_equal(cv["validity_mask"].data, np.array(case["gt_mask"]))


@pytest.mark.parametrize("case", CASES["test_mask_invalid_variable_disparity_range"], ids=lambda c: c["id"])
def test_mask_invalid_variable_disparity_range(oracle, case):  # test_criteria.py:192-330
    """validity_mask on per-pixel disparity grids (incl. criteria_cpp.partially_missing_variable_ranges), then a pixel whose
    costs are all NaN.  The cost volume's NaN pattern (GPU in the product) comes from the oracle here."""
    gmin, gmax = _arr(case["disp_min_grid"]), _arr(case["disp_max_grid"])
    left = _left(case, grids=(gmin, gmax))
    left.attrs["disparity_source"] = [int(np.nanmin(gmin)), int(np.nanmax(gmax))]
    right = make_image(_arr(case["right_data"], np.float32), msk=np.array(case["right_msk"]),
                       valid_pixels=case["right_attrs"]["valid_pixels"], no_data_mask=case["right_attrs"]["no_data_mask"])
    m = matching_cost.AbstractMatchingCost(matching_cost_method="sad", window_size=1, subpix=1)
    cv = m.allocate_cost_volume(left, (left["disparity"].sel(band_disp="min"), left["disparity"].sel(band_disp="max")))
    cv = criteria.validity_mask(left, right, cv)
    dmin, dmax = int(np.nanmin(gmin)), int(np.nanmax(gmax))
    vol = oracle.sad_ssd(_arr(case["left_data"], np.float32), _arr(case["right_data"], np.float32), dmax - dmin + 1, dmin, 1, 1, False)
    vol[1, 0, :] = np.nan
    criteria.mask_invalid_variable_disparity_range(cv, np.min(np.isnan(vol), axis=2))
    np.testing.assert_array_equal(cv["validity_mask"].data, np.array(case["gt_mask"]))
