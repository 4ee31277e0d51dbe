"""Pins the CPU oracle (oracle/oracle.c) against known answers transcribed from the reference's
own tests (tests/golden/known_answers.py).  CPU only."""
import numpy as np
import pytest

from tests.golden import known_answers as ka


def _cv(oracle, method, left, right, win, subpix, dmin, dmax, masked=True, **mask_kw):
    L = np.asarray(left, np.float32)
    R = np.asarray(right, np.float32)
    D = (dmax - dmin) * subpix + 1
    if method == "census":
        cv = oracle.census_cost(L, R, D, dmin, subpix, win)
    elif method in ("sad", "ssd"):
        cv = oracle.sad_ssd(L, R, D, dmin, subpix, win, method == "ssd")
    else:
        cv = oracle.zncc(L, R, D, dmin, subpix, win)
    if masked:
        H, W = L.shape
        oracle.cv_masked(cv, dmin, subpix, win, dmin=np.full((H, W), dmin), dmax=np.full((H, W), dmax), **mask_kw)
    return cv


@pytest.mark.parametrize("case", ka.CENSUS, ids=lambda c: c["cite"])
def test_census_known_answers(oracle, case):
    cv = _cv(oracle, "census", case["left"], case["right"], case["win"], case["subpix"], case["dmin"], case["dmax"])
    exp = np.moveaxis(np.array(case["expected_dhw"], np.float32), 0, -1)
    np.testing.assert_array_equal(cv, exp)


def _bits(code_bytes, nbits):
    v = 0
    for b in code_bytes:
        v = (v << 8) | int(b)
    total = 8 * len(code_bytes)
    return v >> (total - nbits)


def test_census_bit_strings(oracle):
    # the python census_transform of the reference returns the w*w-bit integer, MSB = first window pixel;
    # the C++ one packs the same bit order into bytes MSB-first (census.cpp:27-28,77)
    img = np.asarray(ka.CENSUS_BITS_W3["image"], np.float32)
    codes = oracle.census_transform(img, 3)
    got = [[_bits(codes[r, c], 9) for c in range(1, 5)] for r in range(1, 4)]
    assert got == ka.CENSUS_BITS_W3["expected"]
    codes = oracle.census_transform(img, 5)
    got = [_bits(codes[2, c], 25) for c in (2, 3)]
    assert got == ka.CENSUS_BITS_W5["expected"]


@pytest.mark.parametrize("case", ka.SAD_SSD, ids=lambda c: c["cite"])
def test_sad_ssd_slices(oracle, case):
    cv = _cv(oracle, "ssd" if case["squared"] else "sad", case["left"], case["right"], case["win"], case["subpix"],
             case["dmin"], case["dmax"], masked=case["masked"])
    np.testing.assert_array_equal(cv[:, :, case["disp_index"]], np.array(case["expected"], np.float32))


@pytest.mark.parametrize("case", [ka.SAD_FULL, ka.SAD_SUBPIX], ids=lambda c: c["cite"])
def test_sad_full_volumes(oracle, case):
    cv = _cv(oracle, "sad", case["left"], case["right"], case["win"], case["subpix"], case["dmin"], case["dmax"])
    np.testing.assert_array_equal(cv, np.array(case["expected"], np.float32))


@pytest.mark.parametrize("win", [3, 5])
@pytest.mark.parametrize("squared", [0, 1])
def test_sad_ssd_float_summation_order(oracle, win, squared):
    """Non-integer float32 images: the w*w terms must be added in numpy's order (window columns outer, rows inner,
    sad_ssd.py:340-368).  Fixture produced by tests/golden/gen_sad_float_golden.py from the reference's expressions."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "sad_float_order.npz"))
    cv = oracle.sad_ssd(g["left"], g["right"], 7, -3, 1, win, bool(squared))
    np.testing.assert_array_equal(cv, g[f"w{win}_sq{squared}_d-3_n7"])


def test_zncc_known_answer(oracle):
    c = ka.ZNCC
    L = np.asarray(c["left"], np.float64)
    R = np.asarray(c["right"], np.float64)
    cv = _cv(oracle, "zncc", L, R, c["win"], c["subpix"], c["dmin"], c["dmax"])
    for k, (l0, l1), (r0, r1), col in c["checks"]:
        row, colr = L[:, l0:l1], R[:, r0:r1]
        gt = (np.mean(row * colr) - np.mean(row) * np.mean(colr)) / (np.std(row) * np.std(colr))
        exp = np.full(6, np.nan)
        exp[col] = gt
        np.testing.assert_allclose(cv[2, :, k], exp, rtol=1e-5)
    # everything outside row 2 is NaN with a 5x5 window on a 5-row image
    assert np.isnan(cv[[0, 1, 3, 4]]).all()


@pytest.mark.parametrize("method", ["census", "sad", "ssd", "zncc"])
@pytest.mark.parametrize("case", ka.CV_MASKED, ids=lambda c: c["cite"])
def test_cv_masked_nan_pattern(oracle, case, method):
    cv = _cv(oracle, method, case["left"], case["right"], case["win"], case["subpix"], case["dmin"], case["dmax"],
             mskL=np.array(case["left_mask"], np.int16), mskR=np.array(case["right_mask"], np.int16),
             valid=case["valid"], nodata=case["nodata"])
    np.testing.assert_array_equal(np.isnan(cv), ka.nanmask(case["nan"]))


def test_wta_known_answers(oracle):
    c = ka.WTA
    for (dmin, dmax), gt in c["cases"]:
        cv = _cv(oracle, "sad", c["left"], c["right"], 1, 1, dmin, dmax)
        disp, _ = oracle.wta(cv, dmin, 1, False, 0.0)
        np.testing.assert_array_equal(disp, np.array(gt, np.float32))


def test_cbca_known_answers(oracle):
    c = ka.CBCA
    L = np.asarray(c["left"], np.float32)
    R = np.asarray(c["right"], np.float32)
    arms = oracle.cross_support(L, c["distance"], c["intensity"])
    np.testing.assert_array_equal(arms[:, :, 0], c["arms_left"])
    np.testing.assert_array_equal(arms[:, :, 1], c["arms_right"])
    np.testing.assert_array_equal(arms[:, :, 2], c["arms_top"])
    np.testing.assert_array_equal(arms[:, :, 3], c["arms_bottom"])
    # cost volume of the test's setUp: |L - R_d|, NaN outside the overlap (window 1 SAD)
    cv = oracle.sad_ssd(L, R, 3, -1, 1, 1, False)
    # cbca.py:184-295: arms on the 3x3-median-filtered images
    cl = oracle.cross_support(np.nan_to_num(oracle.median3(L), nan=np.inf), c["distance"], c["intensity"])
    cr = oracle.cross_support(np.nan_to_num(oracle.median3(R), nan=np.inf), c["distance"], c["intensity"])
    oracle.cbca(cv, -1, 1, 0, cl, [cr])
    np.testing.assert_allclose(cv, np.array(c["aggregated"], np.float32), rtol=1e-7)


@pytest.mark.parametrize("case", ka.MEDIAN, ids=lambda c: c["cite"])
def test_median_known_answers(oracle, case):
    disp = np.array(case["disp"], np.float32)
    valid = np.array(case["valid"])
    size = case.get("size", 3)
    masked = disp.copy()
    masked[(valid & 0b01111000011) != 0] = np.nan
    med = oracle.median3(masked) if size == 3 else oracle.median_filter(masked, size)
    np.testing.assert_array_equal(med, oracle.median_filter(masked, size))  # the CBCA helper is the size-3 case
    out = disp.copy()
    ok = np.isfinite(masked)
    out[ok] = med[ok]
    np.testing.assert_array_equal(out, np.array(case["expected"], np.float32))
    # MedianFilter.filter_disparity (median.py:94-131) as one call
    np.testing.assert_array_equal(oracle.filter_median_disparity(disp, valid, size), np.array(case["expected"], np.float32))


@pytest.mark.parametrize("case", ka.CROSS_CHECKING, ids=lambda c: c["cite"])
def test_cross_checking_known_answers(oracle, case):
    val, conf = oracle.cross_checking(np.array(case["left"], np.float32), np.array(case["validity"], np.int64),
                                      np.array(case["right"], np.float32), case["interval"][0], case["interval"][1],
                                      case["threshold"])
    if case["conf"] is not None:
        np.testing.assert_array_equal(conf, np.array(case["conf"], np.float32))
    if case["mask"] is not None:
        np.testing.assert_array_equal(val, np.array(case["mask"], np.int64))


def _bilateral_numpy(disp, valid, sigma_color, sigma_space):
    """The recipe of the reference's own tests (tests/test_filter.py:373-470, :472-616, :618-680): explicit float64
    gaussian weights around every pixel whose window fits, NaN (invalid) window elements dropped."""
    masked = disp.astype(np.float32).copy()
    masked[(valid & 0b01111000011) != 0] = np.nan
    H, W = masked.shape
    win = min(H, W, int(3 * sigma_space + 1))
    off = win // 2
    ii, jj = np.meshgrid(np.arange(win), np.arange(win), indexing="ij")
    gs = np.exp(-((np.sqrt((ii - win // 2) ** 2 + (jj - win // 2) ** 2) / sigma_space) ** 2) * 0.5) / (sigma_space * np.sqrt(2 * np.pi))
    out = disp.astype(np.float32).copy()
    for r in range(off, H - (win - off) + 1):
        for c in range(off, W - (win - off) + 1):
            if not np.isfinite(masked[r, c]):
                continue
            w = masked[r - off:r - off + win, c - off:c - off + win].astype(np.float64)
            gi = np.exp(-(((w - w[off, off]) / sigma_color) ** 2) * 0.5) / (sigma_color * np.sqrt(2 * np.pi))
            wt = gs * gi
            out[r, c] = np.nansum(w * wt) / np.nansum(wt)
    return out


@pytest.mark.parametrize("case", ["test_on_valid_pixels", "test_with_nans", "test_with_invalid_center", "random"])
def test_bilateral_filter_formula(oracle, case):
    disp = np.array([[5, 6, 7, 8, 9], [6, 85, 1, 36, 5], [5, 9, 23, 12, 2], [6, 1, 9, 2, 4], [6, 7, 4, 2, 1]], np.float32)
    valid = np.zeros((5, 5), np.int64)
    sc, ss = 4.0, 6.0
    if case == "test_with_nans":  # test_filter.py:472-616: invalid neighbours are ignored
        valid[1, 1], valid[2, 1], valid[2, 4], valid[3, 4] = 4, 16, 8, 8  # information bits only ...
        valid[0, 2], valid[1, 3] = 1, 2                                    # ... and really invalid ones
    elif case == "test_with_invalid_center":  # test_filter.py:618-680: an invalid centre keeps its value
        valid[2, 2] = 0b01111000011
    elif case == "random":
        rng = np.random.default_rng(8)
        disp = (rng.integers(-30, 5, (16, 21)) + rng.random((16, 21))).astype(np.float32)
        valid = np.where(rng.random((16, 21)) < 0.15, 1, 0).astype(np.int64)
        sc, ss = 2.0, 1.5  # 5x5 window -> many interior pixels
    got = oracle.filter_bilateral_disparity(disp, valid, sc, ss)
    np.testing.assert_allclose(got, _bilateral_numpy(disp, valid, sc, ss), rtol=1e-6)
    if case == "test_with_invalid_center":
        assert got[2, 2] == disp[2, 2]


def test_disparity_range_known_answer(oracle):
    c = ka.DISPARITY_RANGE
    lo, hi = oracle.disparity_range(np.array(c["disp"], np.float32), np.array(c["validity"], np.int64), c["window_size"],
                                    c["marge"], c["dmin"], c["dmax"])
    np.testing.assert_array_equal(lo, np.array(c["range_min"], np.float32))
    np.testing.assert_array_equal(hi, np.array(c["range_max"], np.float32))


@pytest.mark.parametrize("case", ka.INTERPOLATION, ids=lambda c: c["cite"])
def test_interpolation_reference_vectors(oracle, case):
    """AbstractInterpolation.interpolated_disparity (interpolated_disparity.py:200-233 mc-cnn: occlusions then mismatches;
    :318-330 sgm: mismatches then occlusions) on the reference's own five cases."""
    d, v = np.array(case["disp"], np.float32), np.array(case["validity"], np.int32)
    for which in (("occlusion_mc_cnn", "mismatch_mc_cnn") if case["method"] == "mc-cnn" else ("mismatch_sgm", "occlusion_sgm")):
        d, v = oracle.interpolate_disparity(which, d, v)
    np.testing.assert_array_equal(v, np.array(case["out_validity"], np.int32))
    np.testing.assert_array_equal(d, np.array(case["out_disp"], np.float32))


@pytest.mark.parametrize("case", ka.RISK, ids=lambda c: c["cite"])
def test_risk_reference_vectors(oracle, case):
    """risk.cpp:28-197; risk_min is tied to the other maps through the ambiguity integral (risk.cpp:166-167:
    risk_min = mean(1 + span - sampled ambiguity) = 1 + risk_max - integral / nbr_etas)."""
    cv, etas = np.array(case["cv"], np.float32), np.array(case["etas"])
    gmin, gmax, dr = np.array(case["grid_min"], np.int64), np.array(case["grid_max"], np.int64), np.array(case["disp_range"], np.float32)
    risk_max, risk_min, sup, inf = oracle.risk(cv, etas, gmin, gmax, dr)
    np.testing.assert_allclose(risk_max, np.array(case["risk_max"], np.float32), rtol=1e-6)
    np.testing.assert_allclose(sup, np.array(case["disp_sup"], np.float32), rtol=1e-6)
    np.testing.assert_allclose(inf, np.array(case["disp_inf"], np.float32), rtol=1e-6)
    np.testing.assert_allclose(risk_max, sup - inf, rtol=1e-6)  # test_risk.py:159-160
    amb = oracle.ambiguity(cv, etas, gmin, gmax, dr)
    ok = ~np.isnan(risk_max)
    np.testing.assert_allclose(risk_min[ok], (1 + risk_max - amb / len(etas))[ok], rtol=1e-6)


def test_interval_bounds_reference_vector(oracle):
    c = ka.INTERVAL_BOUNDS
    cv, dr = np.array(c["cv"], np.float32), np.array(c["disp_range"], np.float32)
    g = np.ones(cv.shape[:2], np.int64)
    lo, hi = oracle.interval_bounds(cv, c["threshold"], c["type_factor"], -g, g, dr)
    np.testing.assert_allclose(lo, np.array(c["inf"], np.float32), rtol=1e-6)
    np.testing.assert_allclose(hi, np.array(c["sup"], np.float32), rtol=1e-6)
    # and the volume itself is SAD window 1 of the confidence pair with the left mask applied
    sad = oracle.sad_ssd(np.array(ka.CONFIDENCE_LEFT, np.float32), np.array(ka.CONFIDENCE_RIGHT, np.float32), 3, -1, 1, 1, False)
    m = np.array(ka.CONFIDENCE_LEFT_MASK, bool)
    np.testing.assert_array_equal(np.nan_to_num(sad[~m], nan=-1), np.nan_to_num(cv[~m], nan=-1))


def test_oracle_results_do_not_depend_on_the_thread_count(oracle):
    """The OpenMP loops of the oracle (census, SGM, WTA, refinement) give identical bits with 1 thread (the reference's
    serial order) and with every core (bench.py's cpu_baseline_all_cores)."""
    rng = np.random.default_rng(21)
    L = rng.integers(0, 255, (37, 61)).astype(np.float32)
    R = np.roll(L, 3, 1) + rng.integers(-2, 3, L.shape).astype(np.float32)
    outs = []
    for threads in (1, 0, 3):
        oracle.set_threads(threads)
        cv = oracle.census_cost(L, R, 13, -6, 1, 5)
        s = oracle.sgm(cv, 8.0, 32.0, False, 26.0, False)
        disp, val = oracle.wta(s, -6, 1, False, -9999.0)
        outs.append((cv, s, disp, val) + tuple(oracle.refine(s, disp, val, -6, 6, 1, False, "vfit")))
    oracle.set_threads(1)
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("case", ka.CROSS_SUPPORTS, ids=lambda c: c["cite"])
def test_cross_supports_reference_vectors(oracle, case):
    from tests.cbca_helpers import oracle_cross_supports

    L, R = np.asarray(case["left"], np.float32), np.asarray(case["right"], np.float32)
    cl, crs = oracle_cross_supports(oracle, L, R, case["msk_left"], case["msk_right"], case["subpix"], case["win"] // 2,
                                    case["distance"], case["intensity"])
    assert len(crs) == case["subpix"]
    if case["arms_left"] is not None:
        np.testing.assert_array_equal(cl, np.array(case["arms_left"]))
    np.testing.assert_array_equal(crs[case["right_index"]], np.array(case["arms_right"]))


@pytest.mark.parametrize("case", ka.CBCA_PIPELINES, ids=lambda c: c["cite"])
def test_cbca_pipeline_reference_vectors(oracle, case):
    from tests.cbca_helpers import oracle_sad_cbca

    cv = oracle_sad_cbca(oracle, case["left"], case["right"], case["msk_left"], case["msk_right"], case["win"], case["subpix"], -1, 1, 3, 5.0)
    exp = np.array(case["expected"], np.float32)
    got = cv if case["disp_index"] is None else cv[:, :, case["disp_index"]]
    assert exp.shape == got.shape
    np.testing.assert_allclose(got, exp, rtol=1e-7)


@pytest.mark.parametrize("case", ka.REFINEMENT, ids=lambda c: c["cite"])
def test_refinement_reference_vectors(oracle, case):
    """tests/test_refinement.py: the expected maps are float64 formulas (the reference compares the quadratic ones with a
    meaningless tolerance of 1e10 - 7 and the vfit ones exactly); here both within float32 rounding."""
    cv = np.array(case["cv"], np.float32)
    disp, val = np.array(case["disp"], np.float32), np.zeros(np.shape(case["disp"]), np.int64)
    itp, out_disp, out_val = oracle.refine(cv, disp, val, case["d_min"], case["d_max"], case["subpix"], False, case["method"])
    np.testing.assert_array_equal(out_val, np.array(case["mask"]))
    np.testing.assert_allclose(out_disp, np.array(case["out_disp"], np.float32), rtol=2e-7, atol=0)
    np.testing.assert_allclose(itp, np.array(case["itp"], np.float32), rtol=2e-7, atol=0)


import json as _json
import os as _os

with open(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden", "census_cases.json")) as _f:
    CENSUS_CASES = _json.load(_f)["cases"]


def census_case_arrays(case):
    """-> (left, right, dmin, dmax, expected [H][W] or [H][W][D], layer index or None)"""
    exp = np.array([[[np.nan if v is None else v for v in col] if isinstance(col, list) else (np.nan if col is None else col)
                     for col in row] for row in case["expected"]], np.float32)
    dmin, dmax = case["disp_interval"]
    layer = None if case["tested_layer"] == "all" else int(round((case["tested_layer"] - dmin) * case["subpix"]))
    return np.array(case["left"], np.float32), np.array(case["right"], np.float32), dmin, dmax, exp, layer


@pytest.mark.parametrize("case", CENSUS_CASES, ids=lambda c: c["id"])
def test_census_parametrised_reference_cases(oracle, case):
    """tests/test_matching_cost/test_matching_cost_census.py:379-729 (test_census): every window size 3..13, the zero-cost case
    and the full sub-pixel volume; census -> cv_masked as in the reference test."""
    L, R, dmin, dmax, exp, layer = census_case_arrays(case)
    D = (dmax - dmin) * case["subpix"] + 1
    cv = oracle.census_cost(L, R, D, dmin, case["subpix"], case["window_size"])
    oracle.cv_masked(cv, dmin, case["subpix"], case["window_size"])
    np.testing.assert_array_equal(cv if layer is None else cv[:, :, layer], exp)


with open(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden", "cv_masked_cases.json")) as _f:
    CV_MASKED_CASES = [dict(c, method=m) for c in _json.load(_f)["cases"] for m in c["methods"]]


def cv_masked_case_arrays(case):
    """-> (left, right, dmin, dmax, masks tuple or None, grids tuple or None, expected NaN mask bool [H][W][D])"""
    L, R = np.array(case["left"], np.float32), np.array(case["right"], np.float32)
    masks = None
    if case["left_mask"] is not None or case["right_mask"] is not None:
        masks = (None if case["left_mask"] is None else np.array(case["left_mask"], np.int16),
                 None if case["right_mask"] is None else np.array(case["right_mask"], np.int16), case["valid_pixels"], case["no_data_mask"])
    grids = None
    if "disparity_grids" in case:
        g = np.array(case["disparity_grids"])
        grids = (g[0].astype(np.float64), g[1].astype(np.float64))
        dmin, dmax = int(g[0].min()), int(g[1].max())  # matching_cost.py:604-616 get_min_max_from_grid
    else:
        dmin, dmax = case["disparity"]
    return L, R, dmin, dmax, masks, grids, np.array(case["expected_nan_mask"], bool)


@pytest.mark.parametrize("case", CV_MASKED_CASES, ids=lambda c: f"{c['id']}-{c['method']}")
def test_cv_masked_parametrised_reference_cases(oracle, case):
    """tests/test_matching_cost/test_matching_cost.py:699-1786: every cv_masked case of the reference (window 1/3/5, subpix 1/2/4,
    both mask conventions, per-pixel disparity grids) for every matching cost the reference runs it with."""
    L, R, dmin, dmax, masks, grids, exp = cv_masked_case_arrays(case)
    sp, win = case["subpix"], case["window_size"]
    D = (dmax - dmin) * sp + 1
    if case["method"] == "census":
        cv = oracle.census_cost(L, R, D, dmin, sp, win)
    elif case["method"] in ("sad", "ssd"):
        cv = oracle.sad_ssd(L, R, D, dmin, sp, win, case["method"] == "ssd")
    else:
        cv = oracle.zncc(L, R, D, dmin, sp, win)
    kw = {}
    if masks:
        kw.update(mskL=masks[0], mskR=masks[1], valid=masks[2], nodata=masks[3])
    if grids:
        kw.update(dmin=grids[0], dmax=grids[1])
    oracle.cv_masked(cv, dmin, sp, win, **kw)
    np.testing.assert_array_equal(np.isnan(cv), exp)


@pytest.mark.parametrize("case", ka.WTA_MORE, ids=lambda c: c["cite"])
def test_wta_more_reference_vectors(oracle, case):
    L, R = np.array(ka.WTA["left"], np.float32), np.array(ka.WTA["right"], np.float32)
    sp, win, dmin, dmax = case["subpix"], case["win"], case["dmin"], case["dmax"]
    D = (dmax - dmin) * sp + 1
    cv = oracle.zncc(L, R, D, dmin, sp, win) if case["method"] == "zncc" else oracle.sad_ssd(L, R, D, dmin, sp, win, False)
    if case["masked"]:
        oracle.cv_masked(cv, dmin, sp, win)
    disp, _ = oracle.wta(cv, dmin, sp, case["is_max"], float(case["invalid"]))
    np.testing.assert_array_equal(disp, np.array(case["disp"], np.float32))


def _denoiser_vectors():
    import json
    import os

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "disparity_denoiser.json")) as f:
        return json.load(f)


def test_disparity_denoiser_reference_vectors(oracle):
    """The restatement of DisparityDenoiser.filter_disparity against the literal arrays of the reference's own tests
    (tests/golden/make_denoiser_vectors.py): the end-to-end 2x2 case of test_with_valid_pixel_multiband_and_monoband (its expected
    map follows from the test's hand-written distances), get_grad's 3x3 case, and the invalid centre that must keep its value."""
    from scipy.ndimage import gaussian_filter

    v = _denoiser_vectors()
    g = v["get_grad"]
    grad = np.gradient(gaussian_filter(np.array(g["disp"], float), sigma=g["sigma_grad"]))
    np.testing.assert_allclose(grad[0], np.array(g["grad_row"]), atol=1e-7)
    np.testing.assert_allclose(grad[1], np.array(g["grad_col"]), atol=1e-7)
    e = v["end_to_end"]
    disp, band = np.array(e["disp"], np.float32), np.array(e["band"], np.float32)
    grad = np.gradient(gaussian_filter(disp, sigma=1.5))
    got = oracle.denoise_disparity(disp, np.zeros(disp.shape, np.int64), band, grad[0], grad[1], e["cfg"]["filter_size"],
                                   e["cfg"]["sigma_euclidian"], e["cfg"]["sigma_color"], e["cfg"]["sigma_planar"])
    np.testing.assert_allclose(got, np.array(e["expected"]), rtol=2e-7)
    c = v["invalid_center"]
    disp, band = np.array(c["disp"], np.float32), np.array(c["band_green"], np.float32)
    val = np.zeros(disp.shape, np.int64)
    val[tuple(c["invalid_at"])] = 0b01111000011
    grad = np.gradient(gaussian_filter(disp, sigma=1.5))
    got = oracle.denoise_disparity(disp, val, band, grad[0], grad[1], 11, 4.0, 100.0, 12.0)
    assert got[2, 2] == disp[2, 2] and not np.array_equal(got, disp)


def test_disparity_denoiser_against_a_numpy_statement(oracle):
    """... and against the formulas of disparity_denoiser.py:168-232 written with numpy on padded windows (random 9x13 map, a NaN
    disparity - every window that sees it turns NaN, as in the reference -, an invalid pixel, window 5)."""
    from scipy.ndimage import gaussian_filter

    rng = np.random.default_rng(12)
    H, W, ws = 9, 13, 5
    o = ws // 2
    disp = (rng.integers(-20, 5, (H, W)) + rng.random((H, W))).astype(np.float32)
    disp[6, 10] = np.nan
    band = rng.integers(0, 255, (H, W)).astype(np.float32)
    val = np.zeros((H, W), np.int64)
    val[2, 3] = 2
    grad = np.gradient(gaussian_filter(disp, sigma=1.5))
    got = oracle.denoise_disparity(disp, val, band, grad[0], grad[1], ws, 4.0, 100.0, 12.0)
    dp, bp = np.pad(disp, o, "reflect"), np.pad(band, o, "reflect")
    ii, jj = np.meshgrid(np.arange(-o, o + 1), np.arange(-o, o + 1), indexing="ij")
    exp = disp.copy()
    for r in range(H):
        for c in range(W):
            if val[r, c] & 0x3C3 or not np.isfinite(disp[r, c]):
                continue
            win, clr = dp[r:r + ws, c:c + ws], bp[r:r + ws, c:c + ws]
            dist = win - (ii * grad[0][r, c] + jj * grad[1][r, c])
            w = (np.exp(-(np.sqrt(ii ** 2 + jj ** 2) / 4.0) ** 2 / 2) * np.exp(-np.power((clr - clr[o, o]) / 100.0, 2.0) / 2.0)
                 * np.exp(-((dist - dist.mean()) / 12.0) ** 2 / 2))
            exp[r, c] = disp[r, c] + np.sum((dist - win[o, o]) * (w / w.sum()))
    np.testing.assert_allclose(got, exp, rtol=1e-6, equal_nan=True)
    assert np.isnan(got[5, 9]) and got[2, 3] == disp[2, 3]
