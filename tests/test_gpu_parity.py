"""GPU parity: every HIP kernel, called through the C ABI (ctypes -> libpandora_amd.so), against the
CPU oracle on the same seeded inputs.  Bit-exact for census / NaN patterns / WTA indices / integer
costs; float costs within the tolerance written in each test."""
import os

import numpy as np
import pytest

from tests.golden import known_answers as ka

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["lazy", "eager"])
def eng(request):
    """Every test runs twice: with the lazy exact representations (integer fast path: Census -> SGM ->
    WTA without a float volume) and with them off (always float32, the general kernels)."""
    from pandora_amd.engine import Engine

    e = Engine(0)
    e.set_lazy(request.param == "lazy")
    yield e
    e.close()


def pair(H, W, seed, integer=True, shift=3):
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (H, W + 2 * abs(shift) + 4)).astype(np.float32)
    # low-pass so that census has ties
    base = np.floor((base + np.roll(base, 1, 1) + np.roll(base, 1, 0)) / 3.0)
    L = base[:, abs(shift):abs(shift) + W].copy()
    R = base[:, abs(shift) + shift:abs(shift) + shift + W].copy() + rng.integers(-2, 3, (H, W))
    if not integer:
        L += rng.random((H, W)).astype(np.float32)
        R += rng.random((H, W)).astype(np.float32)
    return L.astype(np.float32), R.astype(np.float32)


def gpu_cv(eng, method, L, R, dmin, dmax, subpix, win, masks=None, grids=None, masked=True):
    eng.set_images(L, R, subpix)
    if masks:
        eng.set_masks(masks[0], masks[1], masks[2], masks[3])
    if grids:
        eng.set_disparity_grids(grids[0], grids[1])
    D = (dmax - dmin) * subpix + 1
    cv = eng.alloc_cv(D, dmin)
    if method == "census":
        eng.census(cv, win)
    elif method in ("sad", "ssd"):
        eng.sad_ssd(cv, win, method == "ssd")
    else:
        eng.zncc(cv, win)
    if masked:
        eng.cv_masked(cv, win)
    return cv


def cpu_cv(oracle, method, L, R, dmin, dmax, subpix, win, masks=None, grids=None):
    D = (dmax - dmin) * subpix + 1
    if method == "census":
        cv = oracle.census_cost(L, R, D, dmin, subpix, win)
    elif method in ("sad", "ssd"):
        cv = oracle.sad_ssd(L, R, D, dmin, subpix, win, method == "ssd")
    else:
        cv = oracle.zncc(L, R, D, dmin, subpix, win)
    kw = {}
    if masks:
        kw.update(mskL=masks[0], mskR=masks[1], valid=masks[2], nodata=masks[3])
    if grids:
        kw.update(dmin=grids[0], dmax=grids[1])
    if kw:
        oracle.cv_masked(cv, dmin, subpix, win, **kw)
    return cv


@pytest.mark.parametrize("H,W,dmin,dmax,sp,win", [
    (37, 53, -7, 4, 1, 5), (16, 40, -3, 3, 2, 3), (21, 35, -2, 2, 4, 7), (40, 64, 0, 16, 1, 13),
    (33, 129, -60, 0, 1, 5), (24, 50, 3, 9, 1, 9), (20, 47, -4, 1, 2, 11), (8, 9, -1, 1, 1, 3)])
def test_census_bit_exact(eng, oracle, H, W, dmin, dmax, sp, win):
    L, R = pair(H, W, seed=H * W + win)
    got = gpu_cv(eng, "census", L, R, dmin, dmax, sp, win).to_host()
    np.testing.assert_array_equal(got, cpu_cv(oracle, "census", L, R, dmin, dmax, sp, win))


@pytest.mark.parametrize("case", ka.CENSUS + [ka.SAD_FULL, ka.SAD_SUBPIX], ids=lambda c: c["cite"])
def test_reference_known_answers_on_gpu(eng, case):
    method = "census" if "expected_dhw" in case else "sad"
    L, R = np.array(case["left"], np.float32), np.array(case["right"], np.float32)
    got = gpu_cv(eng, method, L, R, case["dmin"], case["dmax"], case["subpix"], case["win"]).to_host()
    exp = np.moveaxis(np.array(case["expected_dhw"], np.float32), 0, -1) if method == "census" else np.array(case["expected"], np.float32)
    np.testing.assert_array_equal(got, exp)


@pytest.mark.parametrize("method", ["census", "sad", "ssd", "zncc"])
@pytest.mark.parametrize("case", ka.CV_MASKED, ids=lambda c: c["cite"])
def test_reference_nan_patterns_on_gpu(eng, case, method):
    L, R = np.array(case["left"], np.float32), np.array(case["right"], np.float32)
    masks = (np.array(case["left_mask"], np.int16), np.array(case["right_mask"], np.int16), case["valid"], case["nodata"])
    got = gpu_cv(eng, method, L, R, case["dmin"], case["dmax"], case["subpix"], case["win"], masks=masks).to_host()
    np.testing.assert_array_equal(np.isnan(got), ka.nanmask(case["nan"]))


@pytest.mark.parametrize("method,integer", [("sad", True), ("ssd", True), ("sad", False), ("ssd", False)])
@pytest.mark.parametrize("H,W,dmin,dmax,sp,win", [(30, 44, -6, 3, 1, 5), (18, 33, -2, 2, 2, 3), (19, 31, -1, 2, 4, 1), (26, 40, 0, 5, 1, 9)])
def test_sad_ssd(eng, oracle, method, integer, H, W, dmin, dmax, sp, win):
    L, R = pair(H, W, seed=3 * H + W, integer=integer)
    got = gpu_cv(eng, method, L, R, dmin, dmax, sp, win).to_host()
    exp = cpu_cv(oracle, method, L, R, dmin, dmax, sp, win)
    np.testing.assert_array_equal(np.isnan(got), np.isnan(exp))
    # same float32 summation order as the oracle: bit-exact (reference tolerance would be 1e-5)
    np.testing.assert_array_equal(got, exp)


@pytest.mark.parametrize("method", ["sad", "ssd"])
@pytest.mark.parametrize("H,W,dmin,dmax,win", [(23, 210, -70, 75, 5), (17, 140, 3, 70, 7), (20, 90, -129, -1, 3), (9, 75, -2, 66, 1)])
def test_sad_ssd_register_window_kernel_blocks(eng, oracle, method, H, W, dmin, dmax, win):
    """subpix 1, window <= 7 takes the register-window kernel: several 64-disparity blocks with a ragged last one,
    ranges that leave the image, non-integer images (float32 summation order must be the reference's), first/last rows
    where the wide loads run into the image guards."""
    L, R = pair(H, W, seed=H * W, integer=False)
    got = gpu_cv(eng, method, L, R, dmin, dmax, 1, win).to_host()
    exp = cpu_cv(oracle, method, L, R, dmin, dmax, 1, win)
    np.testing.assert_array_equal(got, exp)


@pytest.mark.parametrize("integer", [True, False])
@pytest.mark.parametrize("H,W,dmin,dmax,sp,win", [(30, 44, -6, 3, 1, 5), (18, 33, -2, 2, 2, 3), (26, 40, 0, 5, 1, 11), (14, 30, -3, 0, 4, 5),
                                                  (20, 70, 0, 16, 2, 5), (16, 64, -3, 5, 4, 3), (22, 90, -20, 13, 1, 7)])  # (33 / 33 / 34 cost indices)
def test_zncc(eng, oracle, integer, H, W, dmin, dmax, sp, win):
    L, R = pair(H, W, seed=5 * H + W, integer=integer)
    L[2:8, 3:12] = 7.0  # constant patch: std = 0 -> zncc must be exactly 0 (zncc.py:273-277)
    got = gpu_cv(eng, "zncc", L, R, dmin, dmax, sp, win).to_host()
    exp = cpu_cv(oracle, "zncc", L, R, dmin, dmax, sp, win)
    np.testing.assert_array_equal(np.isnan(got), np.isnan(exp))
    np.testing.assert_allclose(got, exp, rtol=0, atol=1e-5)  # north_star: float costs within 1e-5


@pytest.mark.parametrize("H,W,dmin,dmax,win", [(150, 530, -20, 37, 11), (70, 260, 3, 12, 3), (130, 300, -9, -1, 1),
                                               (66, 249, -300, -240, 5), (65, 247, 240, 262, 7), (80, 300, -5, 6, 15),
                                               (70, 200, -3, 61, 11), (67, 150, 0, 128, 5), (40, 120, -70, -5, 9)])
def test_zncc_marching_kernel_strips_tiles_chunks(eng, oracle, H, W, dmin, dmax, win):
    """subpix == 1 takes the sliding kernel: several row strips (64 rows), column tiles (256 - 2o outputs), disparity
    chunks of 8 with a ragged last chunk, ranges that leave the image entirely; non-integer images.  The last three have
    32 n + 1 / 32 n + 2 disparities: a last block of four wavefronts for one or two cost indices per pixel."""
    L, R = pair(H, W, seed=H + W, integer=False)
    got = gpu_cv(eng, "zncc", L, R, dmin, dmax, 1, win).to_host()
    exp = cpu_cv(oracle, "zncc", L, R, dmin, dmax, 1, win)
    np.testing.assert_array_equal(np.isnan(got), np.isnan(exp))
    np.testing.assert_allclose(got, exp, rtol=0, atol=1e-5)


@pytest.mark.parametrize("win", [5, 11, 15])
def test_zncc_constant_patch_below_bright_rows(eng, oracle, win):
    """window_stats_kernel keeps the sums of the window's rows in a register ring while the window moves down a strip of 8 / 32
    output rows (direct sums for the uncommon windows): a constant patch - std exactly 0, zncc exactly 0 (zncc.py:273-277) -
    directly below rows four decimal orders brighter, non-integer values, several strips.  (A window sum that SLID down - minus
    the row that leaves, plus the row that enters - would carry the bright rows' rounding into the patch and fail here.)"""
    H, W, dmin, dmax = 90, 130, -4, 5
    L, R = pair(H, W, seed=win, integer=False)
    L[:30] *= 1.0e4
    R[:30] *= 1.0e4
    L[30:70, 20:90] = 7.0  # (7 and 1234.5 have exact float32 squares: the direct sums give a variance of exactly 0)
    R[40:80, 10:70] = 1234.5
    got = gpu_cv(eng, "zncc", L, R, dmin, dmax, 1, win).to_host()
    exp = cpu_cv(oracle, "zncc", L, R, dmin, dmax, 1, win)
    np.testing.assert_array_equal(np.isnan(got), np.isnan(exp))
    np.testing.assert_allclose(got, exp, rtol=0, atol=1e-5)
    o = win // 2
    assert np.all(got[30 + o:70 - o, 20 + o:90 - o][~np.isnan(got[30 + o:70 - o, 20 + o:90 - o])] == 0.0)


def test_masks_and_variable_disparity_grids(eng, oracle):
    H, W, dmin, dmax, sp, win = 28, 41, -5, 4, 2, 5
    L, R = pair(H, W, seed=99)
    rng = np.random.default_rng(5)
    mL = rng.choice([0, 0, 0, 0, 1, 2], (H, W)).astype(np.int16)
    mR = rng.choice([0, 0, 0, 0, 0, 1, 3], (H, W)).astype(np.int16)
    gmin = rng.integers(dmin, 0, (H, W)).astype(np.float64)
    gmax = rng.integers(0, dmax + 1, (H, W)).astype(np.float64)
    for method in ("census", "sad"):
        got = gpu_cv(eng, method, L, R, dmin, dmax, sp, win, masks=(mL, mR, 0, 1), grids=(gmin, gmax)).to_host()
        exp = cpu_cv(oracle, method, L, R, dmin, dmax, sp, win, masks=(mL, mR, 0, 1), grids=(gmin, gmax))
        np.testing.assert_array_equal(got, exp)
    eng.set_masks(None, None)
    eng.set_disparity_grids(None, None)


@pytest.mark.parametrize("H,W,dmin,dmax,win,P1,P2", [(24, 37, -6, 3, 5, 8, 32), (33, 70, -60, 0, 5, 8, 32), (17, 90, 0, 128, 3, 4, 20),
                                                      (9, 11, -2, 2, 3, 8, 32), (40, 30, -70, 0, 5, 1, 2)])
def test_sgm_census_bit_exact(eng, oracle, H, W, dmin, dmax, win, P1, P2):
    """Integer costs and integer penalties: every L_r is an exact small integer in float32, so the
    8-path sum is order independent and the comparison is bit-exact."""
    L, R = pair(H, W, seed=H + W)
    cv = gpu_cv(eng, "census", L, R, dmin, dmax, 1, win)
    cpu = cpu_cv(oracle, "census", L, R, dmin, dmax, 1, win)
    eng.sgm(cv, P1, P2, False, float(win * win + 1), False)
    np.testing.assert_array_equal(cv.to_host(), oracle.sgm(cpu, P1, P2, False, float(win * win + 1), False))


@pytest.mark.parametrize("is_max,overcounting", [(False, False), (True, False), (False, True)])
def test_sgm_float_costs(eng, oracle, is_max, overcounting):
    H, W, dmin, dmax, win = 26, 45, -9, 5, 5
    L, R = pair(H, W, seed=8, integer=False)
    method = "zncc" if is_max else "sad"
    cv = gpu_cv(eng, method, L, R, dmin, dmax, 1, win)
    cpu = cv.to_host()
    inv = 2.0 if is_max else 1e4
    eng.sgm(cv, 0.5 if is_max else 8.5, 1.25 if is_max else 32.25, is_max, inv, overcounting)
    got = cv.to_host()
    exp = oracle.sgm(cpu, 0.5 if is_max else 8.5, 1.25 if is_max else 32.25, is_max, inv, overcounting)
    np.testing.assert_array_equal(np.isnan(got), np.isnan(exp))
    # same operation order as the oracle -> bit-exact; north_star tolerance for float costs is 1e-5
    np.testing.assert_allclose(got, exp, rtol=1e-6, atol=0)


@pytest.mark.parametrize("is_max", [False, True])
@pytest.mark.parametrize("H,W,D,sp", [(13, 17, 9, 1), (20, 33, 129, 1), (11, 19, 65, 2), (7, 50, 300, 1), (6, 9, 1, 1)])
def test_wta_and_refinement(eng, oracle, is_max, H, W, D, sp):
    rng = np.random.default_rng(D)
    cvh = rng.integers(0, 6, (H, W, D)).astype(np.float32)  # many ties: first extremum must win
    cvh[rng.random(cvh.shape) < 0.15] = np.nan
    cvh[0, 0, :] = np.nan
    if D > 2:
        cvh[1, 1, :] = 3.0
    L = np.zeros((H, W), np.float32)
    eng.set_images(L, L, sp)
    d0 = -4
    cv = eng.alloc_cv(D, d0)
    cv.from_host(cvh)
    val0 = rng.choice([0, 0, 0, 4, 1, 64], (H, W)).astype(np.int64)
    eng.set_validity(val0)
    eng.wta(cv, is_max, -9999.0)
    disp, val = eng.get_disparity()
    edisp, eval_ = oracle.wta(cvh, d0, sp, is_max, -9999.0, val0)
    np.testing.assert_array_equal(disp, edisp)
    np.testing.assert_array_equal(val, eval_)
    d_max = d0 + (D - 1) / sp
    for method in ("vfit", "quadratic"):
        eng.set_disparity(edisp, eval_)
        eng.refine(cv, method, is_max)
        d, v, itp = eng.get_disparity(want_itp=True)
        eitp, ed, ev = oracle.refine(cvh, edisp, eval_, d0, d_max, sp, is_max, method)
        np.testing.assert_array_equal(d, ed)
        np.testing.assert_array_equal(v, ev)
        np.testing.assert_array_equal(itp, eitp)


def test_wta_reference_known_answers(eng):
    c = ka.WTA
    for (dmin, dmax), gt in c["cases"]:
        cv = gpu_cv(eng, "sad", np.array(c["left"], np.float32), np.array(c["right"], np.float32), dmin, dmax, 1, 1)
        eng.set_validity(None)
        eng.wta(cv, False, 0.0)
        np.testing.assert_array_equal(eng.get_disparity()[0], np.array(gt, np.float32))


@pytest.mark.parametrize("H,W,dmin,dmax,sp,win,method,integer", [
    (24, 38, -5, 3, 1, 5, "census", True), (19, 31, -3, 2, 2, 3, "sad", False), (16, 30, -2, 2, 4, 1, "sad", False),
    (30, 45, -8, 0, 1, 5, "zncc", False)])
def test_cbca(eng, oracle, H, W, dmin, dmax, sp, win, method, integer):
    L, R = pair(H, W, seed=H * 7 + W, integer=integer)
    rng = np.random.default_rng(1)
    mL = rng.choice([0, 0, 0, 0, 0, 0, 1, 2], (H, W)).astype(np.int16)
    mR = rng.choice([0, 0, 0, 0, 0, 0, 1, 2], (H, W)).astype(np.int16)
    cv = gpu_cv(eng, method, L, R, dmin, dmax, sp, win, masks=(mL, mR, 0, 1))
    cpu = cv.to_host()
    off = win // 2
    eng.cbca(cv, off, 30.0, 5)

    def arms(im, msk, shifted):
        m = im.copy()
        bad = msk != 0
        if shifted:
            bad = bad[:, :-1] | bad[:, 1:]
        m[bad] = np.nan
        m = np.nan_to_num(oracle.median3(m), nan=np.inf)
        if off:
            m = m[off:-off, off:-off]
        return oracle.cross_support(np.ascontiguousarray(m), 5, 30.0)

    cl = arms(L, mL, False)
    crs = [arms(im, mR, k > 0) for k, im in enumerate(oracle.shift_right(R, sp))]
    np.testing.assert_array_equal(eng.cross_support(0, off, 30.0, 5), cl)
    for k in range(sp):
        np.testing.assert_array_equal(eng.cross_support(k + 1, off, 30.0, 5), crs[k])
    exp = oracle.cbca(cpu.copy(), dmin, sp, off, cl, crs)
    got = cv.to_host()
    np.testing.assert_array_equal(np.isnan(got), np.isnan(exp))
    np.testing.assert_array_equal(got, exp)  # sequential float32 scans, same order as the reference
    eng.set_masks(None, None)


@pytest.mark.parametrize("distance", [1, 2, 3, 5, 6, 7, 10, 11, 17, 18, 19, 26])
@pytest.mark.parametrize("loop_form", [False, True])
def test_cross_support_arm_lengths_both_forms(eng, oracle, hooks, distance, loop_form):
    """aggregation.cpp:224-321: arms by the branch-free kernel (distance <= 18) and by the loop kernel, smooth images so that the
    arms reach their limit, masked pixels, +-inf and huge values among the neighbours, every border."""
    if loop_form:
        hooks.setenv("PMX_CBCA_ARMS_FLAT", "0")
    rng = np.random.default_rng(distance)
    H, W = 37, 300
    yy, xx = np.mgrid[0:H, 0:W]
    L = (40 * np.sin(xx / 9.0) + 30 * np.cos(yy / 5.0) + rng.normal(0, 4, (H, W))).astype(np.float32)
    R = np.roll(L, 3, axis=1) + rng.normal(0, 2, (H, W)).astype(np.float32)
    L[5, 7] = np.inf; L[9, 200] = -np.inf; L[20, 20] = 3e38; L[21, 20] = -3e38; L[0, 0] = np.nan; L[H - 1, W - 1] = np.nan
    L[12:15, 100:130] = 7.0  # a constant patch: arms limited by the distance only
    mL = (rng.random((H, W)) < 0.03).astype(np.int16)
    mR = (rng.random((H, W)) < 0.03).astype(np.int16)
    for off in (0, 2):
        eng.set_images(L, R, 2)
        eng.set_masks(mL, mR, 0, 1)
        for side, (im, msk, shifted) in enumerate([(L, mL, False), (R, mR, False), (oracle.shift_right(R, 2)[1], mR, True)]):
            m = im.copy()
            bad = msk != 0
            if shifted:
                bad = bad[:, :-1] | bad[:, 1:]
            m[bad] = np.nan
            m = np.nan_to_num(oracle.median3(m), nan=np.inf)
            if off:
                m = m[off:-off, off:-off]
            exp = oracle.cross_support(np.ascontiguousarray(m), distance, 12.5)
            np.testing.assert_array_equal(eng.cross_support(side, off, 12.5, distance), exp)
    eng.set_masks(None, None)


def test_cbca_reference_known_answer(eng):
    c = ka.CBCA
    cv = gpu_cv(eng, "sad", np.array(c["left"], np.float32), np.array(c["right"], np.float32), -1, 1, 1, 1)
    eng.cbca(cv, 0, c["intensity"], c["distance"])
    np.testing.assert_allclose(cv.to_host(), np.array(c["aggregated"], np.float32), rtol=1e-7)


@pytest.mark.parametrize("case", ka.CROSS_SUPPORTS, ids=lambda c: c["cite"])
def test_cross_supports_reference_vectors(eng, case):
    """tests/test_aggregation.py:485-897 (computes_cross_supports): masks, half-pixel right image, window offset."""
    L, R = np.asarray(case["left"], np.float32), np.asarray(case["right"], np.float32)
    eng.set_images(L, R, case["subpix"])
    if case["msk_left"] is not None:
        eng.set_masks(np.array(case["msk_left"], np.int16), np.array(case["msk_right"], np.int16), 0, 1)
    off = case["win"] // 2
    if case["arms_left"] is not None:
        np.testing.assert_array_equal(eng.cross_support(0, off, case["intensity"], case["distance"]), np.array(case["arms_left"]))
    np.testing.assert_array_equal(eng.cross_support(1 + case["right_index"], off, case["intensity"], case["distance"]),
                                  np.array(case["arms_right"]))
    eng.set_masks(None, None)


@pytest.mark.parametrize("case", ka.CBCA_PIPELINES, ids=lambda c: c["cite"])
def test_cbca_pipeline_reference_vectors(eng, case):
    """tests/test_aggregation.py:91-212, 305-483: SAD -> cv_masked -> CBCA with sub-pixel volumes, masks and a window offset."""
    masks = None
    if case["msk_left"] is not None:
        masks = (np.array(case["msk_left"], np.int16), np.array(case["msk_right"], np.int16), 0, 1)
    cv = gpu_cv(eng, "sad", np.array(case["left"], np.float32), np.array(case["right"], np.float32), -1, 1, case["subpix"], case["win"],
                masks=masks)
    eng.cbca(cv, case["win"] // 2, 5.0, 3)
    got = cv.to_host()
    got = got if case["disp_index"] is None else got[:, :, case["disp_index"]]
    np.testing.assert_allclose(got, np.array(case["expected"], np.float32), rtol=1e-7)
    eng.set_masks(None, None)


@pytest.mark.parametrize("case", ka.REFINEMENT, ids=lambda c: c["cite"])
def test_refinement_reference_vectors(eng, case):
    """tests/test_refinement.py:87-655: quadratic and vfit, sub-pixel volumes, NaN neighbours, range borders."""
    cvh = np.array(case["cv"], np.float32)
    H, W, D = cvh.shape
    z = np.zeros((H, W), np.float32)
    eng.set_images(z, z, case["subpix"])
    cv = eng.alloc_cv(D, case["d_min"])
    cv.from_host(cvh)
    eng.set_disparity(np.array(case["disp"], np.float32), np.zeros((H, W), np.int64))
    eng.refine(cv, case["method"], False)
    disp, val, itp = eng.get_disparity(want_itp=True)
    np.testing.assert_array_equal(val, np.array(case["mask"]))
    np.testing.assert_allclose(disp, np.array(case["out_disp"], np.float32), rtol=2e-7, atol=0)
    np.testing.assert_allclose(itp, np.array(case["itp"], np.float32), rtol=2e-7, atol=0)
    np.testing.assert_array_equal(cv.to_host(), cvh)  # the volume is left untouched


from tests.test_oracle_golden import CENSUS_CASES, CV_MASKED_CASES, census_case_arrays, cv_masked_case_arrays  # noqa: E402


@pytest.mark.parametrize("case", CENSUS_CASES, ids=lambda c: c["id"])
def test_census_parametrised_reference_cases(eng, case):
    """tests/test_matching_cost/test_matching_cost_census.py:379-729 on the device (windows 3..13, sub-pixel volume)."""
    L, R, dmin, dmax, exp, layer = census_case_arrays(case)
    got = gpu_cv(eng, "census", L, R, dmin, dmax, case["subpix"], case["window_size"]).to_host()
    np.testing.assert_array_equal(got if layer is None else got[:, :, layer], exp)


@pytest.mark.parametrize("case", CV_MASKED_CASES, ids=lambda c: f"{c['id']}-{c['method']}")
def test_cv_masked_parametrised_reference_cases(eng, case):
    """tests/test_matching_cost/test_matching_cost.py:699-1786 on the device."""
    L, R, dmin, dmax, masks, grids, exp = cv_masked_case_arrays(case)
    cv = gpu_cv(eng, case["method"], L, R, dmin, dmax, case["subpix"], case["window_size"], masks=masks, grids=grids)
    np.testing.assert_array_equal(np.isnan(cv.to_host()), exp)
    eng.set_masks(None, None)
    eng.set_disparity_grids(None, None)


@pytest.mark.parametrize("case", ka.WTA_MORE, ids=lambda c: c["cite"])
def test_wta_more_reference_vectors(eng, case):
    """tests/test_disparity.py:255-430: window offset (invalid frame), sub-pixel volumes, first minimum / first maximum."""
    L, R = np.array(ka.WTA["left"], np.float32), np.array(ka.WTA["right"], np.float32)
    cv = gpu_cv(eng, case["method"], L, R, case["dmin"], case["dmax"], case["subpix"], case["win"], masked=case["masked"])
    eng.set_validity(None)
    eng.wta(cv, case["is_max"], float(case["invalid"]))
    disp, _ = eng.get_disparity()
    np.testing.assert_array_equal(disp, np.array(case["disp"], np.float32))


def test_refine_needs_a_disparity_map_and_never_indexes_outside_the_volume(eng, oracle):
    """pmx_refine before any WTA on the pair is a state error; a caller-provided map with disparities outside the volume
    (or NaN) keeps them and reports a NaN coefficient instead of reading out of bounds."""
    from pandora_amd.engine import PmxError

    L, R = pair(12, 20, seed=3)
    cv = gpu_cv(eng, "census", L, R, -3, 2, 1, 5)
    with pytest.raises(PmxError) as err:
        eng.refine(cv, "vfit", False)
    assert "no disparity map" in str(err.value)
    eng.sgm(cv, 8, 32, False, 26.0, False)
    with pytest.raises(PmxError):
        eng.refine(cv, "vfit", False)
    disp = np.full((12, 20), -1.0, np.float32)
    disp[0, 0], disp[1, 1], disp[2, 2], disp[3, 3] = 7.0, -40.0, np.nan, 2.0
    for _ in range(2):  # on the byte volumes, then on the materialised float32 volume
        eng.set_disparity(disp, np.zeros((12, 20), np.int64))
        eng.refine(cv, "quadratic", False)
        d, v, itp = eng.get_disparity(want_itp=True)
        for r in range(3):
            assert np.isnan(itp[r, r]) and v[r, r] == 0 and (d[r, r] == disp[r, r] or np.isnan(disp[r, r]))
        assert v[3, 3] == 8 and d[3, 3] == 2.0  # the range border stops the interpolation (refinement.cpp:62-67)
        cv.to_host()


@pytest.mark.parametrize("with_mask", [False, True])
def test_variable_disparity_ranges_stay_on_the_integer_path(eng, oracle, with_mask):
    """cv_masked with per-pixel disparity grids (the multiscale case) and / or a left mask on a deferred census volume keeps
    the integer fast path: the byte path volumes exist after SGM, and volume, disparity, validity and coefficient equal the
    oracle (NaN grids = no restriction, empty ranges, ranges that leave the volume, masked and dilated pixels)."""
    if not eng.lazy:
        pytest.skip("the integer fast path only exists in lazy mode")
    H, W, dmin, dmax, win = 31, 57, -14, 6, 5
    D = dmax - dmin + 1
    L, R = pair(H, W, seed=21)
    rng = np.random.default_rng(22)
    lo = rng.integers(dmin - 2, dmax + 1, (H, W)).astype(np.float64)
    hi = np.minimum(dmax + 3, lo + rng.integers(0, 9, (H, W)))
    lo[3, 3], hi[3, 3] = 4.0, 1.0             # empty
    lo[4, 4], hi[4, 4] = np.nan, np.nan       # no restriction
    lo[5, 5], hi[5, 5] = dmax + 5.0, dmax + 9.0  # beyond the volume
    lo[6, 6], hi[6, 6] = -2.5, 1.5            # fractional bounds: -2 .. 1
    mL = np.where(rng.random((H, W)) < 0.06, rng.choice([1, 2], (H, W)), 0).astype(np.int16) if with_mask else None
    eng.set_images(L, R, 1)
    eng.set_masks(mL, None, 0, 1)
    eng.set_disparity_grids(lo, hi)
    cv = eng.alloc_cv(D, dmin)
    eng.census(cv, win)
    eng.cv_masked(cv, win)
    nanpix = eng.nan_pixels(cv)
    eng.sgm(cv, 8, 32, False, 26.0, False)
    raw, _, _ = eng.debug_path_costs(cv, raw=True)  # raises unless the eight byte volumes exist: the fast path ran
    assert raw.shape[0] == 8
    eng.set_validity(None)
    eng.wta(cv, False, np.nan)
    eng.refine(cv, "vfit", False)
    disp, val, itp = eng.get_disparity(want_itp=True)
    ocv = oracle.census_cost(L, R, D, dmin, 1, win)
    oracle.cv_masked(ocv, dmin, 1, win, mskL=mL, mskR=None, valid=0, nodata=1, dmin=lo, dmax=hi)
    np.testing.assert_array_equal(nanpix.astype(bool), np.isnan(ocv).all(axis=2))
    ocv = oracle.sgm(ocv, 8, 32, False, 26.0, False)
    odisp, oval = oracle.wta(ocv, dmin, 1, False, np.nan)
    oitp, odisp, oval = oracle.refine(ocv, odisp, oval, dmin, dmax, 1, False, "vfit")
    np.testing.assert_array_equal(disp, odisp)
    np.testing.assert_array_equal(val, oval)
    np.testing.assert_array_equal(itp, oitp)
    np.testing.assert_array_equal(cv.to_host(), ocv)
    eng.set_masks(None, None)
    eng.set_disparity_grids(None, None)


@pytest.mark.parametrize("is_max,P1,P2", [(False, 8.0, 32.0), (True, 0.3, 1.7)])
def test_float_sgm_schedules_agree(eng, oracle, hooks, is_max, P1, P2):
    """The float32 SGM runs its eight directions one after the other (large volumes) or side by side with an ordered sum
    (small ones): both schedules (PMX_SGM_PAR=0 / 1) give the oracle's bits, overcounting and "max" measures included."""
    rng = np.random.default_rng(30)
    H, W, D = 23, 37, 70
    cvh = (rng.random((H, W, D)).astype(np.float32) * 3 - 1) if is_max else (rng.random((H, W, D)) * 40).astype(np.float32)
    cvh[rng.random(cvh.shape) < 0.1] = np.nan
    cvh[2, 3] = np.nan
    z = np.zeros((H, W), np.float32)
    eng.set_images(z, z, 1)
    for over in (False, True):
        exp = oracle.sgm(cvh, P1, P2, is_max, 45.0, over)
        for mode in ("0", "1"):
            hooks.setenv("PMX_SGM_PAR", mode)
            cv = eng.alloc_cv(D, -9)
            cv.from_host(cvh)
            eng.sgm(cv, P1, P2, is_max, 45.0, over)
            np.testing.assert_array_equal(cv.to_host(), exp)
            cv.free()


def test_float_sgm_mid_size_takes_the_marching_schedule_by_default(eng, hooks):
    """Round 6's size rule of the float32 schedules (profiles/r06_float_sched_rule.txt, r06_fam_shape.txt): from 2400 columns x 384 rows
    the marching schedule, with the 16-lane map when D > 80.  400 x 2600 x 120 (125 M cells: past the side-by-side schedule's bound; 312 000 cells per row: enough for the marching passes)
    runs marching passes by default and gives the bits of one launch per path."""
    rng = np.random.default_rng(31)
    H, W, D = 400, 2600, 120
    cvh = rng.integers(0, 40, (H, W, D)).astype(np.float32)
    cvh[rng.random((H, W)) < 0.02] = np.nan
    z = np.zeros((H, W), np.float32)
    eng.set_images(z, z, 1)
    out = {}
    for sched in (None, "seq"):
        if sched:
            hooks.setenv("PMX_SGM_SCHED", sched)
        cv = eng.alloc_cv(D, -9)
        cv.from_host(cvh)
        eng.set_profiling(True)
        eng.reset_stage_times()
        eng.sgm(cv, 8.0, 32.0, False, 45.0, False)
        eng.sync()
        launches = eng.stage_time("sgm_family")[1]
        eng.set_profiling(False)
        assert (launches > 0) == (sched is None)
        out[sched] = cv.to_host()
        cv.free()
    np.testing.assert_array_equal(out[None], out["seq"])


@pytest.mark.parametrize("H,W,D,md", [(5, 100, 129, -128), (4, 70, 300, -150), (3, 33, 1, 0), (6, 47, 64, 10), (2, 40, 950, -400)])
def test_reverse_cost_volume_tiles(eng, oracle, H, W, D, md):
    """pmx_reverse_cost_volume through its LDS-tiled kernel (32 or 16 columns per tile, partial last tile, ranges leaving the
    image on both sides) and the plain gather kept for very wide ranges == matching_cost.cpp:26-56 restated."""
    rng = np.random.default_rng(H * W + D)
    cvh = rng.random((H, W, D)).astype(np.float32)
    cvh[rng.random(cvh.shape) < 0.1] = np.nan
    z = np.zeros((H, W), np.float32)
    eng.set_images(z, z, 1)
    cv = eng.alloc_cv(D, -D + 1)
    cv.from_host(cvh)
    out = eng.reverse_cost_volume(cv, md)
    np.testing.assert_array_equal(out.to_host(), oracle.reverse_cost_volume(cvh, md))


@pytest.mark.parametrize("win", [7, 9, 11, 13])
def test_wide_census_windows_stay_on_the_integer_path(eng, oracle, win):
    """Census windows of 2, 3, 4 and 6 code words (7x7 ... 13x13) -> SGM -> WTA -> vfit on the packed integer kernels (byte
    costs; the eight byte path volumes exist afterwards) == the oracle."""
    if not eng.lazy:
        pytest.skip("the integer fast path only exists in lazy mode")
    H, W, dmin, dmax = 29, 61, -11, 7
    D = dmax - dmin + 1
    L, R = pair(H, W, seed=win)
    cv = gpu_cv(eng, "census", L, R, dmin, dmax, 1, win)
    eng.sgm(cv, 5, 40, False, float(win * win + 1), False)
    raw, gl, kpl = eng.debug_path_costs(cv, raw=True)   # raises unless the fast path ran
    assert raw.shape[0] == 8 and gl == 16
    eng.set_validity(None)
    eng.wta(cv, False, -9999.0)
    eng.refine(cv, "vfit", False)
    disp, val, itp = eng.get_disparity(want_itp=True)
    ocv = oracle.sgm(cpu_cv(oracle, "census", L, R, dmin, dmax, 1, win), 5, 40, False, float(win * win + 1), False)
    odisp, oval = oracle.wta(ocv, dmin, 1, False, -9999.0)
    oitp, odisp, oval = oracle.refine(ocv, odisp, oval, dmin, dmax, 1, False, "vfit")
    np.testing.assert_array_equal(disp, odisp)
    np.testing.assert_array_equal(val, oval)
    np.testing.assert_array_equal(itp, oitp)
    np.testing.assert_array_equal(cv.to_host(), ocv)


def test_argmin_argmax_split_and_coefficient_map(eng):
    """tests/test_disparity.py:372-473 through the plugin's static methods (argmin_split / argmax_split on sub-pixel volumes,
    coefficient_map = the winner's cost)."""
    from pandora_amd import disparity, matching_cost
    from pandora_amd.dataset import make_image

    L, R = np.array(ka.WTA["left"], np.float32), np.array(ka.WTA["right"], np.float32)
    for case in ka.WTA_MORE[3:]:
        left, right = make_image(L, disparity=[case["dmin"], case["dmax"]]), make_image(R)
        m = matching_cost.AbstractMatchingCost(matching_cost_method=case["method"], window_size=case["win"], subpix=case["subpix"])
        grids = (left["disparity"].sel(band_disp="min"), left["disparity"].sel(band_disp="max"))
        cv = m.compute_cost_volume(left, right, m.allocate_cost_volume(left, grids))
        fn = disparity.WinnerTakesAll.argmax_split if case["is_max"] else disparity.WinnerTakesAll.argmin_split
        np.testing.assert_array_equal(fn(cv), np.array(case["disp"], np.float32))
    left, right = make_image(L, disparity=[-3, 1]), make_image(R)
    m = matching_cost.AbstractMatchingCost(matching_cost_method="sad", window_size=1, subpix=1)
    grids = (left["disparity"].sel(band_disp="min"), left["disparity"].sel(band_disp="max"))
    cv = m.compute_cost_volume(left, right, m.allocate_cost_volume(left, grids))
    m.cv_masked(left, right, cv, *grids)
    d = disparity.AbstractDisparity(disparity_method="wta", invalid_disparity=0)
    d.to_disp(cv)
    np.testing.assert_array_equal(d.coefficient_map(cv).data, np.zeros((3, 4)))  # test_disparity.py:432-473


@pytest.mark.parametrize("case", ka.CROSS_SUPPORTS, ids=lambda c: c["cite"])
def test_computes_cross_supports_plugin_method(eng, case):
    """CrossBasedCostAggregation.computes_cross_supports (cbca.py:184-295) called as the reference's tests call it."""
    from pandora_amd import aggregation, matching_cost
    from pandora_amd.dataset import make_image

    mk = lambda im, msk: make_image(np.array(im, np.float32), disparity=[-1, 1], msk=None if msk is None else np.array(msk, np.int16))
    left, right = mk(case["left"], case["msk_left"]), mk(case["right"], case["msk_right"])
    m = matching_cost.AbstractMatchingCost(matching_cost_method="sad", window_size=case["win"], subpix=case["subpix"])
    grids = (left["disparity"].sel(band_disp="min"), left["disparity"].sel(band_disp="max"))
    cv = m.compute_cost_volume(left, right, m.allocate_cost_volume(left, grids))
    cbca = aggregation.AbstractAggregation(aggregation_method="cbca", cbca_intensity=case["intensity"], cbca_distance=case["distance"])
    cross_left, cross_right = cbca.computes_cross_supports(left, right, cv)
    assert len(cross_right) == case["subpix"]
    if case["arms_left"] is not None:
        np.testing.assert_array_equal(cross_left, np.array(case["arms_left"]))
    np.testing.assert_array_equal(cross_right[case["right_index"]], np.array(case["arms_right"]))


def test_placement_trials_change_nothing_but_the_buffers():
    """pmx_set_placement_trials: volumes chosen among probed candidates (>= 256 MB) - same results, argument checked."""
    from pandora_amd.engine import Engine, PmxError

    e = Engine(0)
    with pytest.raises(PmxError):
        e.set_placement_trials(0)
    e.set_placement_trials(3)
    H, W, dmin, dmax = 512, 1024, -300, 0          # 8 byte path volumes of 159 MB each: one 1.27 GB buffer goes through the probe
    L, R = pair(H, W, seed=5)
    e.set_images(L, R, 1)
    cv = e.alloc_cv(dmax - dmin + 1, dmin)
    e.census(cv, 5)
    e.sgm(cv, 8, 32, False, 26.0, False)
    e.set_validity(None)
    e.wta(cv, False, -9999.0)
    disp, val = e.get_disparity()
    e2 = Engine(0)  # the same pipeline on plain hipMalloc buffers
    e2.set_images(L, R, 1)
    cv2 = e2.alloc_cv(dmax - dmin + 1, dmin)
    e2.census(cv2, 5)
    e2.sgm(cv2, 8, 32, False, 26.0, False)
    e2.set_validity(None)
    e2.wta(cv2, False, -9999.0)
    disp2, val2 = e2.get_disparity()
    np.testing.assert_array_equal(disp, disp2)
    np.testing.assert_array_equal(val, val2)
    e.close()
    e2.close()


def test_reverse_cost_volume(eng, oracle):
    rng = np.random.default_rng(2)
    H, W, D = 9, 21, 7
    cvh = rng.random((H, W, D)).astype(np.float32)
    z = np.zeros((H, W), np.float32)
    eng.set_images(z, z, 1)
    cv = eng.alloc_cv(D, -4)
    cv.from_host(cvh)
    for md in (-2, 0, -6):
        out = eng.reverse_cost_volume(cv, md)
        np.testing.assert_array_equal(out.to_host(), oracle.reverse_cost_volume(cvh, md))


def test_full_size_properties(eng, oracle):
    """BASELINE config C3 shape class (large H*W*D): properties that need no CPU volume.
    (1) census+SGM+WTA is row-translation equivariant for the horizontal-only content: identical
        rows give identical disparities; (2) the first rows/cols of a crop equal the oracle on the
        crop for the stages without long-range coupling (census, WTA)."""
    H, W, dmin, dmax = 256, 512, 0, 128
    L, R = pair(8, W, seed=4, shift=-5)
    Lb, Rb = np.tile(L, (H // 8, 1)), np.tile(R, (H // 8, 1))
    cv = gpu_cv(eng, "census", Lb, Rb, dmin, dmax, 1, 5)
    host = cv.to_host()
    exp_rows = oracle.census_cost(Lb[:24], Rb[:24], dmax - dmin + 1, dmin, 1, 5)
    np.testing.assert_array_equal(host[2:22], exp_rows[2:22])
    eng.sgm(cv, 8, 32, False, 26.0, False)
    eng.set_validity(None)
    eng.wta(cv, False, -9999.0)
    disp, _ = eng.get_disparity()
    # vertically periodic input (period 8): interior rows far from the top/bottom borders repeat
    mid = disp[64:192]
    np.testing.assert_array_equal(mid[:64], mid[64:])


@pytest.mark.parametrize("gl,kpl,dmin,dmax", [
    (16, 4, -20, 6), (16, 5, -40, 30), (16, 8, -3, 100), (16, 9, 0, 128), (16, 12, -100, 80), (16, 13, -97, 100),
    (16, 16, -120, 120), (16, 17, 0, 256), (16, 20, -150, 150),
    (8, 8, -30, 30), (8, 9, -40, 26), (8, 12, -60, 30), (8, 13, 0, 100), (8, 16, -63, 63), (8, 17, 0, 128), (8, 17, -64, 64),
    (8, 20, -100, 50), (4, 16, -60, 0), (4, 17, 0, 64), (4, 20, -70, 5)])
def test_fused_lane_maps(eng, oracle, hooks, gl, kpl, dmin, dmax):
    """Every (lanes per scanline) x (disparities per lane) instantiation of the fused census->SGM kernel and of
    its WTA consumer, forced through PMX_FUSED_MAP on a small pair (the automatic choice would always take the
    widest group here): path costs, WTA, refinement and materialisation must equal the oracle bit for bit."""
    if not eng.lazy:
        pytest.skip("fused kernels only exist on the lazy path")
    hooks.setenv("PMX_FUSED_MAP", f"{gl}x{kpl}")
    H, W, win = 21, 83, 5
    D = dmax - dmin + 1
    assert gl * kpl > D
    L, R = pair(H, W, seed=gl * 100 + kpl, shift=-3)
    eng.set_images(L, R, 1)
    cv = eng.alloc_cv(D, dmin)
    eng.census(cv, win)
    eng.sgm(cv, 8, 32, False, float(win * win + 1), False)
    assert eng.debug_path_costs(cv, raw=True)[1:] == (gl, kpl)  # the forced map was taken
    eng.set_validity(None)
    eng.wta(cv, False, -9999.0)
    disp0, val0 = eng.get_disparity()
    eng.refine(cv, "vfit", False)
    disp, val, itp = eng.get_disparity(want_itp=True)
    c = oracle.census_cost(L, R, D, dmin, 1, win)
    s = oracle.sgm(c, 8, 32, False, float(win * win + 1), False)
    ed0, ev0 = oracle.wta(s, dmin, 1, False, -9999.0)
    np.testing.assert_array_equal(disp0, ed0)
    np.testing.assert_array_equal(val0, ev0)
    eitp, ed, ev = oracle.refine(s, ed0, ev0, dmin, dmax, 1, False, "vfit")
    np.testing.assert_array_equal(disp, ed)
    np.testing.assert_array_equal(val, ev)
    np.testing.assert_array_equal(itp, eitp)
    np.testing.assert_array_equal(cv.to_host(), s)


@pytest.mark.parametrize("sgm8", ["1", "bytes", "0"])
@pytest.mark.parametrize("win,dmin,dmax", [(5, -20, 6), (7, -70, 30), (5, 0, 128), (3, -200, 50), (5, -150, 150)])
def test_packed_and_popcount_fused_kernels_agree_with_the_oracle(eng, oracle, hooks, sgm8, win, dmin, dmax):
    """The default integer path (k_sgm8.hip: packed 16-bit recurrence on five-bit costs for windows up to 5x5, on byte
    costs otherwise or with PMX_COST5=0; KPL 4..20, one- and two-word census codes) and the popcount-fused kernel behind
    PMX_SGM8=0 (k_fused.hip), all against the oracle pipeline."""
    if not eng.lazy:
        pytest.skip("fused kernels only exist on the lazy path")
    hooks.setenv("PMX_SGM8", "0" if sgm8 == "0" else "1")
    hooks.setenv("PMX_COST5", "0" if sgm8 == "bytes" else "1")
    H, W = 19, 70
    D = dmax - dmin + 1
    L, R = pair(H, W, seed=win + D, shift=-2)
    eng.set_images(L, R, 1)
    cv = eng.alloc_cv(D, dmin)
    eng.census(cv, win)
    eng.sgm(cv, 8, 32, False, float(win * win + 1), False)
    raw, gl, kpl = eng.debug_path_costs(cv, raw=True)
    if sgm8 != "0":
        assert gl == 16 and kpl == ((D // 16 + 1) + 3) // 4 * 4  # the packed kernels' map
    eng.set_validity(None)
    eng.wta(cv, False, -9999.0)
    disp0, val0 = eng.get_disparity()
    eng.refine(cv, "quadratic", False)
    disp, val, itp = eng.get_disparity(want_itp=True)
    c = oracle.census_cost(L, R, D, dmin, 1, win)
    s = oracle.sgm(c, 8, 32, False, float(win * win + 1), False)
    ed0, ev0 = oracle.wta(s, dmin, 1, False, -9999.0)
    np.testing.assert_array_equal(disp0, ed0)
    np.testing.assert_array_equal(val0, ev0)
    eitp, ed, ev = oracle.refine(s, ed0, ev0, dmin, dmax, 1, False, "quadratic")
    np.testing.assert_array_equal(disp, ed)
    np.testing.assert_array_equal(val, ev)
    np.testing.assert_array_equal(itp, eitp)
    # the eight byte volumes sum to the oracle's aggregated cost wherever it is a number
    paths = eng.debug_path_costs(cv)
    assert paths.shape == (8, H, W, D)
    np.testing.assert_array_equal(paths.astype(np.float32).sum(axis=0)[~np.isnan(s)], s[~np.isnan(s)])
    np.testing.assert_array_equal(cv.to_host(), s)  # materialisation last (the handle becomes float32)


@pytest.mark.parametrize("win,P1,P2", [(5, 8, 32), (3, 1, 2), (7, 8, 32), (5, 8.5, 32)])
def test_fast_path_wta_refine_equal_general_path(eng, oracle, win, P1, P2):
    """Census -> SGM -> WTA -> vfit/quadratic through the handle WITHOUT downloading the volume in
    between (so the lazy engine never materialises float32) must equal the oracle pipeline; P1=8.5
    is not an integer and must silently take the general path."""
    H, W, dmin, dmax = 45, 77, -20, 6
    L, R = pair(H, W, seed=win)
    D = dmax - dmin + 1
    for method in ("vfit", "quadratic"):
        eng.set_images(L, R, 1)
        cv = eng.alloc_cv(D, dmin)
        eng.census(cv, win)
        eng.cv_masked(cv, win)
        nanpix = eng.nan_pixels(cv)
        eng.sgm(cv, P1, P2, False, float(win * win + 1), False)
        eng.set_validity(None)
        eng.wta(cv, False, -9999.0)
        disp0, val0 = eng.get_disparity()
        eng.refine(cv, method, False)
        disp, val, itp = eng.get_disparity(want_itp=True)
        c = oracle.census_cost(L, R, D, dmin, 1, win)
        np.testing.assert_array_equal(nanpix, np.min(np.isnan(c), axis=2))
        s = oracle.sgm(c, P1, P2, False, float(win * win + 1), False)
        ed0, ev0 = oracle.wta(s, dmin, 1, False, -9999.0)
        np.testing.assert_array_equal(disp0, ed0)
        np.testing.assert_array_equal(val0, ev0)
        eitp, ed, ev = oracle.refine(s, ed0, ev0, dmin, dmax, 1, False, method)
        np.testing.assert_array_equal(disp, ed)
        np.testing.assert_array_equal(val, ev)
        np.testing.assert_array_equal(itp, eitp)
        np.testing.assert_array_equal(cv.to_host(), s)  # materialisation last


def test_streaming_pairs_reuse_device_buffers_and_host_outputs(eng, oracle):
    """A second pair of the same shape reuses the device buffers (pmx_set_images) and the caller's output arrays
    (get_disparity(out=...)); masks of the first pair must not leak into the second."""
    H, W, dmin, dmax, win = 33, 61, -7, 5, 5
    D = dmax - dmin + 1
    outs = None
    for seed, masked in ((1, True), (2, False)):
        L, R = pair(H, W, seed=seed)
        eng.set_images(L, R, 1)
        if masked:
            m = np.zeros((H, W), np.int16)
            m[5:9, 10:30] = 1
            eng.set_masks(m, None, 0, 1)
        cv = eng.alloc_cv(D, dmin)
        eng.census(cv, win)
        eng.cv_masked(cv, win)
        eng.sgm(cv, 8, 32, False, float(win * win + 1), False)
        eng.set_validity(None)
        eng.wta(cv, False, -9999.0)
        outs = eng.get_disparity(want_itp=False, out=outs)
    c = oracle.census_cost(L, R, D, dmin, 1, win)
    s = oracle.sgm(c, 8, 32, False, float(win * win + 1), False)
    ed, ev = oracle.wta(s, dmin, 1, False, -9999.0)
    np.testing.assert_array_equal(outs[0], ed)
    np.testing.assert_array_equal(outs[1], ev)
    with pytest.raises(ValueError):
        eng.get_disparity(out=(np.empty((H, W), np.float64), outs[1]))


@pytest.mark.parametrize("case", ka.CROSS_CHECKING, ids=lambda c: c["cite"])
def test_cross_checking_reference_vectors(eng, case):
    val, conf = eng.cross_checking(np.array(case["left"], np.float32), np.array(case["validity"], np.int64),
                                   np.array(case["right"], np.float32), case["interval"][0], case["interval"][1], case["threshold"])
    if case["conf"] is not None:
        np.testing.assert_array_equal(conf, np.array(case["conf"], np.float32))
    if case["mask"] is not None:
        np.testing.assert_array_equal(val, np.array(case["mask"], np.int64))


@pytest.mark.parametrize("H,W,dmin,dmax,thr", [(37, 300, -20, 9, 1.0), (5, 64, 0, 12, 0.0), (64, 515, -3, 3, 0.5)])
def test_cross_checking_random_maps(eng, oracle, H, W, dmin, dmax, thr):
    """Float (refined) disparities, invalid pixels, NaN in the right map, pixels whose match leaves the row."""
    rng = np.random.default_rng(H * W)
    left = (rng.integers(dmin, dmax + 1, (H, W)) + rng.choice([0.0, 0.25, -0.5, 0.5], (H, W))).astype(np.float32)
    right = -np.roll(left, 2, axis=1) + rng.choice([0.0, 0.0, 1.0, -2.0, 0.5], (H, W)).astype(np.float32)
    right[rng.random((H, W)) < 0.02] = np.nan
    val = np.where(rng.random((H, W)) < 0.1, 1 << 1, 0).astype(np.int64) | np.where(rng.random((H, W)) < 0.05, 1 << 3, 0)
    got_val, got_conf = eng.cross_checking(left, val, right, dmin, dmax, thr)
    exp_val, exp_conf = oracle.cross_checking(left, val, right, dmin, dmax, thr)
    np.testing.assert_array_equal(got_val, exp_val)
    np.testing.assert_array_equal(got_conf, exp_conf)
    assert (exp_val & ((1 << 8) | (1 << 9))).any()


def test_reverse_disp_range(eng, oracle):
    rng = np.random.default_rng(12)
    for H, W in ((7, 33), (20, 300)):
        lo = rng.integers(-9, 4, (H, W)).astype(np.float32)
        hi = lo + rng.integers(0, 8, (H, W)).astype(np.float32)
        lo[0, 3] = np.nan
        hi[1, 5] = np.nan
        lo[2, :] = np.nan  # a whole row without a range -> NaN on the right
        got = eng.reverse_disp_range(lo, hi)
        exp = oracle.reverse_disp_range(lo, hi)
        np.testing.assert_array_equal(got[0], exp[0])
        np.testing.assert_array_equal(got[1], exp[1])


@pytest.mark.parametrize("case", ka.MEDIAN, ids=lambda c: c["cite"])
def test_median_filter_reference_vectors(eng, case):
    got = eng.median_filter_disparity(np.array(case["disp"], np.float32), np.array(case["valid"], np.int64), case.get("size", 3))
    np.testing.assert_array_equal(got, np.array(case["expected"], np.float32))


@pytest.mark.parametrize("size", [1, 3, 5, 7, 9])
def test_median_filter_random_maps(eng, oracle, size):
    """Sub-pixel disparities, ~20 % invalid pixels (ignored inside windows, untouched themselves), NaN and inf values,
    even counts (mean of the two middles in float32), frame of size/2 pixels untouched; register-sorted and generic kernels."""
    H, W = 41, 300
    rng = np.random.default_rng(size)
    disp = (rng.integers(-40, 10, (H, W)) + rng.choice([0.0, 0.25, 0.5, -0.125], (H, W))).astype(np.float32)
    val = np.where(rng.random((H, W)) < 0.2, rng.choice([1, 2, 64, 128, 256, 512], (H, W)), 0).astype(np.int64)
    val |= np.where(rng.random((H, W)) < 0.1, 4 | 8, 0)  # information bits only: still valid
    disp[3, 7] = np.nan
    disp[9, 11] = np.inf
    got = eng.median_filter_disparity(disp, val, size)
    exp = oracle.filter_median_disparity(disp, val, size)
    np.testing.assert_array_equal(got, exp)
    assert not np.array_equal(exp, disp) or size == 1


@pytest.mark.parametrize("H,W,sc,ss", [(5, 5, 4.0, 6.0), (40, 61, 2.0, 6.0), (33, 50, 2.0, 1.5), (25, 30, 3.0, 1.0)])
def test_bilateral_filter(eng, oracle, H, W, sc, ss):
    """bilateral.py:100-255 on the device against the float64 restatement (window 19 / 5 / even window 4, window clipped
    to the image, invalid and NaN neighbours ignored, invalid pixels untouched): 1e-6 relative."""
    rng = np.random.default_rng(H * W)
    disp = (rng.integers(-30, 5, (H, W)) + rng.random((H, W))).astype(np.float32)
    val = np.where(rng.random((H, W)) < 0.15, rng.choice([1, 2, 64, 256], (H, W)), 0).astype(np.int64)
    disp[H // 2, W // 3] = np.nan
    got = eng.bilateral_filter_disparity(disp, val, sc, ss)
    exp = oracle.filter_bilateral_disparity(disp, val, sc, ss)
    np.testing.assert_allclose(got, exp, rtol=1e-6, atol=2e-6, equal_nan=True)  # (atol: weighted means that cancel to ~0; the
    # device's expf and libm's differ by an ulp of a weight, against disparities of tens - tools/fuzz_filters.py)
    inv = (val & 0x3C3) != 0
    np.testing.assert_array_equal(got[inv], disp[inv])


@pytest.mark.parametrize("H,W,ws", [(2, 2, 3), (5, 5, 11), (40, 61, 11), (33, 50, 5), (7, 90, 1)])
def test_disparity_denoiser(eng, oracle, H, W, ws):
    """disparity_denoiser.py:223-313 on the device against the restatement (pinned by the reference's vectors,
    tests/test_oracle_golden.py): windows larger than the image (numpy's reflect padding wraps more than once), NaN disparities
    (every window that sees one turns NaN), invalid pixels untouched, window 1: 1e-6 relative."""
    from scipy.ndimage import gaussian_filter

    rng = np.random.default_rng(H * W + ws)
    disp = (rng.integers(-30, 5, (H, W)) + rng.random((H, W))).astype(np.float32)
    band = rng.integers(0, 4096, (H, W)).astype(np.float32)
    val = np.where(rng.random((H, W)) < 0.15, rng.choice([1, 2, 64, 256], (H, W)), 0).astype(np.int64)
    if H > 8:
        disp[H // 2, W // 3] = np.nan
    grad = np.gradient(gaussian_filter(disp, sigma=1.5)) if min(H, W) > 1 else [np.zeros_like(disp)] * 2
    got = eng.denoise_disparity(disp, val, band, grad[0], grad[1], ws, 4.0, 100.0, 12.0)
    exp = oracle.denoise_disparity(disp, val, band, grad[0], grad[1], ws, 4.0, 100.0, 12.0)
    np.testing.assert_allclose(got, exp, rtol=1e-6, atol=1e-6, equal_nan=True)
    inv = (val & 0x3C3) != 0
    np.testing.assert_array_equal(got[inv], disp[inv])


def test_disparity_range_reference_vector_and_random(eng, oracle):
    c = ka.DISPARITY_RANGE
    lo, hi = eng.disparity_range(np.array(c["disp"], np.float32), np.array(c["validity"], np.int64), c["window_size"], c["marge"],
                                 c["dmin"], c["dmax"])
    np.testing.assert_array_equal(lo, np.array(c["range_min"], np.float32))
    np.testing.assert_array_equal(hi, np.array(c["range_max"], np.float32))
    rng = np.random.default_rng(4)
    for H, W, win, marge in ((33, 70, 5, 1), (20, 300, 3, 0), (12, 12, 11, 2)):
        disp = (rng.integers(-40, 3, (H, W)) + rng.choice([0.0, 0.5, -0.25], (H, W))).astype(np.float32)
        val = np.where(rng.random((H, W)) < 0.2, rng.choice([1, 2, 256], (H, W)), 0).astype(np.int64)
        disp[1, 2] = np.nan
        got = eng.disparity_range(disp, val, win, marge, -41, 3)
        exp = oracle.disparity_range(disp, val, win, marge, -41, 3)
        np.testing.assert_array_equal(got[0], exp[0])
        np.testing.assert_array_equal(got[1], exp[1])


@pytest.mark.parametrize("case", ka.INTERPOLATION, ids=lambda c: c["cite"])
def test_interpolation_reference_vectors(eng, case):
    """validation.AbstractInterpolation on the reference's own five cases (tests/test_validation.py:310-704), through the
    plugin classes."""
    from pandora_amd import validation
    from pandora_amd.dataset import Dataset

    left = Dataset({"disparity_map": (("row", "col"), np.array(case["disp"], np.float32)),
                    "validity_mask": (("row", "col"), np.array(case["validity"], np.uint16))},
                   coords={"row": np.arange(len(case["disp"])), "col": np.arange(len(case["disp"][0]))})
    left.attrs["offset_row_col"] = 0
    validation.AbstractInterpolation(interpolated_disparity=case["method"]).interpolated_disparity(left)
    np.testing.assert_array_equal(left["validity_mask"].data, np.array(case["out_validity"]))
    np.testing.assert_array_equal(left["disparity_map"].data, np.array(case["out_disp"], np.float32))
    assert left.attrs["interpolated_disparity"] == case["method"]


@pytest.mark.parametrize("which", ["occlusion_mc_cnn", "mismatch_mc_cnn", "occlusion_sgm", "mismatch_sgm"])
def test_interpolation_passes_equal_oracle(eng, oracle, which):
    """Each pass of pmx_interpolate_disparity == its restatement (diffed against the compiled validation_cpp in
    test_oracle_vs_reference.py): dense / sparse rejections, rows without a valid pixel, NaN on valid pixels, 1-wide maps."""
    OCC, MIS = 1 << 8, 1 << 9
    rng = np.random.default_rng(13)
    for H, W, p in ((6, 9, 0.3), (33, 300, 0.6), (64, 257, 0.1), (1, 40, 0.5), (40, 1, 0.5), (8, 8, 1.0), (50, 70, 0.0)):
        disp = (rng.integers(-20, 20, (H, W)) + rng.choice([0, 0.25, -0.5], (H, W))).astype(np.float32)
        valid = np.where(rng.random((H, W)) < p, rng.choice([OCC, MIS, 1, 2, 64, OCC + 4, MIS + 8], (H, W)),
                         rng.choice([0, 4, 8, 16, 32, 2048], (H, W))).astype(np.int64)
        if H > 2 and W > 2:
            valid[2, :] = OCC
            disp[1, 1], valid[1, 1] = np.nan, 0
        if p == 0.0:
            valid[10:40, 20:60] = MIS            # a large hole: long walks
            valid[12, 18] = OCC
        got = eng.interpolate_disparity(disp, valid, [which])
        exp = oracle.interpolate_disparity(which, disp, valid)
        np.testing.assert_array_equal(got[0], exp[0])
        np.testing.assert_array_equal(got[1], exp[1])
    # two passes in one call == two calls
    a = eng.interpolate_disparity(disp, valid, ["mismatch_sgm", "occlusion_sgm"])
    b = eng.interpolate_disparity(*eng.interpolate_disparity(disp, valid, ["mismatch_sgm"]), ["occlusion_sgm"])
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])


def test_interpolate_nodata_equals_oracle(eng, oracle):
    """pmx_interpolate_nodata == img_tools.cpp:99-155 restated (pinned against the compiled reference in
    test_oracle_vs_reference.py): sparse and dense masks, ragged widths, all-invalid and all-valid images, NaN values among
    the neighbours, paths that leave the image."""
    rng = np.random.default_rng(8)
    for H, W, p in ((9, 13, 0.3), (33, 300, 0.05), (64, 257, 0.7), (5, 5, 1.0), (7, 9, 0.0), (1, 40, 0.5), (40, 1, 0.5)):
        img = (rng.random((H, W)) * 255).astype(np.float32)
        msk = np.where(rng.random((H, W)) < p, rng.choice([1, 2, 64, 3], (H, W)), rng.choice([0, 4, 1024], (H, W))).astype(np.int32)
        if H > 3 and W > 3:
            img[1, 2] = np.nan
            msk[1, 2] = 0
        got = eng.interpolate_nodata(img, msk, 0b01111000011, 1 << 10)
        exp = oracle.interpolate_nodata(img, msk, 0b01111000011, 1 << 10)
        np.testing.assert_array_equal(got[0], exp[0])
        np.testing.assert_array_equal(got[1], exp[1])
    blob = np.zeros((50, 70), np.int32)
    blob[10:40, 20:60] = 1                       # a large hole: long walks, even/odd neighbour counts
    blob[:, 0] = 2
    img = (rng.random((50, 70)) * 255).astype(np.float32)
    got, exp = eng.interpolate_nodata(img, blob, 0b01111000011, 1 << 10), oracle.interpolate_nodata(img, blob, 0b01111000011, 1 << 10)
    np.testing.assert_array_equal(got[0], exp[0])
    np.testing.assert_array_equal(got[1], exp[1])
    with pytest.raises(ValueError):
        eng.interpolate_nodata(img, blob[:, :5], 1, 2)


def test_row_tiled_local_pipeline_equals_full_image(eng):
    """Row tiles with a margin of the window radius reproduce the untiled census -> WTA -> vfit result exactly (no
    data-path collective; pandora_amd.dist.row_tile / crop_tile / stitch_tiles)."""
    from pandora_amd import dist as pd

    H, W, dmin, dmax, win = 57, 64, -9, 4, 5
    D = dmax - dmin + 1
    L, R = pair(H, W, seed=11)

    def run(left, right):
        eng.set_images(left, right, 1)
        cv = eng.alloc_cv(D, dmin)
        eng.census(cv, win)
        eng.set_validity(None)
        eng.wta(cv, False, -9999.0)
        eng.refine(cv, "vfit", False)
        return eng.get_disparity(want_itp=True)

    full = run(L, R)
    world = 3
    parts = [[], [], []]
    for rank in range(world):
        (lo, hi), (rlo, rhi) = pd.row_tile(H, world, rank, margin=win // 2)
        out = run(L[rlo:rhi], R[rlo:rhi])
        for k in range(3):
            parts[k].append(pd.crop_tile(out[k], H, world, rank, margin=win // 2))
    for k in range(3):
        got = pd.stitch_tiles(parts[k])
        if k == 1:  # validity: tile borders inside the image are not image borders
            inner = np.zeros(H, bool)
            inner[win // 2:H - win // 2] = True
            np.testing.assert_array_equal(got[inner], full[k][inner])
        else:
            np.testing.assert_array_equal(got[win // 2:H - win // 2], full[k][win // 2:H - win // 2])


@pytest.mark.parametrize("H,W,dmin,dmax,sp,negate", [(21, 37, -9, 6, 1, False), (12, 30, -3, 4, 2, True), (9, 70, 0, 40, 1, False)])
def test_ambiguity_integral(eng, oracle, H, W, dmin, dmax, sp, negate):
    """ambiguity.cpp:28-142 on the device against the restatement pinned by the compiled reference: NaN costs in and out
    of the per-pixel range, pixels without any cost, similarity measures (sign flip), 70 etas."""
    rng = np.random.default_rng(H * W)
    D = (dmax - dmin) * sp + 1
    L, R = pair(H, W, seed=H)
    eng.set_images(L, R, sp)
    cv = eng.alloc_cv(D, dmin)
    vol = (rng.random((H, W, D)) * 50 - 10).astype(np.float32)
    vol[rng.random((H, W, D)) < 0.1] = np.nan
    vol[2, 3, :] = np.nan
    cv.from_host(vol)
    gmin = rng.integers(dmin, dmin + 3, (H, W)).astype(np.int64)
    gmax = (gmin + rng.integers(1, dmax - dmin - 2, (H, W))).astype(np.int64)
    etas = np.arange(0.0, 0.7, 0.01)
    got = eng.ambiguity(cv, etas, gmin, gmax, negate)
    disp_range = (dmin + np.arange(D) / sp).astype(np.float32)
    exp = oracle.ambiguity(-vol if negate else vol, etas, gmin, gmax, disp_range)
    np.testing.assert_array_equal(got, exp)
    assert got[2, 3] == len(etas) * D


def _confidence_volume(eng, H, W, dmin, dmax, sp, quantised, seed):
    rng = np.random.default_rng(seed)
    D = (dmax - dmin) * sp + 1
    L, R = pair(H, W, seed=H)
    eng.set_images(L, R, sp)
    cv = eng.alloc_cv(D, dmin)
    vol = rng.integers(0, 14, (H, W, D)).astype(np.float32) if quantised else (rng.random((H, W, D)) * 50 - 10).astype(np.float32)
    vol[rng.random((H, W, D)) < 0.1] = np.nan
    vol[2, 3, :] = np.nan
    vol[4, 5, 1:] = np.nan
    cv.from_host(vol)
    gmin = rng.integers(dmin, dmin + 3, (H, W)).astype(np.int64)
    gmax = (gmin + rng.integers(1, dmax - dmin - 2, (H, W))).astype(np.int64)
    return cv, vol, gmin, gmax, (dmin + np.arange(D) / sp).astype(np.float32)


@pytest.mark.parametrize("quantised", [False, True])
@pytest.mark.parametrize("H,W,dmin,dmax,sp,negate", [(21, 37, -9, 6, 1, False), (12, 30, -3, 4, 2, True), (9, 70, 0, 40, 1, False),
                                                     (8, 33, -70, 66, 1, False), (7, 19, -5, 4, 4, True)])
def test_risk(eng, oracle, H, W, dmin, dmax, sp, negate, quantised):
    """risk.cpp:28-197 (+ the sampled ambiguity of ambiguity.cpp it consumes) on the device against the restatement pinned by
    the compiled reference: four maps, bit for bit; D not a multiple of the 16-disparity chunks, ties, NaN in and out of range."""
    cv, vol, gmin, gmax, disp_range = _confidence_volume(eng, H, W, dmin, dmax, sp, quantised, H * W)
    etas = np.arange(0.0, 0.7, 0.01)
    got = eng.risk(cv, etas, gmin, gmax, negate)
    exp = oracle.risk(-vol if negate else vol, etas, gmin, gmax, disp_range)
    for g, e, name in zip(got, exp, ("risk_max", "risk_min", "disp_sup", "disp_inf")):
        np.testing.assert_array_equal(g, e, err_msg=name)
    assert np.isnan(got[0][2, 3])
    with pytest.raises(Exception, match="ascending"):
        eng.risk(cv, [0.0, 0.2, 0.1], gmin, gmax)


@pytest.mark.parametrize("sp", [1, 2])
def test_confidence_without_grids_searches_the_whole_range(eng, oracle, sp):
    """grid_min == grid_max == NULL (what the plugins pass for constant [min, max] inputs) == grids holding the volume's range."""
    H, W, dmin, dmax = 11, 29, -7, 5
    cv, vol, _, _, disp_range = _confidence_volume(eng, H, W, dmin, dmax, sp, True, 77)
    gmin, gmax = np.full((H, W), dmin, np.int64), np.full((H, W), dmax, np.int64)
    etas = np.arange(0.0, 0.7, 0.01)
    np.testing.assert_array_equal(eng.ambiguity(cv, etas, None, None), oracle.ambiguity(vol, etas, gmin, gmax, disp_range))
    for g, e in zip(eng.risk(cv, etas, None, None), oracle.risk(vol, etas, gmin, gmax, disp_range)):
        np.testing.assert_array_equal(g, e)
    for g, e in zip(eng.interval_bounds(cv, 0.9, -1.0, None, None), oracle.interval_bounds(vol, 0.9, -1.0, gmin, gmax, disp_range)):
        np.testing.assert_array_equal(g, e)
    with pytest.raises(Exception):  # one grid without the other
        eng.risk(cv, etas, gmin, None)


@pytest.mark.parametrize("case", ka.RISK, ids=lambda c: c["cite"])
def test_risk_reference_vectors(eng, case):
    vol, etas = np.array(case["cv"], np.float32), np.array(case["etas"])
    H, W, D = vol.shape
    eng.set_images(np.zeros((H, W), np.float32), np.zeros((H, W), np.float32), 1)
    cv = eng.alloc_cv(D, int(case["disp_range"][0]))
    cv.from_host(vol)
    risk_max, _, sup, inf = eng.risk(cv, etas, np.array(case["grid_min"]), np.array(case["grid_max"]))
    np.testing.assert_allclose(risk_max, np.array(case["risk_max"], np.float32), rtol=1e-6)
    np.testing.assert_allclose(sup, np.array(case["disp_sup"], np.float32), rtol=1e-6)
    np.testing.assert_allclose(inf, np.array(case["disp_inf"], np.float32), rtol=1e-6)


@pytest.mark.parametrize("quantised", [False, True])
@pytest.mark.parametrize("thr,tf", [(0.9, -1.0), (0.5, 1.0), (1.0, -1.0), (0.0, 1.0)])
@pytest.mark.parametrize("H,W,dmin,dmax,sp", [(21, 37, -9, 6, 1), (12, 30, -3, 4, 2), (9, 70, 0, 40, 1), (7, 19, -5, 4, 4)])
def test_interval_bounds(eng, oracle, H, W, dmin, dmax, sp, thr, tf, quantised):
    """interval_bounds.cpp:28-161 on the device, bit for bit against the pinned restatement."""
    cv, vol, gmin, gmax, disp_range = _confidence_volume(eng, H, W, dmin, dmax, sp, quantised, H + W)
    got = eng.interval_bounds(cv, thr, tf, gmin, gmax)
    exp = oracle.interval_bounds(vol, thr, tf, gmin, gmax, disp_range)
    np.testing.assert_array_equal(got[0], exp[0])
    np.testing.assert_array_equal(got[1], exp[1])


def test_interval_bounds_reference_vector(eng):
    c = ka.INTERVAL_BOUNDS
    vol = np.array(c["cv"], np.float32)
    eng.set_images(np.zeros((4, 4), np.float32), np.zeros((4, 4), np.float32), 1)
    cv = eng.alloc_cv(3, -1)
    cv.from_host(vol)
    g = np.ones((4, 4), np.int64)
    lo, hi = eng.interval_bounds(cv, c["threshold"], c["type_factor"], -g, g)
    np.testing.assert_allclose(lo, np.array(c["inf"], np.float32), rtol=1e-6)
    np.testing.assert_allclose(hi, np.array(c["sup"], np.float32), rtol=1e-6)


@pytest.mark.parametrize("fast", ["1", "4", "0"])
@pytest.mark.parametrize("H,W,dmin,dmax,sp,dist", [(70, 150, -12, 5, 1, 5), (45, 90, -4, 3, 2, 9), (40, 61, 0, 9, 1, 2)])
def test_cbca_phase_split_and_generic_kernels(eng, oracle, hooks, fast, H, W, dmin, dmax, sp, dist):
    """CBCA on images large enough for the phase-split kernels (warm-up / steady / drain, four steps in flight), with
    PMX_CBCA_FAST=4 through the in-place four-disparities-per-thread kernels (subpix 1, short arms; else phase-split) and,
    with PMX_CBCA_FAST=0, through the generic ones: all bit-exact against the reference-pinned oracle (sequential fp32
    prefix sums), with masks, sub-pixel volumes, long arms."""
    hooks.setenv("PMX_CBCA_FAST", fast)
    L, R = pair(H, W, seed=H + dist, integer=True)
    rng = np.random.default_rng(dist)
    mskL = rng.choice([0, 0, 0, 0, 0, 0, 1], (H, W)).astype(np.int16)
    mskR = rng.choice([0, 0, 0, 0, 0, 0, 2], (H, W)).astype(np.int16)
    win, off = 5, 2
    cv = gpu_cv(eng, "census", L, R, dmin, dmax, sp, win, masks=(mskL, mskR, 0, 1))
    eng.cbca(cv, off, 30.0, dist)
    got = cv.to_host()
    exp = cpu_cv(oracle, "census", L, R, dmin, dmax, sp, win, masks=(mskL, mskR, 0, 1))

    def arms(im, msk, shifted):
        m = im.copy()
        bad = msk != 0
        if shifted:
            bad = bad[:, :-1] | bad[:, 1:]
        m[bad] = np.nan
        m = np.nan_to_num(oracle.median3(m), nan=np.inf)[off:-off, off:-off]
        return oracle.cross_support(np.ascontiguousarray(m), dist, 30.0)

    cl = arms(L, mskL, False)
    crs = [arms(im, mskR, k > 0) for k, im in enumerate(oracle.shift_right(R, sp))]
    oracle.cbca(exp, dmin, sp, off, cl, crs)
    np.testing.assert_array_equal(got, exp)


@pytest.mark.parametrize("rows,vbuf", [(None, "0"), ("1", "1"), ("3", "1")])
@pytest.mark.parametrize("H,W,dmin,dmax,dist,with_grids,with_left_mask", [
    (70, 150, -12, 5, 5, False, False), (41, 67, 0, 60, 5, True, False), (45, 91, -30, 3, 3, True, True),
    (40, 203, -64, 64, 5, False, True), (38, 77, -5, 4, 9, False, False), (33, 52, -3, 3, 2, True, False),
    (60, 120, -64, 64, 12, False, False), (70, 110, -5, 4, 24, False, False), (80, 110, -3, 2, 32, True, False)])
def test_cbca_whole_rows_and_census_source(eng, oracle, hooks, rows, vbuf, H, W, dmin, dmax, dist, with_grids, with_left_mask):
    """Census + CBCA without a right mask: in lazy mode the census costs are still implicit (codes) when pmx_cbca runs, and pass H
    computes them on the fly - with the per-pixel valid intervals of cv_masked when grids / a left mask are resident - so the
    float volume first exists as the aggregated one (its border cells NaN); in eager mode the same whole-row pass H reads the
    float volume.  Both carry the NaN flags of the input to pass V in the sign bit of E_h.  Widths that are no multiple of the
    flush chunk, D = 61 / 129 / 16, arms from 1 to 8 columns, 1 to 3 rows per workgroup: bit-exact against the oracle.  Arms of 11
    columns at D = 129 need a 64-slot ring (the rows per workgroup follow the LDS); cbca_distance 24 and 32 need 128 slots and run
    through the generic kernels with 64-thread workgroups (they used to fail at launch: 256 KB of LDS)."""
    if rows:
        hooks.setenv("PMX_CBCA_ROWS", rows)
    hooks.setenv("PMX_CBCA_MARCH", "0")  # (plain census geometry with short arms would take the one-kernel route: test_cbca_census_march)
    hooks.setenv("PMX_CBCA_VBUF", vbuf)  # pass V with pointers (what small volumes get) / through buffer instructions (large ones)
    if rows == "3":
        hooks.setenv("PMX_CBCA_VBS", "512")  # ... in the 512-thread workgroups the largest volumes get
    L, R = pair(H, W, seed=H + W + dist, integer=True)
    rng = np.random.default_rng(H * dist)
    win, off = 5, 2
    mskL = rng.choice([0, 0, 0, 0, 0, 0, 0, 1], (H, W)).astype(np.int16) if with_left_mask else None
    grids = None
    if with_grids:
        gmin = rng.integers(dmin, dmin + 3, (H, W)).astype(np.float64)
        gmax = rng.integers(dmax - 3, dmax + 1, (H, W)).astype(np.float64)
        grids = (gmin, gmax)
    masks = (mskL, None, 0, 1) if with_left_mask else None
    eng.set_profiling(True)
    eng.reset_stage_times()
    cv = gpu_cv(eng, "census", L, R, dmin, dmax, 1, win, masks=masks, grids=grids)
    eng.cbca(cv, off, 30.0, dist)
    got = cv.to_host()
    cost_launches = eng.stage_time("census_cost")[1]
    eng.set_profiling(False)
    assert cost_launches == (0 if eng.lazy and dist <= 21 else 1)  # lazy: no census cost kernel ever ran (arms that fit the rows kernel)
    exp = cpu_cv(oracle, "census", L, R, dmin, dmax, 1, win, masks=masks, grids=grids)

    def arms(im, msk):
        m = im.copy()
        if msk is not None:
            m[msk != 0] = np.nan
        m = np.nan_to_num(oracle.median3(m), nan=np.inf)[off:-off, off:-off]
        return oracle.cross_support(np.ascontiguousarray(m), dist, 30.0)

    oracle.cbca(exp, dmin, 1, off, arms(L, mskL), [arms(R, None)])
    np.testing.assert_array_equal(got, exp)
    eng.set_masks(None, None)
    eng.set_disparity_grids(None, None)


def test_small_integer_division_of_the_marching_kernel(eng):
    """every quotient a / b, a < 65536, 1 <= b <= 1024, of the kernel's reciprocal + Newton step equals the IEEE division bit for bit"""
    assert eng.debug_small_division() == 0


def test_cbca_crop_wider_than_the_census_border(eng, oracle):
    """offset 3 with a 5x5 census: the ring of cells between the census border and the crop keeps its costs, so the volume has to
    exist before the aggregation (pmx_cbca_can_fuse_census says no; the census-source kernels used to write NaN there)."""
    H, W, dmin, dmax, dist, win, off = 47, 83, -20, 6, 5, 5, 3
    L, R = pair(H, W, seed=H + W + dist, integer=True)
    cv = gpu_cv(eng, "census", L, R, dmin, dmax, 1, win)
    eng.cbca(cv, off, 30.0, dist)
    got = cv.to_host()
    exp = cpu_cv(oracle, "census", L, R, dmin, dmax, 1, win)

    def arms(im):
        m = np.nan_to_num(oracle.median3(im.copy()), nan=np.inf)[off:-off, off:-off]
        return oracle.cross_support(np.ascontiguousarray(m), dist, 30.0)

    oracle.cbca(exp, dmin, 1, off, arms(L), [arms(R)])
    np.testing.assert_array_equal(got, exp)


@pytest.mark.parametrize("H,W,dmin,dmax,dist,win,off,with_grids,with_left_mask", [
    (70, 150, -12, 5, 5, 5, 2, False, False),      # D = 18: 8 columns per workgroup, two staging words per thread
    (41, 67, 0, 60, 5, 5, 2, False, False),        # D = 61
    (45, 203, -64, 64, 5, 5, 2, False, False),     # D = 129: 7 columns per workgroup (the BASELINE configurations' D), two halo cells per thread
    (33, 52, -6, 6, 2, 5, 2, False, False),        # arms of one pixel, D = 13 (fewer than 11 disparities keep passes H and V)
    (38, 77, -5, 4, 3, 3, 1, False, False),        # census 3x3
    (52, 90, -100, 99, 4, 5, 2, False, False),     # D = 200: 5 columns per workgroup
    (40, 300, 0, 255, 5, 5, 2, False, False),      # D = 256: 4 columns per workgroup
    (47, 83, -20, 6, 5, 5, 1, False, False),       # crop narrower than the census border: the general geometry test
    (36, 61, -7, 7, 5, 5, 0, False, False),        # no crop at all
    (20, 20, -8, 7, 5, 5, 2, False, False),        # 16 x 16 cropped pixels: the smallest image the long-scan kernels take
    (41, 67, 0, 60, 5, 5, 2, True, False),         # per-pixel disparity ranges: the valid intervals cv_masked left decide what is a cost
    (45, 203, -64, 64, 5, 5, 2, True, True),       # ... and a left mask, D = 129
    (38, 77, -5, 4, 3, 3, 1, False, True),
])
def test_cbca_census_march(eng, oracle, H, W, dmin, dmax, dist, win, off, with_grids, with_left_mask):
    """cbca_census_march_kernel (census costs, horizontal and vertical scans in one marching kernel on exact integer sums, no E_h
    volume; the default whenever it is legal): bit-exact against the oracle, which follows the reference's float32 scans - all integers
    below 2^24.  Passes H and V must not have run."""
    if not eng.lazy:
        pytest.skip("census codes are only kept in lazy mode")
    L, R = pair(H, W, seed=H + W + dist, integer=True)
    rng = np.random.default_rng(H * dist)
    mskL = rng.choice([0, 0, 0, 0, 0, 0, 0, 1], (H, W)).astype(np.int16) if with_left_mask else None
    grids = None
    if with_grids:
        grids = (rng.integers(dmin, dmin + 3, (H, W)).astype(np.float64), rng.integers(dmax - 3, dmax + 1, (H, W)).astype(np.float64))
    masks = (mskL, None, 0, 1) if with_left_mask else None
    eng.set_profiling(True)
    eng.reset_stage_times()
    cv = gpu_cv(eng, "census", L, R, dmin, dmax, 1, win, masks=masks, grids=grids)
    eng.cbca(cv, off, 30.0, dist)
    got = cv.to_host()
    launches = {k: eng.stage_time(k)[1] for k in ("census_cost", "cbca_h", "cbca_v")}
    eng.set_profiling(False)
    eng.set_masks(None, None)
    eng.set_disparity_grids(None, None)
    assert launches == {"census_cost": 0, "cbca_h": 0, "cbca_v": 1}
    exp = cpu_cv(oracle, "census", L, R, dmin, dmax, 1, win, masks=masks, grids=grids)

    def arms(im, msk):
        m = im.copy()
        if msk is not None:
            m[msk != 0] = np.nan
        m = np.nan_to_num(oracle.median3(m), nan=np.inf)
        m = m[off:m.shape[0] - off, off:m.shape[1] - off]
        return oracle.cross_support(np.ascontiguousarray(m), dist, 30.0)

    oracle.cbca(exp, dmin, 1, off, arms(L, mskL), [arms(R, None)])
    np.testing.assert_array_equal(got, exp)


@pytest.mark.parametrize("H,W,dmin,dmax,method,win", [(23, 41, -9, 3, "census", 5), (30, 70, -4, 60, "sad", 3), (9, 300, -20, 0, "zncc", 5),
                                                       (64, 33, -2, 2, "census", 7)])
def test_sgm_with_penalty_maps(eng, oracle, H, W, dmin, dmax, method, win):
    """pmx_sgm_p2maps (P2 per pixel and path direction: the libSGM plugin's gradient-driven penalty methods) against the
    restatement, bit for bit, on random maps, in both modes (lazy census codes are materialised: float32 kernels only), "min" and
    "max" measures, D up to 65, the side-by-side and the one-after-the-other schedules; constant maps reproduce pmx_sgm."""
    L, R = pair(H, W, seed=H * W, integer=method != "zncc")
    rng = np.random.default_rng(H + W)
    is_max = method == "zncc"
    maps = rng.uniform(6.0, 90.0, (8, H, W)).astype(np.float32)
    ocv = cpu_cv(oracle, method, L, R, dmin, dmax, 1, win)
    invalid_cost = float(win * win + 1) if method == "census" else float(np.nanmax(np.abs(ocv)) + 1)
    exp = oracle.sgm_p2maps(ocv, 4.5, maps, is_max, invalid_cost, False)
    for sched in ("par", "seq"):
        eng.set_option("SGM_SCHED", sched)
        try:
            cv = gpu_cv(eng, method, L, R, dmin, dmax, 1, win)
            eng.sgm_p2maps(cv, 4.5, maps, is_max, invalid_cost, False)
            np.testing.assert_array_equal(cv.to_host(), exp)
            cv.free()
        finally:
            eng.set_option("SGM_SCHED", None)
    cv = gpu_cv(eng, method, L, R, dmin, dmax, 1, win)
    eng.sgm_p2maps(cv, 4.5, np.full((8, H, W), 31.0, np.float32), is_max, invalid_cost, True)
    np.testing.assert_array_equal(cv.to_host(), oracle.sgm(ocv, 4.5, 31.0, is_max, invalid_cost, True))
    cv.free()


@pytest.mark.parametrize("method", ["zncc", "sad", "census"])
@pytest.mark.parametrize("sp", [1, 2, 4])
def test_image_exactly_one_window_wide(eng, oracle, method, sp):
    """W == window_size: one column of valid cells at the integer disparities, none at the sub-pixel phases (their shifted
    right images are one column narrower).  pmx_zncc used to answer all-NaN here (found by tools/fuzz_more.py, seed 1397)."""
    win, H, W, dmin, dmax = 7, 22, 7, -5, 6
    L, R = pair(H, W, seed=11 + sp, integer=False)
    D = (dmax - dmin) * sp + 1
    cv = gpu_cv(eng, method, L, R, dmin, dmax, sp, win, masked=False)
    got = cv.to_host()
    exp = cpu_cv(oracle, method, L, R, dmin, dmax, sp, win)
    np.testing.assert_array_equal(np.isnan(got), np.isnan(exp))
    assert np.isfinite(exp).sum() == (H - win + 1)  # one column, disparity 0 only
    if method == "zncc":
        np.testing.assert_allclose(got, exp, rtol=0, atol=1e-5, equal_nan=True)
    else:
        np.testing.assert_array_equal(got, exp)
    assert got.shape == (H, W, D)
    cv.free()


@pytest.mark.parametrize("sp", [1, 2])
def test_cbca_on_one_cropped_column(eng, oracle, sp):
    """Image exactly one window wide: the volume CBCA sees has ONE column, whose rows still aggregate along the vertical arms
    (pmx_cbca used to return such a volume untouched; tools/fuzz_more.py, seeds 3995 and 4456)."""
    win, H, W, dmin, dmax, dist = 9, 18, 9, -3, 4, 3
    off = win // 2
    L, R = pair(H, W, seed=5 + sp, integer=True)
    cv = gpu_cv(eng, "sad", L, R, dmin, dmax, sp, win, masked=False)
    before = cv.to_host()
    eng.cbca(cv, off, 30.0, dist)
    got = cv.to_host()
    exp = cpu_cv(oracle, "sad", L, R, dmin, dmax, sp, win)

    def arms(im):
        m = np.nan_to_num(oracle.median3(im), nan=np.inf)[off:-off, off:-off]
        return oracle.cross_support(np.ascontiguousarray(m), dist, 30.0)

    oracle.cbca(exp, dmin, sp, off, arms(L), [arms(im) for im in oracle.shift_right(R, sp)])
    np.testing.assert_array_equal(got, exp)
    assert not np.array_equal(got, before, equal_nan=True)
    cv.free()


def test_validity_put_together_on_the_device(eng):
    """pmx_cv_mark_missing / pmx_compose_validity against the host functions they replace (criteria.py:291-353): the snapshot of
    the all-NaN pixels is the volume's state WHEN IT WAS TAKEN, a line base is broadcast over the rows, a full base is taken as
    it is, the frame overrides everything."""
    from pandora_amd import criteria

    rng = np.random.default_rng(77)
    H, W, D = 37, 53, 9
    L, R = pair(H, W, 5)
    eng.set_images(L, R, 1)
    cv = eng.alloc_cv(D, -4)
    vol = rng.random((H, W, D)).astype(np.float32)
    gone = rng.random((H, W)) < 0.2
    vol[gone] = np.nan
    vol[rng.random((H, W, D)) < 0.3] = np.nan
    cv.from_host(vol)
    missing = np.isnan(vol).all(axis=2)
    eng.mark_missing(cv)
    np.testing.assert_array_equal(eng.nan_pixels(cv), missing)
    vol2 = vol.copy()
    vol2[:5] = np.nan
    cv.from_host(vol2)  # the volume moves on, the snapshot does not
    np.testing.assert_array_equal(eng.get_missing(cv), missing)
    assert eng.nan_pixels(cv)[:5].all()
    for border in (0, 1, 3, 19, 40):
        for full in (False, True):
            base = rng.integers(0, 1 << 12, (H, W) if full else (W,)).astype(np.int64)
            for use_missing in (False, True):
                eng.compose_validity(base, cv if use_missing else None, border)
                want = np.empty((H, W), np.int64)
                want[:] = base
                if use_missing:
                    criteria._or_missing(want, missing)
                if border:
                    criteria._frame(want, border)
                got = eng.get_disparity()[1]
                np.testing.assert_array_equal(got, want, err_msg=f"border {border} full {full} missing {use_missing}")
    with pytest.raises(Exception, match="mark_missing"):
        eng.compose_validity(np.zeros(W, np.int64), eng.alloc_cv(D, 0), 0)
    with pytest.raises(ValueError):
        eng.compose_validity(np.zeros(W + 1, np.int64), None, 0)


def test_result_maps_survive_on_the_device_until_read(eng):
    """engine.DeviceMapArray: maps nobody has read move into a device-side snapshot (pmx_map_snapshot) when the next WTA, the next
    refinement or the next PAIR overwrites the engine's maps; reading them later gives the values of their own time."""
    from pandora_amd.engine import DeviceMapArray

    L, R = pair(40, 64, 9)
    want = {}
    lazies = {}
    for tag, (a, b, shape) in {"first": (L, R, (40, 64)), "second": (R[:30, :50].copy(), L[:30, :50].copy(), (30, 50))}.items():
        eng.set_images(a, b, 1)
        cv = eng.alloc_cv(7, -3)
        eng.census(cv, 5)
        eng.set_validity(None)
        eng.wta(cv, False, -9999.0)
        lazies[tag] = (DeviceMapArray(eng, "disp"), DeviceMapArray(eng, "validity"))
        assert lazies[tag][0].on_device() and lazies[tag][0].shape == shape
        eng.refine(cv, "vfit", False)  # overwrites disp / validity: the two lazies above are not superseded -> snapshots
        assert not lazies[tag][0].on_device() and lazies[tag][0]._snap is not None and lazies[tag][0]._host is None
        lazies[tag + " refined"] = (DeviceMapArray(eng, "disp"), DeviceMapArray(eng, "itp"))
        eng.set_validity(None)  # (the refinement left its flags in the engine's mask)
        eng.wta(cv, False, -9999.0)
        want[tag] = [x.copy() for x in eng.get_disparity()[:2]]
        eng.refine(cv, "vfit", False)
        want[tag + " refined"] = [eng.get_disparity(want_itp=True)[i].copy() for i in (0, 2)]
    for tag, (x, y) in lazies.items():  # read long after: both pairs are gone from the engine's maps
        np.testing.assert_array_equal(x.data, want[tag][0], err_msg=tag)
        np.testing.assert_array_equal(y.data, want[tag][1], err_msg=tag)
        assert x._snap is None and x.data is x.data


def test_steps_on_snapshots_equal_the_host_pointer_forms(eng):
    """pmx_median_filter_maps / pmx_cross_checking_maps / pmx_validity_frame_map / pmx_maps_restore work on device-side snapshots
    and give what the host-pointer entry points (checked against the oracle elsewhere in this file) give on the same maps."""
    from pandora_amd.engine import DeviceMapArray

    rng = np.random.default_rng(31)
    H, W = 45, 70
    L, R = pair(H, W, 12)
    eng.set_images(L, R, 1)
    dl = rng.integers(-12, 3, (H, W)).astype(np.float32) + rng.random((H, W)).astype(np.float32)
    dr = -dl + rng.integers(-2, 3, (H, W)).astype(np.float32)
    dl[rng.random((H, W)) < 0.05] = np.nan
    val = (rng.random((H, W)) < 0.15).astype(np.int64) * 2 + (rng.random((H, W)) < 0.1).astype(np.int64) * 4
    # host maps -> engine maps -> snapshots
    eng.set_disparity(dl, val)
    left = (DeviceMapArray(eng, "disp"), DeviceMapArray(eng, "validity"))
    snap_l, snap_v = left[0].device_snapshot(), left[1].device_snapshot()
    eng.set_disparity(dr, np.zeros((H, W), np.int64))
    snap_r = DeviceMapArray(eng, "disp")
    for size in (3, 5):
        out = eng.median_filter_maps(snap_l, snap_v, size)
        got = DeviceMapArray.from_snapshot(eng, "disp", out).data
        np.testing.assert_array_equal(got, eng.median_filter_disparity(dl.copy(), val, size))
    with pytest.raises(Exception, match="in place|snapshot"):
        from pandora_amd import _lib
        from pandora_amd.engine import check
        check(_lib.lib().pmx_median_filter_maps(eng.ctx, snap_l, snap_v, 3, snap_l), "pmx_median_filter_maps")
    want_val, want_conf = eng.cross_checking(dl, val, dr, -12, 3, 1.0)
    conf = eng.cross_checking_maps(snap_l, snap_v, snap_r.device_snapshot(), -12, 3, 1.0, border=2)
    from pandora_amd import criteria
    criteria._frame(want_val, 2)
    np.testing.assert_array_equal(left[1].data, want_val)  # updated in place in the left mask's own snapshot
    np.testing.assert_array_equal(DeviceMapArray.from_snapshot(eng, "conf", conf).data, want_conf)
    # restore: the engine's maps are the snapshots' values again
    a, b = DeviceMapArray(eng, "disp"), DeviceMapArray(eng, "validity")  # (current maps: the right side's)
    eng.maps_restore(snap_l, a.device_snapshot() and b.device_snapshot())
    np.testing.assert_array_equal(eng.get_disparity()[0], dl)
    np.testing.assert_array_equal(eng.get_disparity()[1], np.zeros((H, W), np.int64))
    np.testing.assert_array_equal(a.data, dr)


@pytest.mark.parametrize("subpix", [1, 2, 4])
def test_swapped_pair_equals_the_pair_uploaded_the_other_way_round(eng, subpix):
    """pmx_swap_images (the right-side volume of a cross-checked run: same two images, other order) against pmx_set_images(right,
    left): images, masks and the rebuilt sub-pixel right images give the same volumes, masked the same way."""
    rng = np.random.default_rng(3)
    H, W = 33, 61
    L, R = pair(H, W, 21, integer=False)
    mL = (rng.random((H, W)) < 0.02).astype(np.int16) * rng.integers(1, 3, (H, W)).astype(np.int16)
    mR = (rng.random((H, W)) < 0.03).astype(np.int16) * rng.integers(1, 3, (H, W)).astype(np.int16)
    D = 9 * subpix + 1
    vols = []
    for way in ("swap", "upload"):
        if way == "swap":
            eng.set_images(L, R, subpix)
            eng.set_masks(mL, mR, 0, 1)
            eng.swap_images()
        else:
            eng.set_images(R, L, subpix)
            eng.set_masks(mR, mL, 0, 1)
        got = []
        for method in ("census", "zncc", "sad"):
            cv = eng.alloc_cv(D, -4)
            {"census": lambda: eng.census(cv, 5), "zncc": lambda: eng.zncc(cv, 3), "sad": lambda: eng.sad_ssd(cv, 3, False)}[method]()
            eng.cv_masked(cv, 5 if method == "census" else 3)
            got.append(cv.to_host())
            cv.free()
        vols.append(got)
    for a, b in zip(*vols):
        np.testing.assert_array_equal(a, b)
    assert all(np.isfinite(v).any() and np.isnan(v).any() for v in vols[0])


def test_percentiles_from_device_order_statistics(eng):
    """pmx_order_statistics (radix selection) == np.partition's elements, and ambiguity.percentiles == np.percentile bit for bit:
    random maps, heavy ties, negative values, infinities, a map with NaNs (numpy's answer is NaN), percentiles at both ends."""
    from pandora_amd.cost_volume_confidence.ambiguity import percentiles

    rng = np.random.default_rng(17)
    n = 300 * 401
    maps = {"normal": rng.normal(0, 50, n).astype(np.float32),
            "ties": rng.integers(0, 70, n).astype(np.float32),
            "ambiguity-like": (rng.integers(0, 71, n) + rng.integers(0, 2, n) * 0.5).astype(np.float32),
            "with infinities": np.where(rng.random(n) < 0.01, np.inf, rng.normal(0, 1, n)).astype(np.float32) * np.where(rng.random(n) < 0.5, -1, 1).astype(np.float32),
            "tiny and huge": (rng.normal(0, 1, n) * 10.0 ** rng.integers(-30, 30, n)).astype(np.float32)}
    for name, a in maps.items():
        ranks = sorted({0, 1, n // 3, n // 2, n - 2, n - 1, int(0.01 * (n - 1)), int(0.99 * (n - 1))})
        want = np.sort(a)[ranks]
        np.testing.assert_array_equal(eng.order_statistics(a, ranks), want, err_msg=name)
        for qs in ((1.0, 99.0), (0.0, 100.0), (37.123, 50.0), (2.5, 97.5)):
            with np.errstate(invalid="ignore"):  # (inf - inf inside numpy's interpolation between two infinities)
                got = percentiles(a.reshape(300, 401), qs)
                want_q = [np.percentile(a, q) for q in qs]
            for g, q, w in zip(got, qs, want_q):
                assert g.dtype == w.dtype and (g == w or (np.isnan(g) and np.isnan(w))), (name, q, g, w)
    b = maps["normal"].copy()
    b[[5, 777]] = [np.nan, -np.nan]
    assert np.isnan(eng.order_statistics(b, [n - 1, n - 2])).all() and not np.isnan(eng.order_statistics(b, [n - 3])).any()
    assert all(np.isnan(p) for p in percentiles(b, (1.0, 99.0)))
