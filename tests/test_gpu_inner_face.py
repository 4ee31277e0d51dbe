"""GPU: pandora_amd.inner_cpp - the pybind11 face with the reference's own native-function signatures (matching_cost_cpp /
aggregation_cpp / refinement_cpp) - against the reference's COMPILED modules (oracle/_ref, built from /root/reference by
oracle/Makefile) on the same numpy arguments, call for call, bit for bit.  Where oracle/_ref cannot be built (a box without the
reference sources) the C restatement (oracle/liboracle.so, itself diffed against those modules) stands in."""
import numpy as np
import pytest

from tests.test_gpu_parity import pair

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def face():
    from pandora_amd import inner_cpp

    return inner_cpp


def _ref(name):
    from oracle import ref

    return ref.load(name)


@pytest.mark.parametrize("win,subpix,dmin,D", [(5, 1, -7, 12), (3, 2, -3, 11), (7, 4, 0, 17), (11, 1, -20, 9)])
def test_compute_matching_costs_like_the_reference_module(face, oracle, win, subpix, dmin, D):
    H, W = 23, 41
    L, R = pair(H, W, seed=win + subpix)
    rights = oracle.shift_right(R, subpix)  # what img_tools.shift_right_img hands to census.py:113-133
    disps = (dmin + np.arange(D) / subpix).astype(np.float32)
    cv = np.full((H, W, D), np.nan, np.float32)
    got = face.compute_matching_costs(L, [np.ascontiguousarray(r) for r in rights], cv, disps, win, win)
    assert got is cv  # written into and returned (census.cpp:106,179)
    mc = _ref("matching_cost_cpp")
    if mc is not None:
        exp = mc.compute_matching_costs(L, [np.ascontiguousarray(r) for r in rights], np.full((H, W, D), np.nan, np.float32), disps, win, win)
    else:
        exp = oracle.census_cost(L, R, D, dmin, subpix, win)
    np.testing.assert_array_equal(got, exp)
    # forcecast semantics: a float64, strided left image and an integer disparity vector are silently converted
    got2 = face.compute_matching_costs(np.asfortranarray(L.astype(np.float64)), [np.ascontiguousarray(r) for r in rights],
                                       np.full((H, W, D), np.nan, np.float32), disps.astype(np.float64), win, win)
    np.testing.assert_array_equal(got2, exp)
    # cells the kernel cannot compute keep what the caller passed
    seven = face.compute_matching_costs(L, [np.ascontiguousarray(r) for r in rights], np.full((H, W, D), 7.0, np.float32), disps, win, win)
    np.testing.assert_array_equal(seven, np.where(np.isnan(exp), np.float32(7.0), exp))


def test_cross_support_and_cbca_like_the_reference_module(face, oracle):
    H, W, dmin, D = 31, 47, -6, 9
    L, R = pair(H, W, seed=5, integer=False)
    L[3, 4] = np.inf
    R[10:12, 20:23] = np.inf
    agg = _ref("aggregation_cpp")
    armsL, armsR = face.cross_support(L, 5, 30.0), face.cross_support(R, 5, 30.0)
    assert armsL.dtype == np.int16 and armsL.shape == (H, W, 4)
    if agg is not None:
        np.testing.assert_array_equal(armsL, agg.cross_support(L, 5, 30.0))
        np.testing.assert_array_equal(armsR, agg.cross_support(R, 5, 30.0))
    else:
        np.testing.assert_array_equal(armsL, oracle.cross_support(L, 5, 30.0))
    cv = np.random.default_rng(1).random((H, W, D)).astype(np.float32) * 40
    cv[np.random.default_rng(2).random(cv.shape) < 0.1] = np.nan
    range_col = np.arange(W)
    for k in range(D):  # the loop of cbca.py:152-171
        right = range_col + (dmin + k)
        valid = np.where((right >= 0) & (right < W))
        e, n = face.cbca(cv[:, :, k], armsL, armsR, range_col[valid], right[valid].astype(int))
        assert e.dtype == np.float32 and n.dtype == np.float32 and e.shape == (H, W)
        if agg is not None:
            ee, en = agg.cbca(cv[:, :, k], armsL, armsR, range_col[valid], right[valid].astype(int))
            np.testing.assert_array_equal(e, ee)
            np.testing.assert_array_equal(n, en)
    # the whole python recipe of cbca.py:127-177 around the face equals the fused device aggregation's restatement
    exp = cv.copy()
    oracle.cbca(exp, dmin, 1, 0, armsL, [armsR])
    out = np.empty_like(cv)
    for k in range(D):
        right = range_col + (dmin + k)
        valid = np.where((right >= 0) & (right < W))
        e, n = face.cbca(cv[:, :, k], armsL, armsR, range_col[valid], right[valid].astype(int))
        out[:, :, k] = (cv[:, :, k] * 0 + e) / (n + 1)
    np.testing.assert_array_equal(out, exp)


@pytest.mark.parametrize("method", ["vfit", "quadratic"])
@pytest.mark.parametrize("measure,subpix", [("min", 1), ("max", 2)])
def test_loop_refinement_like_the_reference_module(face, oracle, method, measure, subpix):
    H, W, dmin, dmax = 19, 27, -4, 3
    D = (dmax - dmin) * subpix + 1
    rng = np.random.default_rng(8)
    cv = rng.integers(0, 9, (H, W, D)).astype(np.float32)
    cv[rng.random(cv.shape) < 0.1] = np.nan
    disp, mask = oracle.wta(cv, dmin, subpix, measure == "max", -9999.0)
    mask = mask.astype(np.int64)
    fn = getattr(face, f"{method}_refinement_method")

    def callback(cost, d, meas):  # the shape of vfit.py:43-45's staticmethod
        return fn(cost, d, meas, 8)

    itp, d2, m2 = face.loop_refinement(cv, disp.copy(), mask.copy(), float(dmin), float(dmax), subpix, measure, callback, 963, 8)
    rf = _ref("refinement_cpp")
    if rf is not None:
        rfn = getattr(rf, f"{method}_refinement_method")
        eitp, ed, em = rf.loop_refinement(cv, disp.copy(), mask.copy(), float(dmin), float(dmax), subpix, measure,
                                          lambda cost, d, meas: rfn(cost, d, meas, 8), 963, 8)
        for probe in ([3, 1, 2], [1, 1, 1], [np.nan, 1, 2], [2, 5, 3]):
            c = np.array(probe, np.float32)
            assert fn(c, 0.0, measure, 8) == pytest.approx(rfn(c, 0.0, measure, 8), nan_ok=True, abs=0)
    else:
        eitp, ed, em = oracle.refine(cv, disp.copy(), mask.copy(), dmin, dmax, subpix, measure == "max", method)
    np.testing.assert_array_equal(itp, eitp)
    np.testing.assert_array_equal(d2, ed)
    np.testing.assert_array_equal(m2, em)
    with pytest.raises(Exception):
        face.loop_refinement(cv, disp, mask, float(dmin), float(dmax), subpix, measure, lambda c, d, m: (0.0, 0.0, 0), 963, 8)


def _approx_restated(cv, disp, mask, dmin, dmax, subpix, measure, fn):
    """refinement/cpp/src/refinement.cpp:103-182 in python loops (the unchecked indices as flat offsets of the contiguous volume)"""
    H, W, D = cv.shape
    flat = cv.reshape(-1)
    itp = np.empty((H, W), np.float32)
    disp, mask = disp.copy(), mask.copy()
    for r in range(H):
        for c in range(W):
            if mask[r, c] & 963:
                itp[r, c] = np.nan
                continue
            raw = np.float32(disp[r, c])
            dsp = int((-float(raw) - dmin) * subpix)
            diag = int(np.float32(c) + raw)
            at = (r * W + diag) * D + dsp
            c1 = flat[at]
            if np.isnan(c1):
                itp[r, c] = c1
                continue
            if raw == dmin or raw == dmax or diag == 0 or diag == W - 1:
                itp[r, c] = c1
                mask[r, c] += 8
                continue
            x, y, v = fn(np.array([flat[at - D + subpix], c1, flat[at + D - subpix]], np.float32), raw, measure, 8)
            disp[r, c] = raw + np.float32(x) / np.float32(subpix)
            itp[r, c] = y
            mask[r, c] += v
    return itp, disp, mask


@pytest.mark.parametrize("method", ["vfit", "quadratic"])
@pytest.mark.parametrize("measure,subpix,dmin,dmax", [("min", 1, -4, 3), ("max", 2, -2, 5), ("min", 4, 0, 3), ("min", 1, -6, -1)])
def test_loop_approximate_refinement_like_the_reference_module(face, method, measure, subpix, dmin, dmax):
    """refinement_cpp.loop_approximate_refinement (refinement_cpp.pyi:82-122): a right map (disparities in [-dmax, -dmin], every
    pixel's diagonal inside the image, some pixels invalid) refined on the left volume - against the reference's compiled module where
    it is built (oracle/_ref) and against the restatement above; asymmetric ranges reach the neighbouring pixel's run of the volume
    exactly as the reference's unchecked indices do."""
    H, W = 19, 27
    D = (dmax - dmin) * subpix + 1
    rng = np.random.default_rng(D + subpix)
    cv = rng.integers(0, 9, (H, W, D)).astype(np.float32)
    cv[rng.random(cv.shape) < 0.1] = np.nan
    disp = (-(dmin + rng.integers(0, D, (H, W)) / subpix)).astype(np.float32)  # right disparities, on the volume's grid
    mask = np.zeros((H, W), np.int64)
    diag = (np.arange(W, dtype=np.float32)[None, :] + disp).astype(np.int64)
    mask[(diag < 0) | (diag >= W)] = 1 << 1  # (the reference would read outside the volume there)
    mask[rng.random((H, W)) < 0.05] |= 1
    mask[rng.random((H, W)) < 0.05] |= 4  # (an information bit: refined all the same)
    fn = getattr(face, f"{method}_refinement_method")
    itp, d2, m2 = face.loop_approximate_refinement(cv, disp.copy(), mask.copy(), float(dmin), float(dmax), subpix, measure,
                                                   lambda cost, d, meas: fn(cost, d, meas, 8), 963, 8)
    eitp, ed, em = _approx_restated(cv, disp, mask, dmin, dmax, subpix, measure, fn)
    np.testing.assert_array_equal(itp, eitp)
    np.testing.assert_array_equal(d2, ed)
    np.testing.assert_array_equal(m2, em)
    rf = _ref("refinement_cpp")
    if rf is not None:
        rfn = getattr(rf, f"{method}_refinement_method")
        ritp, rd, rm = rf.loop_approximate_refinement(cv, disp.copy(), mask.copy(), float(dmin), float(dmax), subpix, measure,
                                                      lambda cost, d, meas: rfn(cost, d, meas, 8), 963, 8)
        np.testing.assert_array_equal(itp, ritp)
        np.testing.assert_array_equal(d2, rd)
        np.testing.assert_array_equal(m2, rm)


NAN = np.nan


def test_reverse_cost_volume_like_the_reference_module(face):
    # the two known answers of tests/test_cpp/test_matching_cost/test_matching_cost.cpp:80-100 (d = [1, 4] -> right min -4) and
    # :181-201 (d = [-2, 2] -> right min -2), transcribed
    left = np.array([[[12, 13, 14, 15], [23, 24, 25, 26], [34, 35, 36, NAN], [45, 46, NAN, NAN], [56, NAN, NAN, NAN],
                      [NAN, NAN, NAN, NAN]]], np.float32)
    right = np.array([[[NAN, NAN, NAN, NAN], [NAN, NAN, NAN, 12], [NAN, NAN, 13, 23], [NAN, 14, 24, 34], [15, 25, 35, 45],
                       [26, 36, 46, 56]]], np.float32)
    got = face.reverse_cost_volume(left, -4)
    assert got.dtype == np.float32 and got.shape == left.shape
    np.testing.assert_array_equal(got, right)
    left = np.array([[[NAN, NAN, 11, 12, 13], [NAN, 21, 22, 23, 24], [31, 32, 33, 34, 35], [42, 43, 44, 45, 46], [53, 54, 55, 56, NAN],
                      [64, 65, 66, NAN, NAN]]], np.float32)
    right = np.array([[[NAN, NAN, 11, 21, 31], [NAN, 12, 22, 32, 42], [13, 23, 33, 43, 53], [24, 34, 44, 54, 64], [35, 45, 55, 65, NAN],
                       [46, 56, 66, NAN, NAN]]], np.float32)
    np.testing.assert_array_equal(face.reverse_cost_volume(left, -2), right)
    # random volumes against the reference's compiled module, forcecast included (float64, Fortran order)
    mc = _ref("matching_cost_cpp")
    rng = np.random.default_rng(12)
    for H, W, D, dmin in ((7, 33, 9, -5), (11, 20, 14, 0), (5, 9, 6, -12), (3, 40, 40, -20)):
        cv = rng.random((H, W, D)).astype(np.float32)
        cv[rng.random(cv.shape) < 0.15] = np.nan
        got = face.reverse_cost_volume(cv, dmin)
        if mc is not None:
            np.testing.assert_array_equal(got, mc.reverse_cost_volume(cv, dmin))
            np.testing.assert_array_equal(face.reverse_cost_volume(np.asfortranarray(cv.astype(np.float64)), dmin), got)
        else:  # matching_cost.cpp:43-52 restated
            exp = np.full_like(cv, np.nan)
            for j in range(W):
                for d in range(D):
                    col = j + d + dmin
                    if 0 <= col < W:
                        exp[:, j, d] = cv[:, col, D - 1 - d]
            np.testing.assert_array_equal(got, exp)


def test_reverse_disp_range_like_the_reference_module(face):
    mc = _ref("matching_cost_cpp")
    rng = np.random.default_rng(13)
    for H, W, lo, hi in ((9, 31, -6, 4), (4, 50, 0, 12), (6, 17, -30, -3), (5, 8, -2, 2)):
        a = rng.integers(lo, hi + 1, (H, W))
        b = a + rng.integers(0, 5, (H, W))
        lmin, lmax = a.astype(np.float32), b.astype(np.float32)
        lmin[rng.random((H, W)) < 0.1] = np.nan
        lmax[rng.random((H, W)) < 0.1] = np.nan
        rmin, rmax = face.reverse_disp_range(lmin, lmax)
        assert rmin.dtype == np.float32 and rmin.shape == (H, W)
        if mc is not None:
            emin, emax = mc.reverse_disp_range(lmin, lmax)
        else:  # matching_cost.cpp:84-118 restated
            emin, emax = np.full((H, W), np.inf, np.float32), np.full((H, W), -np.inf, np.float32)
            for r in range(H):
                for c in range(W):
                    if np.isnan(lmin[r, c]) or np.isnan(lmax[r, c]):
                        continue
                    for d in range(int(lmin[r, c]), int(lmax[r, c]) + 1):
                        if 0 <= c + d < W:
                            emin[r, c + d] = min(emin[r, c + d], -d)
                            emax[r, c + d] = max(emax[r, c + d], -d)
            none = np.isinf(emin)
            emin[none] = np.nan
            emax[none] = np.nan
        np.testing.assert_array_equal(rmin, emin)
        np.testing.assert_array_equal(rmax, emax)
    # every range NaN: nothing reaches any column
    n = np.full((3, 5), np.nan, np.float32)
    rmin, rmax = face.reverse_disp_range(n, n)
    assert np.isnan(rmin).all() and np.isnan(rmax).all()


def test_cbca_refuses_arms_that_do_not_fit_the_image(face):
    """arms index the kernels' prefix sums (aggregation.cpp:99-117, :181-213, unchecked there): arms that leave the image are an
    error of the caller, reported before anything is launched"""
    H, W = 6, 8
    cv = np.zeros((H, W), np.float32)
    arms = np.zeros((H, W, 4), np.int16)
    cols = np.arange(W)
    face.cbca(cv, arms, arms, cols, cols)
    for k, (r, c) in enumerate(((2, 1), (2, 6), (1, 3), (4, 3))):  # left, right, up, down arms longer than the image
        bad = arms.copy()
        bad[r, c, k] = 3
        with pytest.raises(RuntimeError, match="do not fit"):
            face.cbca(cv, bad, bad, cols, cols)
    neg = arms.copy()
    neg[3, 3, 1] = -2
    with pytest.raises(RuntimeError, match="do not fit"):
        face.cbca(cv, neg, neg, cols, cols)


@pytest.mark.parametrize("method", ["vfit", "quadratic"])
@pytest.mark.parametrize("measure,dmin,dmax", [("min", -4, 3), ("max", -2, 5), ("min", -6, -1)])
def test_plugin_approximate_subpixel_refinement(face, method, measure, dmin, dmax):
    """AbstractRefinement.approximate_subpixel_refinement (refinement/refinement.py:124-158) on Datasets, as a plugin user calls it:
    the right map given as host arrays and as device-resident maps (which go back device to device) - both equal the restatement
    above and, where it is built, the reference's compiled loop_approximate_refinement."""
    from pandora_amd import refinement
    from pandora_amd.dataset import Dataset, DeviceVolumeArray
    from pandora_amd.engine import DeviceMapArray, Engine

    H, W = 19, 27
    D = dmax - dmin + 1
    rng = np.random.default_rng(D)
    cvh = rng.integers(0, 9, (H, W, D)).astype(np.float32)
    cvh[rng.random(cvh.shape) < 0.1] = np.nan
    disp = (-(dmin + rng.integers(0, D, (H, W)))).astype(np.float32)
    mask = np.zeros((H, W), np.int64)
    diag = (np.arange(W, dtype=np.float32)[None, :] + disp).astype(np.int64)
    mask[(diag < 0) | (diag >= W)] = 1 << 1
    mask[rng.random((H, W)) < 0.05] |= 1
    fn = getattr(face, f"{method}_refinement_method")
    eitp, ed, em = _approx_restated(cvh, disp, mask, dmin, dmax, 1, measure, fn)
    rf = _ref("refinement_cpp")
    if rf is not None:
        rfn = getattr(rf, f"{method}_refinement_method")
        ritp, rd, rm = rf.loop_approximate_refinement(cvh, disp.copy(), mask.copy(), float(dmin), float(dmax), 1, measure,
                                                      lambda cost, d, meas: rfn(cost, d, meas, 8), 963, 8)
        np.testing.assert_array_equal(eitp, ritp)
        np.testing.assert_array_equal(ed, rd)
        np.testing.assert_array_equal(em, rm)
    eng = Engine(0)
    try:
        z = np.zeros((H, W), np.float32)
        eng.set_images(z, z, 1)
        dcv = eng.alloc_cv(D, dmin)
        dcv.from_host(cvh)
        coords = {"row": np.arange(H), "col": np.arange(W)}
        cv = Dataset({"cost_volume": DeviceVolumeArray(dcv, coords)}, coords=coords,
                     attrs={"type_measure": measure, "subpixel": 1})
        plugin = refinement.AbstractRefinement(refinement_method=method)
        for resident in (False, True):
            if resident:  # the maps as a step before would have left them: in the engine's buffers, never read
                eng.set_disparity(disp, mask)
                right = Dataset({"disparity_map": DeviceMapArray(eng, "disp", coords=coords),
                                 "validity_mask": DeviceMapArray(eng, "validity", coords=coords)}, coords=coords)
            else:
                right = Dataset({"disparity_map": (("row", "col"), disp.copy()), "validity_mask": (("row", "col"), mask.copy())},
                                coords=coords)
            out = plugin.approximate_subpixel_refinement(cv, right)
            assert out is right and right.attrs["refinement"] == method
            np.testing.assert_array_equal(np.asarray(right["interpolated_coeff"].data), eitp)
            np.testing.assert_array_equal(np.asarray(right["disparity_map"].data), ed)
            np.testing.assert_array_equal(np.asarray(right["validity_mask"].data), em)
        dcv.free()
    finally:
        eng.close()
