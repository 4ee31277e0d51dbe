"""Diffs the C restatement (oracle/oracle.c) against the reference's OWN compiled C++ (oracle/_ref,
built from /root/reference by oracle/Makefile) on random inputs.  CPU only; skipped if _ref is absent."""
import numpy as np
import pytest

from oracle import ref

mc = ref.load("matching_cost_cpp")
ag = ref.load("aggregation_cpp")
rf = ref.load("refinement_cpp")
needs_ref = pytest.mark.skipif(mc is None or ag is None or rf is None, reason="oracle/_ref not built")


@needs_ref
@pytest.mark.parametrize("H,W,dmin,dmax,sp,win", [(20, 30, -5, 3, 1, 5), (15, 25, -3, 4, 2, 3), (17, 23, -2, 2, 4, 7),
                                                  (30, 40, 0, 8, 1, 13), (12, 50, -20, -4, 1, 9), (16, 33, 2, 9, 2, 11)])
def test_census_matches_reference(oracle, H, W, dmin, dmax, sp, win):
    rng = np.random.default_rng(H * W + win)
    L = rng.integers(0, 40, (H, W)).astype(np.float32)
    R = rng.integers(0, 40, (H, W)).astype(np.float32)
    D = (dmax - dmin) * sp + 1
    cv = np.full((H, W, D), np.nan, np.float32)
    out = mc.compute_matching_costs(L, oracle.shift_right(R, sp), cv, np.arange(D) / sp + dmin, win, win)
    np.testing.assert_array_equal(out, oracle.census_cost(L, R, D, dmin, sp, win))


@needs_ref
def test_shift_right_matches_scipy_zoom(oracle):
    from scipy.ndimage import zoom

    rng = np.random.default_rng(3)
    R = (rng.random((9, 14)) * 255).astype(np.float32)
    nx = R.shape[1]
    for sp in (2, 4):
        for k in range(1, sp):
            z = zoom(R, (1, (nx * sp - (sp - 1)) / float(nx)), order=1)[:, k::sp]  # img_tools.py:742
            np.testing.assert_array_equal(z, oracle.shift_right(R, sp)[k])


@needs_ref
def test_cross_support_matches_reference(oracle):
    rng = np.random.default_rng(1)
    for _ in range(8):
        H, W = rng.integers(5, 30, 2)
        img = (rng.random((H, W)) * 60).astype(np.float32)
        img[rng.random((H, W)) < 0.1] = np.inf
        la, it = int(rng.integers(1, 8)), float(rng.random() * 30 + 1)
        np.testing.assert_array_equal(ag.cross_support(img.copy(), la, it), oracle.cross_support(img, la, it))


def _ref_cbca_volume(cv, d0, sp, off, cl, crs):
    """numpy glue mirroring aggregation/cbca.py:127-177 around the reference's aggregation_cpp.cbca"""
    H, W, D = cv.shape
    cvd = cv[off:H - off, off:W - off] if off > 0 else cv
    nrow, ncol, nd = cvd.shape
    agg = np.zeros((nd, ncol, nrow), np.float32)
    agg += np.swapaxes(cvd, 0, 2)
    agg *= 0
    disps = d0 + np.arange(D) / sp
    rc = np.arange(0, ncol)
    for k in range(nd):
        ir = int((disps[k] % 1) * sp)
        rcr = rc + disps[k]
        vi = np.where((rcr >= 0) & (rcr < crs[ir].shape[1]))
        s4, n4 = ag.cbca(cvd[:, :, k], cl, crs[ir], rc[vi], rcr[vi].astype(int))
        n4 += 1
        agg[k] += np.swapaxes(s4, 0, 1)
        agg[k] /= np.swapaxes(n4, 0, 1)
    out = cv.copy()
    if off > 0:
        out[off:H - off, off:W - off] = np.swapaxes(agg, 0, 2)
    else:
        out = np.swapaxes(agg, 0, 2).copy()
    return out


@needs_ref
@pytest.mark.parametrize("H,W,dmin,dmax,sp,win", [(20, 30, -5, 3, 1, 5), (15, 25, -3, 4, 2, 3), (17, 23, -2, 2, 4, 1)])
def test_cbca_matches_reference(oracle, H, W, dmin, dmax, sp, win):
    rng = np.random.default_rng(7)
    L = (rng.random((H, W)) * 100).astype(np.float32)
    R = (rng.random((H, W)) * 100).astype(np.float32)
    D = (dmax - dmin) * sp + 1
    off = win // 2
    cv = oracle.sad_ssd(L, R, D, dmin, sp, win, False)
    cv[rng.random(cv.shape) < 0.05] = np.nan

    def arms(im):
        m = np.nan_to_num(oracle.median3(im), nan=np.inf)
        if off:
            m = m[off:-off, off:-off]
        return oracle.cross_support(np.ascontiguousarray(m), 5, 30.0)

    cl, crs = arms(L), [arms(i) for i in oracle.shift_right(R, sp)]
    np.testing.assert_array_equal(_ref_cbca_volume(cv, dmin, sp, off, cl, crs), oracle.cbca(cv.copy(), dmin, sp, off, cl, crs))


@needs_ref
@pytest.mark.parametrize("measure", ["min", "max"])
@pytest.mark.parametrize("method", ["vfit", "quadratic"])
def test_refinement_matches_reference(oracle, measure, method):
    rng = np.random.default_rng(11)
    H, W, D = 12, 14, 9
    cv = (rng.random((H, W, D)) * 10).astype(np.float32)
    cv[rng.random(cv.shape) < 0.1] = np.nan
    cv[0, 0, :] = np.nan
    cv[1, 1, :] = 3.0  # all-equal costs: quadratic hits 0/0
    disp, val = oracle.wta(cv, -4, 1, measure == "max", -9999.0)
    if method == "vfit":
        f = lambda c, d, m: rf.vfit_refinement_method(c, d, m, 8)  # noqa: E731
    else:
        f = lambda c, d, m: rf.quadratic_refinement_method(c, d, m, 8)  # noqa: E731
    itp_r, d_r, v_r = rf.loop_refinement(cv, disp.copy(), val.copy(), -4.0, 4.0, 1, measure, f, 0x3C3, 8)
    itp, d, v = oracle.refine(cv, disp, val, -4.0, 4.0, 1, measure == "max", method)
    np.testing.assert_array_equal(itp, itp_r)
    np.testing.assert_array_equal(d, d_r)
    np.testing.assert_array_equal(v, v_r)


@needs_ref
def test_reverse_cost_volume_matches_reference(oracle):
    rng = np.random.default_rng(5)
    cv = rng.random((6, 9, 5)).astype(np.float32)
    for md in (-2, 0, -4):
        np.testing.assert_array_equal(mc.reverse_cost_volume(cv, md), oracle.reverse_cost_volume(cv, md))


@needs_ref
def test_reverse_disp_range_matches_reference(oracle):
    rng = np.random.default_rng(6)
    for H, W in ((5, 17), (3, 40)):
        lo = rng.integers(-6, 3, (H, W)).astype(np.float32)
        hi = lo + rng.integers(0, 7, (H, W)).astype(np.float32)
        lo[0, 3] = np.nan
        hi[1, 5] = np.nan
        rmin_r, rmax_r = mc.reverse_disp_range(lo, hi)
        rmin, rmax = oracle.reverse_disp_range(lo, hi)
        np.testing.assert_array_equal(rmin, rmin_r)
        np.testing.assert_array_equal(rmax, rmax_r)
    # the reference's own vectors (tests/test_cpp/test_matching_cost.cpp:103-201) are covered by the compiled
    # module itself; a constant range gives -max / -min clipped at the borders
    rmin, rmax = oracle.reverse_disp_range(np.full((2, 6), -2, np.float32), np.full((2, 6), 1, np.float32))
    np.testing.assert_array_equal(rmin[0], [0, -1, -1, -1, -1, -1])
    np.testing.assert_array_equal(rmax[0], [2, 2, 2, 2, 1, 0])


cc = ref.load("cost_volume_confidence_cpp")


@pytest.mark.skipif(cc is None, reason="oracle/_ref not built")
@pytest.mark.parametrize("H,W,D,d0,sp", [(9, 14, 12, -5, 1), (6, 11, 17, -2, 2), (5, 8, 9, 0, 4)])
def test_ambiguity_matches_reference(oracle, H, W, D, d0, sp):
    """ambiguity.cpp:28-142: NaN costs inside / outside the per-pixel range, pixels without any cost, variable grids."""
    rng = np.random.default_rng(H * W + D)
    cv = (rng.random((H, W, D)) * 30).astype(np.float32)
    cv[rng.random((H, W, D)) < 0.15] = np.nan
    cv[1, 2, :] = np.nan
    disp_range = (d0 + np.arange(D) / sp).astype(np.float32)
    gmin = rng.integers(d0, d0 + 2, (H, W)).astype(np.int64)
    gmax = (gmin + rng.integers(1, max(2, (D - 1) // sp), (H, W))).astype(np.int64)
    etas = np.arange(0.0, 0.7, 0.01)
    grids = np.array([gmin, gmax], dtype=np.int64)
    exp = cc.compute_ambiguity_and_sampled_ambiguity(cv, etas, len(etas), grids, disp_range, False)[0]
    got = oracle.ambiguity(cv, etas, gmin, gmax, disp_range)
    np.testing.assert_array_equal(got, exp)
    assert got[1, 2] == len(etas) * D


def _confidence_case(H, W, D, d0, sp, seed, quantised):
    rng = np.random.default_rng(seed)
    if quantised:  # census-like integer costs: many exact ties with the extremum
        cv = rng.integers(0, 12, (H, W, D)).astype(np.float32)
    else:
        cv = (rng.random((H, W, D)) * 30 - 8).astype(np.float32)
    cv[rng.random((H, W, D)) < 0.15] = np.nan
    cv[1, 2, :] = np.nan
    cv[2, 3, 1:] = np.nan  # a single cost
    disp_range = (d0 + np.arange(D) / sp).astype(np.float32)
    gmin = rng.integers(d0, d0 + 2, (H, W)).astype(np.int64)
    gmax = (gmin + rng.integers(1, max(2, (D - 1) // sp), (H, W))).astype(np.int64)
    return cv, disp_range, gmin, gmax


@pytest.mark.skipif(cc is None, reason="oracle/_ref not built")
@pytest.mark.parametrize("quantised", [False, True])
@pytest.mark.parametrize("H,W,D,d0,sp", [(9, 14, 12, -5, 1), (6, 11, 17, -2, 2), (5, 8, 9, 0, 4), (7, 9, 33, -30, 1)])
def test_risk_matches_reference(oracle, H, W, D, d0, sp, quantised):
    """risk.cpp:28-197 driven like risk.py:144-166 (sampled ambiguity from ambiguity.cpp, then the risk)."""
    cv, disp_range, gmin, gmax = _confidence_case(H, W, D, d0, sp, H * W + D, quantised)
    etas = np.arange(0.0, 0.7, 0.01)
    grids = np.array([gmin, gmax], dtype=np.int64)
    _, samp = cc.compute_ambiguity_and_sampled_ambiguity(cv, etas, len(etas), grids, disp_range, True)
    exp = cc.compute_risk_and_sampled_risk(cv, samp, etas, len(etas), grids, disp_range, False)
    got = oracle.risk(cv, etas, gmin, gmax, disp_range)
    for g, e in zip(got, exp):  # risk_max, risk_min, disp_sup, disp_inf
        np.testing.assert_array_equal(g, e)
    assert np.isnan(got[0][1, 2])  # no cost at all


@pytest.mark.skipif(cc is None, reason="oracle/_ref not built")
@pytest.mark.parametrize("quantised", [False, True])
@pytest.mark.parametrize("thr,tf", [(0.9, -1.0), (0.5, 1.0), (1.0, -1.0), (0.0, 1.0)])
@pytest.mark.parametrize("H,W,D,d0,sp", [(9, 14, 12, -5, 1), (6, 11, 17, -2, 2), (5, 8, 9, 0, 4), (7, 9, 33, -30, 1)])
def test_interval_bounds_match_reference(oracle, H, W, D, d0, sp, thr, tf, quantised):
    """interval_bounds.cpp:28-161: both measure types, thresholds at the ends of [0, 1], NaN holes, empty ranges."""
    cv, disp_range, gmin, gmax = _confidence_case(H, W, D, d0, sp, H + W + D, quantised)
    grids = np.array([gmin, gmax], dtype=np.int64)
    exp = cc.compute_interval_bounds(cv, disp_range, thr, tf, grids, disp_range)
    got = oracle.interval_bounds(cv, thr, tf, gmin, gmax, disp_range)
    for g, e in zip(got, exp):
        np.testing.assert_array_equal(g, e)


it = ref.load("img_tools_cpp")


@pytest.mark.skipif(it is None, reason="oracle/_ref not built")
def test_interpolate_nodata_matches_reference(oracle):
    """img_tools.cpp:99-155: median of the first valid pixel along 8 directions, isolated invalid regions, an image
    without any valid pixel on some paths, NaN values among the neighbours."""
    rng = np.random.default_rng(3)
    for H, W in ((9, 13), (20, 31)):
        img = (rng.random((H, W)) * 200).astype(np.float32)
        msk = rng.choice([0, 0, 0, 1, 2], (H, W)).astype(np.int32)
        msk[:, 0] = 1
        msk[3:7, 4:9] = 2
        img[1, 1] = np.nan
        exp_i, exp_m = it.interpolate_nodata_sgm(img, msk, 0b01111000011, 1 << 10)
        got_i, got_m = oracle.interpolate_nodata(img, msk, 0b01111000011, 1 << 10)
        np.testing.assert_array_equal(got_i, exp_i)
        np.testing.assert_array_equal(got_m, exp_m)


vc = ref.load("validation_cpp")


@pytest.mark.skipif(vc is None, reason="oracle/_ref not built")
@pytest.mark.parametrize("which", ["occlusion_mc_cnn", "mismatch_mc_cnn", "occlusion_sgm", "mismatch_sgm"])
def test_interpolation_passes_match_reference(oracle, which):
    """interpolated_disparity.cpp: random maps with every validity flag, dense and sparse rejections, rows without any
    valid pixel, NaN disparities on valid pixels, 1-pixel-wide maps."""
    OCC, MIS, FOCC, FMIS, INV = 1 << 8, 1 << 9, 1 << 4, 1 << 5, 0b01111000011
    rng = np.random.default_rng(12)
    for H, W, p in ((6, 9, 0.3), (17, 23, 0.6), (30, 41, 0.1), (1, 12, 0.5), (12, 1, 0.5), (8, 8, 1.0)):
        disp = (rng.integers(-20, 20, (H, W)) + rng.choice([0, 0.25, -0.5], (H, W))).astype(np.float32)
        valid = np.where(rng.random((H, W)) < p, rng.choice([OCC, MIS, 1, 2, 64, OCC + 4, MIS + 8], (H, W)),
                         rng.choice([0, 4, 8, 16, 32, 2048], (H, W))).astype(np.int32)
        if H > 2 and W > 2:
            valid[2, :] = OCC
            disp[1, 1], valid[1, 1] = np.nan, 0
        if which == "occlusion_mc_cnn":
            exp = vc.interpolate_occlusion_mc_cnn(disp, valid, OCC, FOCC, INV)
        elif which == "mismatch_mc_cnn":
            exp = vc.interpolate_mismatch_mc_cnn(disp, valid, MIS, FMIS, INV)
        elif which == "occlusion_sgm":
            exp = vc.interpolate_occlusion_sgm(disp, valid, OCC, FOCC, INV)
        else:
            exp = vc.interpolate_mismatch_sgm(disp, valid, MIS, FMIS, OCC, INV)
        got = oracle.interpolate_disparity(which, disp, valid)
        np.testing.assert_array_equal(got[0], exp[0])
        np.testing.assert_array_equal(got[1], exp[1])


itv = ref.load("interval_tools_cpp")


@pytest.mark.skipif(itv is None, reason="oracle/_ref not built")
@pytest.mark.parametrize("depth", [0, 1, 2, 5])
@pytest.mark.parametrize("quantile", [1.0, 0.9, 0.5, 0.0])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_interval_regularization_matches_reference(seed, depth, quantile):
    """pandora_amd.interval_tools (host side of interval_bounds' regularisation) against the reference's compiled
    create_connected_graph / graph_regularization (cpp/src/interval_tools.cpp:32-234) on random ambiguous zones."""
    from pandora_amd import interval_tools as mine

    rng = np.random.default_rng(seed)
    H, W = 14, 40
    amb = rng.random((H, W))
    amb[rng.random((H, W)) < 0.05] = np.nan
    inf = rng.integers(-9, 0, (H, W)).astype(np.float32) / 2
    sup = inf + rng.integers(0, 9, (H, W)).astype(np.float32) / 4
    hole = rng.random((H, W)) < 0.1
    inf[hole], sup[hole] = np.nan, np.nan
    inf[3], sup[3] = np.nan, np.nan  # whole segments without any bound
    got = mine.interval_regularization(inf, sup, amb, 0.6, 5, depth, quantile)
    # the reference's driver (interval_tools.py:36-96) restated on its compiled functions
    pad = 2
    conf = np.nanmin(np.lib.stride_tricks.sliding_window_view(np.hstack((np.ones((H, pad)), amb, np.ones((H, pad)))), 5, axis=1), axis=-1)
    conf[:, -1] = 1
    steps = np.diff(np.hstack([np.ones((H, 1)), conf >= 0.6]), axis=-1)
    bl, br = np.argwhere(steps == -1), np.argwhere(steps == 1)
    br[:, 1] -= 1
    graph = itv.create_connected_graph(bl, br, depth)
    np.testing.assert_array_equal(mine.create_connected_graph(bl, br, depth), graph)
    exp = itv.graph_regularization(inf, sup, bl, br, graph, quantile)
    for g, e in zip(got, exp):
        np.testing.assert_array_equal(g, e)
    assert got[2].any() and not got[2].all()
