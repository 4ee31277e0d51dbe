"""GPU: whole pipelines driven through the reference-shaped plugin API (PandoraMachine ->
AbstractMatchingCost / Aggregation / Optimization / Disparity / Refinement -> C ABI -> HIP) against
the same pipeline composed from the CPU oracle."""
import json
import os

import numpy as np
import pytest

import pandora_amd
from pandora_amd import criteria, matching_cost
from pandora_amd.dataset import make_image
from pandora_amd.state_machine import PandoraMachine
from tests.test_gpu_parity import pair

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def oracle_pipeline(oracle, L, R, cfg, dmin, dmax, mskL=None, mskR=None, validity0=None, layers=None):
    p = cfg["pipeline"]
    mc = p["matching_cost"]
    win, sp = mc.get("window_size", 5), mc.get("subpix", 1)
    D = (dmax - dmin) * sp + 1
    method = mc["matching_cost_method"]
    if method == "census":
        cv, is_max, cmax = oracle.census_cost(L, R, D, dmin, sp, win), False, win * win
    elif method in ("sad", "ssd"):
        cv, is_max = oracle.sad_ssd(L, R, D, dmin, sp, win, method == "ssd"), False
        a, b = abs(L.max() - R.min()), abs(R.max() - L.min())
        # (sad_ssd.py:132-137's own expression: in float32 x * (w ** 2) and x * w * w round differently, and after CBCA's
        #  scaling cmax + 1 - SGM's cost of an invalid cell - then differs by a float32 step of 16: fuzz seed 21870)
        cmax = int(max(a, b) * (win ** 2)) if method == "sad" else int(max(a ** 2, b ** 2) * (win ** 2))
    else:
        cv, is_max, cmax = oracle.zncc(L, R, D, dmin, sp, win), True, 1
    if mskL is not None or mskR is not None:
        oracle.cv_masked(cv, dmin, sp, win, mskL=mskL, mskR=mskR)
    off = win // 2
    if "aggregation" in p:
        dist, inten = p["aggregation"].get("cbca_distance", 5), p["aggregation"].get("cbca_intensity", 30.0)

        def arms(im, msk, shifted):
            m = im.copy()
            if msk is not None:
                bad = msk != 0
                if shifted:
                    bad = bad[:, :-1] | bad[:, 1:]
                m[bad] = np.nan
            m = np.nan_to_num(oracle.median3(m), nan=np.inf)
            if off:
                m = m[off:-off, off:-off]
            return oracle.cross_support(np.ascontiguousarray(m), dist, inten)

        cl = arms(L, mskL, False)
        crs = [arms(im, mskR, k > 0) for k, im in enumerate(oracle.shift_right(R, sp))]
        oracle.cbca(cv, dmin, sp, off, cl, crs)
        cmax = cmax * (2 * dist - 1) ** 2
    if "optimization" in p:
        pen = p["optimization"].get("penalty", {})
        prior = p["optimization"].get("geometric_prior") or {"source": "internal"}
        if pen.get("p2_method", "constant") == "constant" and prior["source"] == "internal":
            cv = oracle.sgm(cv, pen.get("P1", 8), pen.get("P2", 32), is_max, float(cmax) + 1.0, p["optimization"].get("overcounting", False))
        else:  # penalty maps: the plugin's host-side 2-D work (hand-checked in test_host_api.py), the recurrence restated
            from pandora_amd import optimization as popt

            plug = popt.AbstractOptimization(None, **json.loads(json.dumps(p["optimization"])))
            maps = (np.full((8,) + L.shape, np.float32(plug._p2), np.float32) if plug._p2_method == "constant" else plug.p2_maps(L))
            if prior["source"] != "internal":
                maps[plug.path_cuts(make_image(L, **(layers or {})))] = 0
            cv = oracle.sgm_p2maps(cv, pen.get("P1", 8), maps, is_max, float(cmax) + 1.0, p["optimization"].get("overcounting", False))
    inv = p["disparity"].get("invalid_disparity", -9999)
    inv = np.nan if inv == "NaN" else inv
    disp, val = oracle.wta(cv, dmin, sp, is_max, inv, validity0)
    itp = None
    if "refinement" in p:
        itp, disp, val = oracle.refine(cv, disp, val, dmin, dmax, sp, is_max, p["refinement"]["refinement_method"])
    return cv, disp, val, itp


def run_machine(L, R, cfg, dmin, dmax, mskL=None, mskR=None, layers=None):
    left = make_image(L, disparity=[dmin, dmax], msk=mskL, **(layers or {}))
    right = make_image(R, msk=mskR)
    machine = PandoraMachine()
    cfg = json.loads(json.dumps(cfg))
    cfg["pipeline"] = machine.check_conf(cfg, left, right)["pipeline"]
    left_disp, _ = pandora_amd.run(machine, left, right, cfg)
    return machine, left_disp


def expected_validity(L, R, cfg, dmin, dmax, mskL, mskR, nan_pixels):
    left = make_image(L, disparity=[dmin, dmax], msk=mskL)
    right = make_image(R, msk=mskR)
    m = matching_cost.AbstractMatchingCost(**cfg["pipeline"]["matching_cost"])
    cv = m.allocate_cost_volume(left, (left["disparity"].sel(band_disp="min"), left["disparity"].sel(band_disp="max")))
    cv = criteria.validity_mask(left, right, cv)
    criteria.mask_invalid_variable_disparity_range(cv, nan_pixels)
    if cv.attrs["offset_row_col"] > 0:
        criteria.mask_border(cv)
    return cv["validity_mask"].data


CASES = [
    ("census5+sgm+wta+vfit", {"pipeline": {"matching_cost": {"matching_cost_method": "census", "window_size": 5},
                                          "optimization": {"optimization_method": "sgm", "penalty": {"P1": 8, "P2": 32}},
                                          "disparity": {"disparity_method": "wta", "invalid_disparity": "NaN"},
                                          "refinement": {"refinement_method": "vfit"}}}, (48, 80, -12, 3), True, True),
    ("zncc5+cbca+wta+quadratic", {"pipeline": {"matching_cost": {"matching_cost_method": "zncc", "window_size": 5},
                                              "aggregation": {"aggregation_method": "cbca", "cbca_intensity": 30.0, "cbca_distance": 5},
                                              "disparity": {"disparity_method": "wta", "invalid_disparity": -9999},
                                              "refinement": {"refinement_method": "quadratic"}}}, (40, 70, -8, 2), False, False),
    ("sad3/subpix2+wta+vfit", {"pipeline": {"matching_cost": {"matching_cost_method": "sad", "window_size": 3, "subpix": 2},
                                           "disparity": {"disparity_method": "wta", "invalid_disparity": -9999},
                                           "refinement": {"refinement_method": "vfit"}}}, (30, 55, -4, 4), True, True),
    ("census5+cbca+sgm+wta+vfit (BASELINE configs[1])",
     {"pipeline": {"matching_cost": {"matching_cost_method": "census", "window_size": 5},
                   "aggregation": {"aggregation_method": "cbca"},
                   "optimization": {"optimization_method": "sgm", "penalty": {"P1": 8, "P2": 32}},
                   "disparity": {"disparity_method": "wta", "invalid_disparity": "NaN"},
                   "refinement": {"refinement_method": "vfit"}}}, (36, 64, -10, 0), True, False),
]


@pytest.mark.parametrize("name,cfg,shape,integer,with_masks", CASES, ids=[c[0] for c in CASES])
def test_machine_matches_oracle_pipeline(oracle, name, cfg, shape, integer, with_masks):
    H, W, dmin, dmax = shape
    L, R = pair(H, W, seed=H + W, integer=integer)
    mskL = mskR = None
    if with_masks:
        rng = np.random.default_rng(3)
        mskL = rng.choice([0, 0, 0, 0, 0, 0, 0, 1, 2], (H, W)).astype(np.int16)
        mskR = rng.choice([0, 0, 0, 0, 0, 0, 0, 1, 2], (H, W)).astype(np.int16)
    machine, left_disp = run_machine(L, R, cfg, dmin, dmax, mskL, mskR)
    got_cv = machine.left_cv["cost_volume"].data
    # expected: oracle matching cost -> NaN pixels -> host validity -> oracle rest
    mc_only = {"pipeline": {"matching_cost": cfg["pipeline"]["matching_cost"], "disparity": {"disparity_method": "wta"}}}
    cv0, _, _, _ = oracle_pipeline(oracle, L, R, mc_only, dmin, dmax, mskL, mskR)
    val0 = expected_validity(L, R, cfg, dmin, dmax, mskL, mskR, np.min(np.isnan(cv0), axis=2))
    ecv, edisp, eval_, eitp = oracle_pipeline(oracle, L, R, cfg, dmin, dmax, mskL, mskR, val0)
    np.testing.assert_array_equal(np.isnan(got_cv), np.isnan(ecv))
    if cfg["pipeline"]["matching_cost"]["matching_cost_method"] == "zncc":
        np.testing.assert_allclose(got_cv, ecv, rtol=0, atol=1e-5)  # float64 window sums: within 1e-5
        # WTA / refinement re-checked on the GPU volume itself (ties can flip on 1e-7 differences)
        edisp, eval_ = oracle.wta(got_cv, dmin, 1, True, -9999, val0)
        eitp, edisp, eval_ = oracle.refine(got_cv, edisp, eval_, dmin, dmax, 1, True, "quadratic")
    else:
        np.testing.assert_array_equal(got_cv, ecv)
    np.testing.assert_array_equal(left_disp["disparity_map"].data, edisp)
    np.testing.assert_array_equal(left_disp["validity_mask"].data, eval_)
    np.testing.assert_array_equal(left_disp["interpolated_coeff"].data, eitp)
    assert left_disp.attrs["refinement"] == cfg["pipeline"]["refinement"]["refinement_method"]
    assert machine.left_cv.attrs["type_measure"] in ("min", "max")


with open(os.path.join(ROOT, "tests", "golden", "validity_mask_cases.json")) as f:
    VM_CASES = json.load(f)["cases"]


@pytest.mark.parametrize("case", VM_CASES, ids=lambda c: c["id"])
def test_validity_mask_goldens_through_the_gpu_path(case):
    """tests/test_criteria.py::test_validity_mask of the reference, with the product's own
    compute_cost_volume + cv_masked (HIP) instead of the oracle."""
    L, R = np.array(case["left_data"], np.float32), np.array(case["right_data"], np.float32)
    left = make_image(L, disparity=case["disparity"], msk=np.array(case["left_msk"]), valid_pixels=case["left_valid"],
                      no_data_mask=case["left_nodata"])
    right = make_image(R, msk=np.array(case["right_msk"]), valid_pixels=case["right_valid"], no_data_mask=case["right_nodata"])
    m = matching_cost.AbstractMatchingCost(matching_cost_method="sad", window_size=case["window_size"], subpix=1)
    dmin, dmax = left["disparity"].sel(band_disp="min"), left["disparity"].sel(band_disp="max")
    cv = m.allocate_cost_volume(left, (dmin, dmax))
    cv = criteria.validity_mask(left, right, cv)
    cv = m.compute_cost_volume(left, right, cv)
    m.cv_masked(left, right, cv, dmin, dmax)
    np.testing.assert_array_equal(cv["validity_mask"].data, np.array(case["gt_mask"]))


def test_cost_volume_is_lazy_and_writable():
    L, R = pair(20, 30, seed=1)
    left, right = make_image(L, disparity=[-3, 3]), make_image(R)
    m = matching_cost.AbstractMatchingCost(matching_cost_method="census", window_size=3)
    cv = m.allocate_cost_volume(left, (left["disparity"].sel(band_disp="min"), left["disparity"].sel(band_disp="max")))
    cv = m.compute_cost_volume(left, right, cv)
    host = cv["cost_volume"].data
    assert host.shape == (20, 30, 7) and host.dtype == np.float32
    host2 = np.where(np.isnan(host), np.nan, host + 1).astype(np.float32)
    cv["cost_volume"].data = host2  # the reference mutates cv["cost_volume"].data between steps
    np.testing.assert_array_equal(cv["cost_volume"].data, host2)


# ---- validation step (cross-checking) through the machine -----------------------------------
VALIDATION_CFG = {"pipeline": {"matching_cost": {"matching_cost_method": "census", "window_size": 5},
                               "optimization": {"optimization_method": "sgm", "penalty": {"P1": 8, "P2": 32}},
                               "disparity": {"disparity_method": "wta", "invalid_disparity": "NaN"},
                               "refinement": {"refinement_method": "vfit"},
                               "validation": {"validation_method": "cross_checking_accurate", "cross_checking_threshold": 1.0}}}


def _geometry_validity(left_arr, right_arr, cfg, dmin, dmax):
    left = make_image(left_arr, disparity=[dmin, dmax])
    right = make_image(right_arr)
    m = matching_cost.AbstractMatchingCost(**cfg["pipeline"]["matching_cost"])
    cv = m.allocate_cost_volume(left, (left["disparity"].sel(band_disp="min"), left["disparity"].sel(band_disp="max")))
    return criteria.validity_mask(left, right, cv)


@pytest.mark.parametrize("method", ["cross_checking_accurate", "cross_checking_fast"])
def test_machine_validation_matches_oracle(oracle, method):
    """Left AND right pipelines on the device, then the two cross-checks (state_machine.py:493-519): validity masks,
    the left-right distance layer and the final disparities must equal the oracle composition."""
    H, W, dmin, dmax = 44, 90, -11, 2
    L, R = pair(H, W, seed=77)
    R[10:20, 30:40] = 255 - R[10:20, 30:40]  # provoke mismatches / occlusions
    cfg = json.loads(json.dumps(VALIDATION_CFG))
    cfg["pipeline"]["validation"]["validation_method"] = method
    machine, left_disp = run_machine(L, R, cfg, dmin, dmax)
    right_disp = machine.right_disparity
    no_val = {"pipeline": {k: v for k, v in cfg["pipeline"].items() if k != "validation"}}
    mc_only = {"pipeline": {"matching_cost": cfg["pipeline"]["matching_cost"], "disparity": {"disparity_method": "wta"}}}
    # left side
    cv0, _, _, _ = oracle_pipeline(oracle, L, R, mc_only, dmin, dmax)
    val0 = expected_validity(L, R, cfg, dmin, dmax, None, None, np.min(np.isnan(cv0), axis=2))
    lcv, ldisp, lval, _ = oracle_pipeline(oracle, L, R, no_val, dmin, dmax, validity0=val0)
    # right side
    if method == "cross_checking_accurate":
        rcv0, _, _, _ = oracle_pipeline(oracle, R, L, mc_only, -dmax, -dmin)
        rval0 = expected_validity(R, L, cfg, -dmax, -dmin, None, None, np.min(np.isnan(rcv0), axis=2))
        rcv, rdisp, rval, _ = oracle_pipeline(oracle, R, L, no_val, -dmax, -dmin, validity0=rval0)
    else:
        rcv = oracle.reverse_cost_volume(lcv, -dmax)
        rval0 = _geometry_validity(R, L, cfg, -dmax, -dmin)["validity_mask"].data  # no cv_masked on the fast right side
        rdisp, rval = oracle.wta(rcv, -dmax, 1, False, np.nan, rval0)
        _, rdisp, rval = oracle.refine(rcv, rdisp, rval, -dmax, -dmin, 1, False, "vfit")
    thr = cfg["pipeline"]["validation"]["cross_checking_threshold"]
    lval2, lconf = oracle.cross_checking(ldisp, lval, rdisp, dmin, dmax, thr)
    rval2, rconf = oracle.cross_checking(rdisp, rval, ldisp, -dmax, -dmin, thr)
    for ds_val in (lval2, rval2):  # mask_border (validation.py:368-369)
        ds_val[:2, :] = ds_val[-2:, :] = 1
        ds_val[2:-2, :2] = ds_val[2:-2, -2:] = 1
    np.testing.assert_array_equal(left_disp["disparity_map"].data, ldisp)
    np.testing.assert_array_equal(left_disp["validity_mask"].data, lval2)
    assert list(left_disp.coords["indicator"])[-1] == "confidence_from_left_right_consistency"
    np.testing.assert_array_equal(left_disp["confidence_measure"].data[:, :, -1], lconf)
    assert left_disp.attrs["validation"] == method
    assert (lval2 & (1 << 8)).any() or (lval2 & (1 << 9)).any()
    if method == "cross_checking_accurate":
        np.testing.assert_array_equal(right_disp["disparity_map"].data, rdisp)
        np.testing.assert_array_equal(right_disp["validity_mask"].data, rval2)
        np.testing.assert_array_equal(right_disp["confidence_measure"].data[:, :, -1], rconf)
    else:
        assert not right_disp.data_vars and machine.right_cv is None  # state_machine.py:514-519


def test_machine_filter_then_validation(oracle):
    """The sample pipeline order (a_semi_global_matching.json: ... disparity, refinement, filter, validation) with the
    median filter applied to the left AND right maps before the cross-check (state_machine.py:449-473)."""
    H, W, dmin, dmax = 40, 72, -9, 3
    L, R = pair(H, W, seed=5)
    cfg = json.loads(json.dumps(VALIDATION_CFG))
    cfg["pipeline"] = {"matching_cost": cfg["pipeline"]["matching_cost"], "optimization": cfg["pipeline"]["optimization"],
                       "disparity": cfg["pipeline"]["disparity"], "refinement": cfg["pipeline"]["refinement"],
                       "filter": {"filter_method": "median", "filter_size": 3},
                       "validation": {"validation_method": "cross_checking_accurate"}}
    machine, left_disp = run_machine(L, R, cfg, dmin, dmax)
    base = {"pipeline": {k: v for k, v in cfg["pipeline"].items() if k not in ("filter", "validation")}}
    mc_only = {"pipeline": {"matching_cost": cfg["pipeline"]["matching_cost"], "disparity": {"disparity_method": "wta"}}}
    sides = []
    for a, b, lo, hi in ((L, R, dmin, dmax), (R, L, -dmax, -dmin)):
        cv0, _, _, _ = oracle_pipeline(oracle, a, b, mc_only, lo, hi)
        val0 = expected_validity(a, b, cfg, lo, hi, None, None, np.min(np.isnan(cv0), axis=2))
        _, d, v, _ = oracle_pipeline(oracle, a, b, base, lo, hi, validity0=val0)
        sides.append((oracle.filter_median_disparity(d, v, 3), v))
    (ld, lv), (rd, rv) = sides
    lv2, _ = oracle.cross_checking(ld, lv, rd, dmin, dmax, 1.0)
    lv2[:2, :] = lv2[-2:, :] = 1
    lv2[2:-2, :2] = lv2[2:-2, -2:] = 1
    np.testing.assert_array_equal(left_disp["disparity_map"].data, ld)
    np.testing.assert_array_equal(left_disp["validity_mask"].data, lv2)
    assert left_disp.attrs["filter"] == "median" and left_disp.attrs["validation"] == "cross_checking_accurate"


def test_machine_with_gradient_penalties(oracle):
    """optimization.penalty.p2_method "negativeGradient" / "inverseGradient" through the machine (plugin_libsgm.rst:20-27,
    168-290): the plugin builds P2 per pixel and path from the left image's gradient (its own reading of the documentation:
    module docstring of optimization/sgm.py), the device runs the recurrence; the maps after WTA + vfit equal the oracle
    pipeline run with the same maps.  UNPINNED against libSGM, like the constant method."""
    from pandora_amd.optimization.sgm import Sgm

    H, W, dmin, dmax = 40, 64, -8, 2
    L, R = pair(H, W, seed=33)
    for pen in ({"p2_method": "negativeGradient", "P1": 4, "P2": 12, "alpha": 0.5, "gamma": 60},
                {"p2_method": "inverseGradient", "P1": 4, "P2": 12, "alpha": 90.0, "beta": 2, "gamma": 5}):
        cfg = {"pipeline": {"matching_cost": {"matching_cost_method": "census", "window_size": 5},
                            "optimization": {"optimization_method": "sgm", "penalty": dict(pen)},
                            "disparity": {"disparity_method": "wta", "invalid_disparity": "NaN"},
                            "refinement": {"refinement_method": "vfit"}}}
        _, got = run_machine(L, R, json.loads(json.dumps(cfg)), dmin, dmax)
        plugin = Sgm(**cfg["pipeline"]["optimization"])
        maps = plugin.p2_maps(L)
        assert maps.shape == (8, H, W) and maps.min() >= 12.0 and maps.max() > 12.0
        np.testing.assert_array_equal(maps[0][:, 0], np.float32(max(12.0, pen["gamma"] + (0 if "beta" not in pen else pen["alpha"] / pen["beta"]))))
        ocv = oracle.census_cost(L, R, dmax - dmin + 1, dmin, 1, 5)
        exp_cv = oracle.sgm_p2maps(ocv, 4.0, maps, False, 26.0, False)
        val0 = expected_validity(L, R, cfg, dmin, dmax, None, None, np.min(np.isnan(ocv), axis=2))
        odisp, oval = oracle.wta(exp_cv, dmin, 1, False, np.nan, val0)
        _, odisp, oval = oracle.refine(exp_cv, odisp, oval, dmin, dmax, 1, False, "vfit")
        np.testing.assert_array_equal(got["disparity_map"].data, odisp)
        np.testing.assert_array_equal(got["validity_mask"].data, oval)


def test_machine_with_disparity_denoiser(oracle):
    """filter_method "disparity_denoiser" through the machine (disparity_denoiser.py:223-313): the plugin against the reference's
    end-to-end vector (tests/golden/disparity_denoiser.json: mono- and multiband image, named band), then census + SGM + WTA + vfit
    + the filter on a pair, where the filtered map must equal the restatement applied to the unfiltered run's map."""
    import os

    from scipy.ndimage import gaussian_filter

    from pandora_amd import filter as flt
    from pandora_amd.dataset import Dataset, make_image

    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "disparity_denoiser.json")) as f:
        e = json.load(f)["end_to_end"]
    band = np.array(e["band"], np.float32)
    for left, cfg in ((make_image(band), {}),
                      (make_image(np.stack([band, band * 3 + 1, band * 2]), band_names=["red", "green", "blue"]), {"band": "red"})):
        ds = Dataset({"disparity_map": (("row", "col"), np.array(e["disp"], np.float32)),
                      "validity_mask": (("row", "col"), np.zeros((2, 2), np.uint16))}, coords={"row": np.arange(2), "col": np.arange(2)})
        flt.AbstractFilter(cfg={"filter_method": "disparity_denoiser", **e["cfg"], **cfg}).filter_disparity(ds, left)
        np.testing.assert_allclose(ds["disparity_map"].data, np.array(e["expected"]), rtol=2e-7)
        assert ds.attrs["filter"] == "disparity_denoiser"
    H, W, dmin, dmax = 48, 80, -9, 3
    L, R = pair(H, W, seed=17)
    base = {"pipeline": {"matching_cost": {"matching_cost_method": "census", "window_size": 5},
                         "optimization": {"optimization_method": "sgm", "penalty": {"P1": 8, "P2": 32}},
                         "disparity": {"disparity_method": "wta", "invalid_disparity": "NaN"},
                         "refinement": {"refinement_method": "vfit"}}}
    _, plain = run_machine(L, R, json.loads(json.dumps(base)), dmin, dmax)
    cfg = json.loads(json.dumps(base))
    cfg["pipeline"]["filter"] = {"filter_method": "disparity_denoiser", "filter_size": 7}
    _, filtered = run_machine(L, R, cfg, dmin, dmax)
    d0 = np.asarray(plain["disparity_map"].data)
    grad = np.gradient(gaussian_filter(d0, sigma=1.5))
    exp = oracle.denoise_disparity(d0, np.asarray(plain["validity_mask"].data), L, grad[0], grad[1], 7, 4.0, 100.0, 12.0)
    np.testing.assert_allclose(filtered["disparity_map"].data, exp, rtol=1e-6, atol=1e-6, equal_nan=True)
    assert filtered.attrs["filter"] == "disparity_denoiser" and not np.array_equal(exp, d0, equal_nan=True)


def test_machine_confidence_steps(oracle):
    """cost_volume_confidence steps between the cost volume and the disparity (state_machine.py:558-587): std_intensity
    (host) and ambiguity (device) land as indicator layers on the disparity dataset; the ambiguity layer equals
    1 - percentile-normalised integral of the restatement pinned by the compiled reference."""
    H, W, dmin, dmax = 36, 60, -7, 3
    L, R = pair(H, W, seed=21)
    cfg = {"pipeline": {"matching_cost": {"matching_cost_method": "census", "window_size": 5},
                        "cost_volume_confidence": {"confidence_method": "std_intensity"},
                        "cost_volume_confidence.amb": {"confidence_method": "ambiguity", "eta_max": 0.7, "eta_step": 0.01},
                        "disparity": {"disparity_method": "wta", "invalid_disparity": "NaN"},
                        "validation": {"validation_method": "cross_checking_fast"}}}
    machine, left_disp = run_machine(L, R, cfg, dmin, dmax)
    names = list(left_disp.coords["indicator"])
    assert names == ["confidence_from_intensity_std", "confidence_from_ambiguity.amb", "confidence_from_left_right_consistency"]
    conf = left_disp["confidence_measure"].data
    assert conf.shape == (H, W, 3)
    cv = oracle.census_cost(L, R, dmax - dmin + 1, dmin, 1, 5)
    etas = np.arange(0.0, 0.7, 0.01)
    amb = oracle.ambiguity(cv, etas, np.full((H, W), dmin), np.full((H, W), dmax), (dmin + np.arange(dmax - dmin + 1)).astype(np.float32))
    lo, hi = np.percentile(amb, 1.0), np.percentile(amb, 99.0)
    n = np.clip(amb, lo, hi)
    exp = 1 - (n - n.min()) / (n.max() - n.min())
    np.testing.assert_allclose(conf[:, :, 1], exp, rtol=1e-6)
    assert np.isnan(conf[0, 0, 0]) and np.isfinite(conf[5, 5, 0])


def test_interval_bounds_pipeline_of_the_reference():
    """tests/test_confidence/test_interval_bounds.py:30-116 as written: SAD window 1 on the confidence pair (left mask on two
    pixels), interval_bounds at possibility 0.7, WTA, median filter -> the two layers and their expected maps."""
    from tests.golden import known_answers as ka

    L, R = np.array(ka.CONFIDENCE_LEFT, np.float32), np.array(ka.CONFIDENCE_RIGHT, np.float32)
    cfg = {"pipeline": {"matching_cost": {"matching_cost_method": "sad", "window_size": 1, "subpix": 1},
                        "cost_volume_confidence": {"confidence_method": "interval_bounds", "possibility_threshold": 0.7},
                        "disparity": {"disparity_method": "wta"},
                        "filter": {"filter_method": "median"}}}
    _, left = run_machine(L, R, cfg, -1, 1, mskL=np.array(ka.CONFIDENCE_LEFT_MASK, np.int16), mskR=np.zeros((4, 4), np.int16))
    assert list(left.coords["indicator"]) == ["confidence_from_interval_bounds_inf", "confidence_from_interval_bounds_sup"]
    np.testing.assert_allclose(left["confidence_measure"].data[:, :, 0], np.array(ka.INTERVAL_BOUNDS["inf"], np.float32), rtol=1e-6)
    np.testing.assert_allclose(left["confidence_measure"].data[:, :, 1], np.array(ka.INTERVAL_BOUNDS["sup"], np.float32), rtol=1e-6)


def test_machine_risk_and_regularized_interval_bounds(oracle):
    """ambiguity, then interval_bounds regularised with that ambiguity (interval_bounds.py:171-189), then risk, after a Census
    volume: layer names in the reference's order and every layer equal to the pinned restatements."""
    from pandora_amd import interval_tools

    H, W, dmin, dmax = 40, 70, -8, 3
    L, R = pair(H, W, seed=5)
    cfg = {"pipeline": {"matching_cost": {"matching_cost_method": "census", "window_size": 5},
                        "cost_volume_confidence.amb": {"confidence_method": "ambiguity", "eta_max": 0.7, "eta_step": 0.01},
                        "cost_volume_confidence.int": {"confidence_method": "interval_bounds", "regularization": True,
                                                       "ambiguity_indicator": "amb", "vertical_depth": 2, "quantile_regularization": 0.9},
                        "cost_volume_confidence.risk": {"confidence_method": "risk", "eta_max": 0.5, "eta_step": 0.02},
                        "disparity": {"disparity_method": "wta", "invalid_disparity": "NaN"}}}
    _, left = run_machine(L, R, cfg, dmin, dmax)
    assert list(left.coords["indicator"]) == [
        "confidence_from_ambiguity.amb", "confidence_from_interval_bounds_inf.int", "confidence_from_interval_bounds_sup.int",
        "confidence_from_risk_max.risk", "confidence_from_risk_min.risk", "confidence_from_disp_sup_from_risk.risk",
        "confidence_from_disp_inf_from_risk.risk"]
    conf = left["confidence_measure"].data
    cv = oracle.census_cost(L, R, dmax - dmin + 1, dmin, 1, 5)
    gmin, gmax = np.full((H, W), dmin, np.int64), np.full((H, W), dmax, np.int64)
    disp_range = (dmin + np.arange(dmax - dmin + 1)).astype(np.float32)
    lo, hi = oracle.interval_bounds(cv, 0.9, -1.0, gmin, gmax, disp_range)
    lo, hi, mask = interval_tools.interval_regularization(lo, hi, conf[:, :, 0], 0.6, 5, 2, 0.9)
    assert mask.any()
    np.testing.assert_array_equal(conf[:, :, 1], lo)
    np.testing.assert_array_equal(conf[:, :, 2], hi)
    exp = oracle.risk(cv, np.arange(0.0, 0.5, 0.02), gmin, gmax, disp_range)
    for k in range(4):
        np.testing.assert_array_equal(conf[:, :, 3 + k], exp[k])


def _interval_dataset(layers, names):
    from pandora_amd.dataset import DataArray, Dataset

    H, W = np.shape(layers[0])
    ds = Dataset(coords={"row": np.arange(H), "col": np.arange(W), "indicator": np.array(names)})
    ds["disparity_map"] = (("row", "col"), np.zeros((H, W), np.float32))
    ds["validity_mask"] = (("row", "col"), np.zeros((H, W), np.int16))
    ds["confidence_measure"] = DataArray(np.stack([np.array(x, np.float32) for x in layers], axis=2), ("row", "col", "indicator"),
                                         {"indicator": list(names)})
    return ds


def test_median_for_intervals_reference_vectors():
    """tests/test_filter.py:662-800: the filter on the two bound layers, plain and with the regularisation of ambiguous segments
    (expected bounds, and the INTERVAL_REGULARIZED bit on the regularised pixels)."""
    from pandora_amd import filter as flt
    from pandora_amd.margins import Margins
    from tests.golden import known_answers as ka

    c = ka.MEDIAN_FOR_INTERVALS
    names = ["confidence_from_interval_bounds_inf", "confidence_from_interval_bounds_sup"]
    ds = _interval_dataset([c["inf"], c["sup"]], names)
    f = flt.AbstractFilter(cfg=dict(c["plain"]["cfg"]))
    assert f.margins == Margins(3, 3, 3, 3) and flt.AbstractFilter(cfg=dict(c["plain"]["cfg"]), step=2).margins == Margins(6, 6, 6, 6)
    f.filter_disparity(ds)
    np.testing.assert_array_equal(ds["confidence_measure"].sel({"indicator": names[0]}).data, np.array(c["plain"]["inf"], np.float32))
    np.testing.assert_array_equal(ds["confidence_measure"].sel({"indicator": names[1]}).data, np.array(c["plain"]["sup"], np.float32))

    r = c["regularized"]
    ds = _interval_dataset([r["ambiguity"], c["inf"], c["sup"]], ["confidence_from_ambiguity"] + names)
    flt.AbstractFilter(cfg=dict(r["cfg"])).filter_disparity(ds)
    np.testing.assert_allclose(ds["confidence_measure"].sel({"indicator": names[0]}).data, np.array(r["inf"], np.float32), 1e-7, 1e-7)
    np.testing.assert_allclose(ds["confidence_measure"].sel({"indicator": names[1]}).data, np.array(r["sup"], np.float32), 1e-7, 1e-7)
    np.testing.assert_array_equal(ds["validity_mask"].data, np.array(r["validity"], np.int16))


def test_machine_with_interval_filter_after_cross_checking(oracle):
    """A pipeline with interval bounds and the median_for_intervals filter under cross_checking_fast (the right side skips this
    filter, state_machine.py:466-473): the left layers equal the device median of the pinned bounds."""
    H, W, dmin, dmax = 30, 52, -6, 2
    L, R = pair(H, W, seed=9)
    cfg = {"pipeline": {"matching_cost": {"matching_cost_method": "census", "window_size": 5},
                        "cost_volume_confidence": {"confidence_method": "interval_bounds"},
                        "disparity": {"disparity_method": "wta", "invalid_disparity": "NaN"},
                        "filter": {"filter_method": "median_for_intervals", "filter_size": 3},
                        "validation": {"validation_method": "cross_checking_fast"}}}
    _, left = run_machine(L, R, cfg, dmin, dmax)
    cv = oracle.census_cost(L, R, dmax - dmin + 1, dmin, 1, 5)
    lo, hi = oracle.interval_bounds(cv, 0.9, -1.0, np.full((H, W), dmin, np.int64), np.full((H, W), dmax, np.int64),
                                    (dmin + np.arange(dmax - dmin + 1)).astype(np.float32))
    conf = left["confidence_measure"].data
    for k, m in enumerate((lo, hi)):
        # NaN bounds (the census frame) stay NaN, the others are NaN-ignoring 3x3 medians of the pinned bounds
        np.testing.assert_array_equal(conf[:, :, k], oracle.filter_median_disparity(m, np.zeros((H, W), np.int64), 3))
    assert list(left.coords["indicator"])[:2] == ["confidence_from_interval_bounds_inf", "confidence_from_interval_bounds_sup"]


@pytest.mark.parametrize("method,order,subpix,with_cbca", [("sad", 3, 2, False), ("census", 2, 4, False), ("zncc", 5, 2, False), ("sad", 3, 2, True)])
def test_spline_order_resamples_the_right_image_like_the_reference(oracle, method, order, subpix, with_cbca):
    """matching_cost's "spline_order" (img_tools.py:713-752): the sub-pixel right images are scipy.ndimage.zoom(order=...) of the right
    image - the reference's own expression, evaluated on the host and handed to the device (pmx_set_shifted_right); the costs are the
    oracle's on those images.  CBCA's cross supports keep order 1 whatever the matching cost's (cbca.py:248-250 calls shift_right_img
    with its default), so an aggregation step re-uploads the pair with linear shifts."""
    from scipy.ndimage import zoom

    H, W, dmin, dmax, win = 40, 66, -5, 3, 3 if method != "census" else 5
    L, R = pair(H, W, seed=order, integer=False)
    D = (dmax - dmin) * subpix + 1
    z = zoom(R, (1, (W * subpix - (subpix - 1)) / float(W)), order=order)
    shifted = [np.ascontiguousarray(z[:, k::subpix], np.float32) for k in range(1, subpix)]
    if method == "census":
        exp = oracle.census_cost(L, R, D, dmin, subpix, win, shifted=shifted)
    elif method == "sad":
        exp = oracle.sad_ssd(L, R, D, dmin, subpix, win, False, shifted=shifted)
    else:
        exp = oracle.zncc(L, R, D, dmin, subpix, win, shifted=shifted)
    pipe = {"matching_cost": {"matching_cost_method": method, "window_size": win, "subpix": subpix, "spline_order": order},
            "disparity": {"disparity_method": "wta", "invalid_disparity": -9999}}
    if with_cbca:
        pipe = {"matching_cost": pipe["matching_cost"], "aggregation": {"aggregation_method": "cbca"}, "disparity": pipe["disparity"]}
        off = win // 2

        def arms(im):
            m = np.nan_to_num(oracle.median3(im.copy()), nan=np.inf)[off:-off, off:-off]
            return oracle.cross_support(np.ascontiguousarray(m), 5, 30.0)

        oracle.cbca(exp, dmin, subpix, off, arms(L), [arms(im) for im in oracle.shift_right(R, subpix)])  # linear shifts here
    # through the plugin API directly: PandoraMachine drops "spline_order" when it checks the configuration, exactly like the
    # reference (matching_cost.py:153-156: "a pandora2d setting"), so a machine run always resamples with order 1
    from pandora_amd import aggregation

    left, right = make_image(L, disparity=[dmin, dmax]), make_image(R)
    mc = matching_cost.AbstractMatchingCost(**pipe["matching_cost"])
    cv = mc.allocate_cost_volume(left, (left["disparity"].sel(band_disp="min"), left["disparity"].sel(band_disp="max")))
    cv = mc.compute_cost_volume(left, right, cv)
    if with_cbca:
        aggregation.AbstractAggregation(**pipe["aggregation"]).cost_volume_aggregation(left, right, cv)
    got = cv["cost_volume"].data
    if method == "zncc":
        np.testing.assert_allclose(got, exp, rtol=0, atol=1e-5)
    else:
        np.testing.assert_array_equal(got, exp)
    machine, _ = run_machine(L, R, {"pipeline": pipe}, dmin, dmax)  # order 1 through the machine: a different volume
    assert not np.array_equal(np.nan_to_num(got), np.nan_to_num(machine.left_cv["cost_volume"].data))


def test_machine_keeps_its_maps_on_the_device_until_they_are_read():
    """A PandoraMachine run of census + SGM + WTA + vfit never brings a full-size map to the host by itself: the volume's validity
    mask stays a recipe (criteria.LazyValidity) that the device carries out (pmx_compose_validity), the result maps are
    engine.DeviceMapArrays still on the GPU when run() returns; reading them - and the volume's mask, afterwards - gives what the
    eager host path gives (the oracle composition checks of this file cover the values)."""
    import pandora_amd
    from pandora_amd import criteria
    from pandora_amd.engine import DeviceMapArray
    from pandora_amd.state_machine import PandoraMachine

    L, R = pair(60, 90, 3)
    left, right = make_image(L, disparity=[-9, 2]), make_image(R)
    machine = PandoraMachine()
    cfg = {"pipeline": {"matching_cost": {"matching_cost_method": "census", "window_size": 5},
                        "optimization": {"optimization_method": "sgm", "penalty": {"P1": 8, "P2": 32}},
                        "disparity": {"disparity_method": "wta", "invalid_disparity": "NaN"},
                        "refinement": {"refinement_method": "vfit"}}}
    cfg["pipeline"] = machine.check_conf(cfg, left, right)["pipeline"]
    out, _ = pandora_amd.run(machine, left, right, cfg)
    lazy = machine.left_cv["validity_mask"]
    assert isinstance(lazy, criteria.LazyValidity) and lazy.pending
    for k in ("disparity_map", "validity_mask", "interpolated_coeff"):
        assert isinstance(out[k], DeviceMapArray) and out[k].on_device(), k
    vm = np.array(out["validity_mask"].data)
    # the volume's own mask, materialised on the host now (snapshot of the all-NaN pixels taken at cv_masked time), is the
    # result's mask minus what WTA / refinement added
    np.testing.assert_array_equal(lazy.data & 0b111, vm & 0b111)
    assert (lazy.data[:2] == 1).all() and (lazy.data[:, -2:] == 1).all() and not lazy.pending
    assert np.isfinite(out["disparity_map"].data[10:-10, 20:-10]).all()


def test_geometric_prior_optimises_every_segment_on_its_own():
    """plugin_libsgm.rst:49-78 (3SGM's piecewise optimisation): "for each segment, optimization will only be applied inside this
    segment".  (1) The machine with a segmentation, the reference's classification file or an edge map == the restatement with
    the same penalty maps (the cut (pixel, direction) pairs are zeros of the P2 maps), bit for bit; (2) what the sentence says:
    with the image cut in two by a vertical segment border, each half comes out exactly as if it had been optimised alone."""
    from PIL import Image

    from oracle import capi as orc
    from pandora_amd import optimization, runtime

    cones = os.path.join(os.path.dirname(__file__), "golden", "cones")
    L = np.array(Image.open(os.path.join(cones, "left.png"))).astype(np.float32)[100:180, 60:200]
    R = np.array(Image.open(os.path.join(cones, "right.png"))).astype(np.float32)[100:180, 60:200]
    from pandora_amd.tiff_reader import read_tiff
    classif = read_tiff(os.path.join(cones, "left_classif.tif"))[0][:, 100:180, 60:200].astype(np.int16)
    H, W = L.shape
    segm = np.zeros((H, W), np.int16)
    segm[:, 77:] = 4
    segm[20:40, 30:60] = 9
    edges = np.zeros((H, W), np.int16)
    edges[50, :] = 1
    edges[:, 100] = 3
    layers = {"segm": dict(segm=segm), "edges": dict(edges=edges), "classif": dict(classif=(classif, ["cornfields", "olive tree", "forest"]))}
    for source, kw in layers.items():
        for p2_method in ("constant", "inverseGradient"):
            prior = {"source": source}
            if source == "classif":
                prior["classes"] = ["olive tree", "forest"]
            pen = {"P1": 8, "P2": 32, "p2_method": p2_method}
            cfg = {"pipeline": {"matching_cost": {"matching_cost_method": "census", "window_size": 5},
                                "optimization": {"optimization_method": "sgm", "penalty": pen, "geometric_prior": prior},
                                "disparity": {"disparity_method": "wta", "invalid_disparity": "NaN"}}}
            left, right = make_image(L, disparity=[-30, 0], **kw), make_image(R)
            machine = PandoraMachine()
            cfg["pipeline"] = machine.check_conf(cfg, left, right)["pipeline"]
            pandora_amd.run(machine, left, right, cfg)
            got = machine.left_cv["cost_volume"].data
            plugin = optimization.AbstractOptimization(None, **cfg["pipeline"]["optimization"])
            maps = (np.full((8, H, W), 32, np.float32) if p2_method == "constant" else plugin.p2_maps(L))
            cuts = plugin.path_cuts(left)
            assert cuts.any() and not cuts.all()
            maps[cuts] = 0
            cost = orc.census_cost(L, R, 31, -30, 1, 5)
            want = orc.sgm_p2maps(cost, 8.0, maps, False, 26.0)
            np.testing.assert_array_equal(got, want, err_msg=f"{source} {p2_method}")
    # (2) a vertical border at column 77: every half as if optimised alone
    eng = runtime.get_engine()
    cost = orc.census_cost(L, R, 31, -30, 1, 5)
    cuts = optimization.AbstractOptimization(None, optimization_method="sgm", geometric_prior={"source": "segm"}).path_cuts(
        make_image(L, segm=(np.arange(W) >= 77).astype(np.int16)[None, :].repeat(H, 0)))
    maps = np.full((8, H, W), 32, np.float32)
    maps[cuts] = 0
    eng.set_images(L, R, 1)
    cv = eng.alloc_cv(31, -30)
    cv.from_host(cost)
    eng.sgm_p2maps(cv, 8.0, maps, False, 26.0)
    whole = cv.to_host()
    cv.free()
    for lo, hi in ((0, 77), (77, W)):
        eng.set_images(np.ascontiguousarray(L[:, lo:hi]), np.ascontiguousarray(R[:, lo:hi]), 1)
        half = eng.alloc_cv(31, -30)
        half.from_host(np.ascontiguousarray(cost[:, lo:hi]))
        eng.sgm(half, 8.0, 32.0, False, 26.0)
        np.testing.assert_array_equal(whole[:, lo:hi], half.to_host(), err_msg=f"columns {lo}:{hi}")
        half.free()


# ---- BASELINE.json configurations as they are worded, on the cones pair, whole volumes against the oracle ------------------------
def _cones():
    from PIL import Image

    d = os.path.join(ROOT, "tests", "golden", "cones")
    return (np.array(Image.open(os.path.join(d, "left.png"))).astype(np.float32),
            np.array(Image.open(os.path.join(d, "right.png"))).astype(np.float32),
            np.array(Image.open(os.path.join(d, "disp_left.tif"))).astype(np.float32))


@pytest.mark.parametrize("lazy", [True, False], ids=["lazy", "eager"])
def test_config2_as_stated_on_cones(oracle, lazy):
    """BASELINE configs[1]: cones, Census 5x5 + CBCA (intensity 30., distance 5: the defaults, cbca.py:46-47) + SGM (P1 = 8, P2 = 32,
    a_semi_global_matching.json:18-28) + WTA + vfit, d = [-60, 0]: the full 375 x 450 x 61 volume after the optimisation and the
    three maps equal the oracle's pipeline bit for bit, with the lazy representations and without."""
    from pandora_amd import runtime

    L, R, gt = _cones()
    cfg = {"pipeline": {"matching_cost": {"matching_cost_method": "census", "window_size": 5, "subpix": 1},
                        "aggregation": {"aggregation_method": "cbca", "cbca_intensity": 30.0, "cbca_distance": 5},
                        "optimization": {"optimization_method": "sgm", "overcounting": False,
                                         "penalty": {"penalty_method": "sgm_penalty", "P1": 8, "P2": 32, "p2_method": "constant"}},
                        "disparity": {"disparity_method": "wta", "invalid_disparity": "NaN"},
                        "refinement": {"refinement_method": "vfit"}}}
    dmin, dmax = -60, 0
    runtime.get_engine().set_lazy(lazy)
    try:
        machine, left_disp = run_machine(L, R, cfg, dmin, dmax)
        got_cv = machine.left_cv["cost_volume"].data
    finally:
        runtime.get_engine().set_lazy(True)
    mc_only = {"pipeline": {"matching_cost": cfg["pipeline"]["matching_cost"], "disparity": {"disparity_method": "wta"}}}
    cv0, _, _, _ = oracle_pipeline(oracle, L, R, mc_only, dmin, dmax)
    val0 = expected_validity(L, R, cfg, dmin, dmax, None, None, np.min(np.isnan(cv0), axis=2))
    ecv, edisp, eval_, eitp = oracle_pipeline(oracle, L, R, cfg, dmin, dmax, validity0=val0)
    assert got_cv.shape == (375, 450, 61)
    np.testing.assert_array_equal(got_cv, ecv)
    np.testing.assert_array_equal(left_disp["disparity_map"].data, edisp)
    np.testing.assert_array_equal(left_disp["validity_mask"].data, eval_)
    np.testing.assert_array_equal(left_disp["interpolated_coeff"].data, eitp)
    bad = (np.abs(np.nan_to_num(edisp, nan=1e4) + gt) > 1) & (gt != 0)
    assert bad.sum() / gt.size <= 0.20  # the reference's gate (tests/functional_tests/test_basic.py:120-156)


def test_config1_as_baseline_words_it(oracle):
    """BASELINE configs[0] as its string describes it: cones, SAD 5x5, d = [-64, 0], WTA (+ vfit) - beside the
    as-written a_local_block_matching.json run of test_quality_cones.py.  Whole volume and maps against the oracle, bit for bit
    (float32 summation order included)."""
    L, R, gt = _cones()
    cfg = {"pipeline": {"matching_cost": {"matching_cost_method": "sad", "window_size": 5, "subpix": 1},
                        "disparity": {"disparity_method": "wta", "invalid_disparity": "NaN"},
                        "refinement": {"refinement_method": "vfit"}}}
    dmin, dmax = -64, 0
    machine, left_disp = run_machine(L, R, cfg, dmin, dmax)
    got_cv = machine.left_cv["cost_volume"].data
    cv0, _, _, _ = oracle_pipeline(oracle, L, R, {"pipeline": {"matching_cost": cfg["pipeline"]["matching_cost"],
                                                              "disparity": {"disparity_method": "wta"}}}, dmin, dmax)
    val0 = expected_validity(L, R, cfg, dmin, dmax, None, None, np.min(np.isnan(cv0), axis=2))
    ecv, edisp, eval_, eitp = oracle_pipeline(oracle, L, R, cfg, dmin, dmax, validity0=val0)
    assert got_cv.shape == (375, 450, 65)
    np.testing.assert_array_equal(got_cv, ecv)
    np.testing.assert_array_equal(left_disp["disparity_map"].data, edisp)
    np.testing.assert_array_equal(left_disp["validity_mask"].data, eval_)
    np.testing.assert_array_equal(left_disp["interpolated_coeff"].data, eitp)
