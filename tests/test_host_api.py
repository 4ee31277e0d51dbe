"""CPU tests of the host layer: plugin registries / error behaviour (mirrors the reference's own
API tests), the state machine's sequencing rules, the validity-mask criteria against the reference's
golden masks, the C ABI's exported surface, and the 'no oracle / no CPU fallback in the product'
rules.  No GPU needed."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

import pandora_amd
from pandora_amd import (_lib, aggregation, cost_volume_confidence, criteria, disparity, matching_cost, multiscale, optimization,
                         refinement, validation)
from pandora_amd.dataset import DataArray, Dataset, make_image
from pandora_amd.matching_cost import ConfigError
from pandora_amd.state_machine import MachineError, PandoraMachine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- registries (reference: tests/test_plugins.py, test_matching_cost*.py window-size tests) --------------
def test_registries_hold_the_reference_short_names():
    assert set(matching_cost.AbstractMatchingCost.matching_cost_methods_avail) >= {"census", "sad", "ssd", "zncc"}
    assert "cbca" in aggregation.AbstractAggregation.aggreg_methods_avail
    assert "sgm" in optimization.AbstractOptimization.optimization_methods_avail
    assert "wta" in disparity.AbstractDisparity.disparity_methods_avail
    assert set(refinement.AbstractRefinement.subpixel_methods_avail) >= {"vfit", "quadratic"}
    assert set(validation.AbstractValidation.validation_methods_avail) >= {"cross_checking_accurate", "cross_checking_fast"}


@pytest.mark.parametrize("factory,key,msg", [
    (lambda **c: matching_cost.AbstractMatchingCost(**c), "matching_cost_method", "No matching cost method named {} supported"),
    (lambda **c: aggregation.AbstractAggregation(**c), "aggregation_method", "No aggregation method named {} supported"),
    (lambda **c: optimization.AbstractOptimization(None, **c), "optimization_method", "No optimization method named {} supported"),
    (lambda **c: disparity.AbstractDisparity(**c), "disparity_method", "No disparity method named {} supported"),
    (lambda **c: refinement.AbstractRefinement(**c), "refinement_method", "No refinement method named {} supported"),
    (lambda **c: validation.AbstractValidation(**c), "validation_method", "No validation method named {} supported"),
])
def test_unknown_method_raises_keyerror_like_the_reference(factory, key, msg):
    with pytest.raises(KeyError) as err:
        factory(**{key: "does_not_exist"})
    assert msg.format("does_not_exist") in str(err.value)


def test_register_subclass_adds_a_plugin():
    @matching_cost.AbstractMatchingCost.register_subclass("my_cost", "my_alias")
    class Mine(matching_cost.AbstractMatchingCost):
        def __init__(self, **cfg):
            self.cfg = cfg

        def compute_cost_volume(self, img_left, img_right, cost_volume):
            return cost_volume

    assert isinstance(matching_cost.AbstractMatchingCost(matching_cost_method="my_alias"), Mine)
    del matching_cost.AbstractMatchingCost.matching_cost_methods_avail["my_cost"]
    del matching_cost.AbstractMatchingCost.matching_cost_methods_avail["my_alias"]


@pytest.mark.parametrize("window_size", [3, 5, 7, 9, 11, 13])
def test_census_nominal_window_size(window_size):  # test_matching_cost_census.py:41-45
    m = matching_cost.AbstractMatchingCost(matching_cost_method="census", window_size=window_size)
    assert m.cfg["window_size"] == window_size


@pytest.mark.parametrize("window_size", [-5, -1, 0, 1, 2, 4, 6, 8, 14, 15])
def test_census_rejects_invalid_window_size(window_size):  # test_matching_cost_census.py:47-51
    with pytest.raises(ConfigError) as err:
        matching_cost.AbstractMatchingCost(matching_cost_method="census", window_size=window_size)
    assert "window_size" in str(err.value)


def test_defaults_and_checks():
    m = matching_cost.AbstractMatchingCost(matching_cost_method="zncc")
    assert m.cfg == {"matching_cost_method": "zncc", "window_size": 5, "subpix": 1, "band": None, "step": 1}
    with pytest.raises(ValueError):
        matching_cost.AbstractMatchingCost(matching_cost_method="sad", step=2)  # matching_cost.py:176-178
    with pytest.raises(ConfigError):
        matching_cost.AbstractMatchingCost(matching_cost_method="sad", subpix=3)
    a = aggregation.AbstractAggregation(aggregation_method="cbca")
    assert a.cfg["cbca_intensity"] == 30.0 and a.cfg["cbca_distance"] == 5  # cbca.py:46-47
    with pytest.raises(ConfigError):
        aggregation.AbstractAggregation(aggregation_method="cbca", cbca_intensity=-1.0)
    d = disparity.AbstractDisparity(disparity_method="wta")
    assert d.cfg["invalid_disparity"] == -9999  # disparity.py:356
    assert np.isnan(disparity.AbstractDisparity(disparity_method="wta", invalid_disparity="NaN").cfg["invalid_disparity"])
    o = optimization.AbstractOptimization(None, optimization_method="sgm")
    assert o.cfg["penalty"]["P1"] == 8 and o.cfg["penalty"]["P2"] == 32 and o.cfg["overcounting"] is False
    assert optimization.AbstractOptimization.margins_value == (40, 40, 40, 40)  # optimization.py:43
    with pytest.raises(ConfigError):
        optimization.AbstractOptimization(None, optimization_method="sgm", penalty={"P1": 8, "P2": 4})


def test_disparity_range_like_the_reference():  # matching_cost.py:409-427
    f = matching_cost.AbstractMatchingCost.get_disparity_range
    np.testing.assert_array_equal(f(-2, 2, 1), [-2, -1, 0, 1, 2])
    np.testing.assert_array_equal(f(-2, 2, 2), [-2, -1.5, -1, -0.5, 0, 0.5, 1, 1.5, 2])
    np.testing.assert_array_equal(f(0, 1, 4), [0, 0.25, 0.5, 0.75, 1])


# ---- state machine -----------------------------------------------------------------------------------------
PIPE = {"pipeline": {"matching_cost": {"matching_cost_method": "census"}, "optimization": {"optimization_method": "sgm"},
                     "disparity": {"disparity_method": "wta", "invalid_disparity": "NaN"},
                     "refinement": {"refinement_method": "vfit"}}}


def test_check_conf_fills_defaults_in_pipeline_order():
    cfg = PandoraMachine().check_conf(json.loads(json.dumps(PIPE)))
    assert list(cfg["pipeline"]) == ["matching_cost", "optimization", "disparity", "refinement"]
    assert cfg["pipeline"]["matching_cost"]["window_size"] == 5


def test_bad_sequencing_is_rejected():
    m = PandoraMachine()
    with pytest.raises(MachineError):  # disparity before any cost volume
        m.check_conf({"pipeline": {"disparity": {"disparity_method": "wta"}}})
    with pytest.raises(MachineError):  # aggregation after the disparity map exists
        m.check_conf({"pipeline": {"matching_cost": {"matching_cost_method": "sad"}, "disparity": {"disparity_method": "wta"},
                                   "aggregation": {"aggregation_method": "cbca"}}})
    with pytest.raises(MachineError) as err:  # out-of-scope step is named
        m.check_conf({"pipeline": {"matching_cost": {"matching_cost_method": "sad"}, "disparity": {"disparity_method": "wta"},
                                   "semantic_segmentation": {"segmentation_method": "ARNN"}}})
    assert "semantic_segmentation" in str(err.value)
    with pytest.raises(MachineError) as err:  # a filter nobody has: the plugin's KeyError, wrapped as the reference does
        m.check_conf({"pipeline": {"matching_cost": {"matching_cost_method": "sad"}, "disparity": {"disparity_method": "wta"},
                                   "filter": {"filter_method": "guided"}}})
    assert "No filter method named guided supported" in str(err.value)
    out = PandoraMachine().check_conf({"pipeline": {"matching_cost": {"matching_cost_method": "sad"}, "disparity": {"disparity_method": "wta"},
                                                    "filter": {"filter_method": "disparity_denoiser"}}})
    assert out["pipeline"]["filter"] == {"filter_method": "disparity_denoiser", "filter_size": 11, "sigma_euclidian": 4.0,
                                         "sigma_color": 100.0, "sigma_planar": 12.0, "sigma_grad": 1.5, "band": None}  # disparity_denoiser.py:56-62
    from pandora_amd import filter as flt
    from pandora_amd.matching_cost.matching_cost import ConfigError

    for bad in ({"filter_size": 0}, {"filter_size": 4}, {"sigma_color": 0.0}, {"sigma_planar": 3}, {"sigma_grad": -1.0}, {"band": 2},
                {"sigma": 1.0}):  # test_disparity_denoiser.py:134-147 + the reference's odd-size assertion
        with pytest.raises(ConfigError):
            flt.AbstractFilter(cfg={"filter_method": "disparity_denoiser", **bad})
    assert flt.AbstractFilter(cfg={"filter_method": "disparity_denoiser", "filter_size": 5, "band": "red"}).cfg["band"] == "red"
    out = PandoraMachine().check_conf({"pipeline": {"matching_cost": {"matching_cost_method": "sad"},
                                                    "disparity": {"disparity_method": "wta"}, "filter": {"filter_method": "bilateral"}}})
    assert out["pipeline"]["filter"]["sigma_color"] == 2.0 and out["pipeline"]["filter"]["sigma_space"] == 6.0  # bilateral.py:47-48
    out = PandoraMachine().check_conf({"pipeline": {"matching_cost": {"matching_cost_method": "sad"},
                                                    "disparity": {"disparity_method": "wta"}, "filter": {"filter_method": "median"}}})
    assert out["pipeline"]["filter"]["filter_size"] == 3  # median.py:50


def test_validation_configuration_like_the_reference():  # test_validation.py:78-102, validation.py:196-217
    v = validation.AbstractValidation(validation_method="cross_checking_fast")
    assert isinstance(v, validation.CrossCheckingAccurate) and v.cfg["cross_checking_threshold"] == 1.0
    assert validation.AbstractValidation(validation_method="cross_checking_accurate", cross_checking_threshold=0).cfg[
        "cross_checking_threshold"] == 0
    with pytest.raises(KeyError):
        validation.AbstractValidation()  # the method is mandatory
    with pytest.raises(ConfigError):
        validation.AbstractValidation(validation_method="cross_checking_fast", cross_checking_threshold="1")
    with pytest.raises(ConfigError):
        validation.AbstractValidation(validation_method="cross_checking_fast", interpolated_disparity="linear")


def test_validation_step_is_sequenced_after_the_disparity_map():
    pipe = json.loads(json.dumps(PIPE))
    pipe["pipeline"]["validation"] = {"validation_method": "cross_checking_fast"}
    m = PandoraMachine()
    out = m.check_conf(pipe)
    assert out["pipeline"]["validation"]["cross_checking_threshold"] == 1.0 and m.right_disp_map == "cross_checking_fast"
    with pytest.raises(MachineError):  # no disparity map yet
        PandoraMachine().check_conf({"pipeline": {"matching_cost": {"matching_cost_method": "sad"},
                                                  "validation": {"validation_method": "cross_checking_fast"}}})
    pipe["pipeline"]["validation"]["interpolated_disparity"] = "sgm"  # state_machine.py:907-908: the plugin is instantiated
    assert PandoraMachine().check_conf(pipe)["pipeline"]["validation"]["interpolated_disparity"] == "sgm"
    assert isinstance(validation.AbstractInterpolation(interpolated_disparity="mc-cnn"), validation.McCnnInterpolation)
    with pytest.raises(KeyError):
        validation.AbstractInterpolation(interpolated_disparity="linear")
    # disparity_source consistency (state_machine.py:912-918)
    left = make_image(np.zeros((4, 6)), disparity=[-2, 1])
    right = make_image(np.zeros((4, 6)), disparity=[-1, 3])
    left.attrs["disparity_source"], right.attrs["disparity_source"] = [-2, 1], [-1, 3]
    del pipe["pipeline"]["validation"]["interpolated_disparity"]
    with pytest.raises(MachineError, match="A problem occurs during Pandora checking. Be sure of your sequencing"):  # test_config.py:266
        PandoraMachine().check_conf(pipe, left, right)


def test_allocate_confidence_map_appends_an_indicator():  # cost_volume_confidence.py:141-246
    ds = Dataset({"disparity_map": (("row", "col"), np.zeros((2, 3), np.float32))}, coords={"row": [0, 1], "col": [0, 1, 2]})
    a = np.arange(6, dtype=np.float32).reshape(2, 3)
    ds, _ = validation.allocate_confidence_map("left_right_consistency", a, ds, None)
    ds, _ = validation.allocate_confidence_map("ambiguity.disp_min", a + 1, ds, None)
    assert list(ds.coords["indicator"]) == ["confidence_from_left_right_consistency", "ambiguity.disp_min"]
    np.testing.assert_array_equal(ds["confidence_measure"].data[:, :, 0], a)
    np.testing.assert_array_equal(ds["confidence_measure"].data[:, :, 1], a + 1)


def test_multiscale_configuration_like_the_reference():  # fixed_zoom_pyramid.py:44-98, test_multiscale.py:245-262
    left, right = make_image(np.zeros((8, 8)), disparity=[-30, 0]), make_image(np.zeros((8, 8)), disparity=[0, 30])
    m = multiscale.AbstractMultiscale(left, right, multiscale_method="fixed_zoom_pyramid")
    assert (m.cfg["num_scales"], m.cfg["scale_factor"], m.cfg["marge"]) == (2, 2, 1)
    with pytest.raises(KeyError) as err:
        multiscale.AbstractMultiscale(left, right, multiscale_method="nope")
    assert "No multiscale method named nope supported" in str(err.value)
    with pytest.raises(ConfigError):
        multiscale.AbstractMultiscale(left, right, multiscale_method="fixed_zoom_pyramid", num_scales=1)
    grid = make_image(np.zeros((8, 8)), disparity_grids=(np.full((8, 8), -3), np.full((8, 8), 2)))
    with pytest.raises(TypeError, match="Multiscale processing does not accept input disparity grids."):
        multiscale.AbstractMultiscale(grid, right, multiscale_method="fixed_zoom_pyramid")
    pipe = json.loads(json.dumps(PIPE))
    pipe["pipeline"]["multiscale"] = {"multiscale_method": "fixed_zoom_pyramid", "num_scales": 3}
    out = PandoraMachine().check_conf(pipe, left, right)
    assert out["pipeline"]["multiscale"]["scale_factor"] == 2
    assert multiscale.read_multiscale_params(left, right, out) == (3, 2)
    # mask_invalid_disparities (multiscale.py:129-153, test_multiscale.py:139-243)
    ds = Dataset({"disparity_map": (("row", "col"), np.arange(6, dtype=np.float32).reshape(2, 3)),
                  "validity_mask": (("row", "col"), np.array([[1, 4, 0], [8, 512, 0]]))})
    np.testing.assert_array_equal(np.isnan(multiscale.AbstractMultiscale.mask_invalid_disparities(ds)),
                                  [[True, False, False], [False, True, False]])


def test_pyramid_shapes_and_order():
    """img_tools.py:479-572: coarsest first, the full-resolution dataset is the original object, disparities travel as
    int64 grids, attrs are shared."""
    rng = np.random.default_rng(0)
    left = make_image(rng.random((37, 50)).astype(np.float32) * 255, disparity=[-8, 2])
    right = make_image(rng.random((37, 50)).astype(np.float32) * 255)
    pl, pr = multiscale.prepare_pyramid(left, right, 3, 2)
    assert [p["im"].data.shape for p in pl] == [(10, 13), (19, 25), (37, 50)] and pl[-1] is left and pr[-1] is right
    assert pl[0]["disparity"].data.dtype == np.int64 and pl[0]["disparity"].data.shape == (2, 10, 13)
    assert pl[0].attrs is left.attrs and "disparity" not in pr[0].data_vars
    # a 2x reduction of a constant image is the constant (gaussian + bilinear resize preserve it)
    flat = multiscale.get_pyramids(np.full((16, 16), 7.0, np.float32), 2, 2)
    np.testing.assert_allclose(flat[1], 7.0, rtol=1e-6)


def test_confidence_configuration_like_the_reference():  # ambiguity.py:56-104, std_intensity.py:50-72
    a = cost_volume_confidence.AbstractCostVolumeConfidence(confidence_method="ambiguity")
    assert (a.cfg["eta_max"], a.cfg["eta_step"], a.cfg["normalization"]) == (0.7, 0.01, True) and a._nbr_etas == 70
    assert cost_volume_confidence.AbstractCostVolumeConfidence(confidence_method="std_intensity").cfg["indicator"] == ""
    with pytest.raises(KeyError) as err:
        cost_volume_confidence.AbstractCostVolumeConfidence(confidence_method="mc_cnn_confidence")
    assert "No confidence method named mc_cnn_confidence supported" in str(err.value)
    r = cost_volume_confidence.AbstractCostVolumeConfidence(confidence_method="risk", eta_max=0.5, eta_step=0.1)  # risk.py:56-104
    assert r._nbr_etas == 5 and r._indicator_max == "risk_max" and r._indicator_disp_inf == "disp_inf_from_risk"
    ib = cost_volume_confidence.AbstractCostVolumeConfidence(confidence_method="interval_bounds")  # interval_bounds.py:54-120
    assert (ib.cfg["possibility_threshold"], ib.cfg["regularization"], ib.cfg["ambiguity_kernel_size"], ib.cfg["vertical_depth"],
            ib.cfg["quantile_regularization"]) == (0.9, False, 5, 0, 1.0)
    for bad in ({"possibility_threshold": 1.5}, {"ambiguity_kernel_size": 4}, {"vertical_depth": -1}, {"regularization": 1}):
        with pytest.raises(ConfigError):
            cost_volume_confidence.AbstractCostVolumeConfidence(confidence_method="interval_bounds", **bad)
    with pytest.raises(ConfigError):
        cost_volume_confidence.AbstractCostVolumeConfidence(confidence_method="ambiguity", eta_max=1.5)
    pipe = {"pipeline": {"matching_cost": {"matching_cost_method": "zncc"},
                         "cost_volume_confidence": {"confidence_method": "std_intensity"},
                         "cost_volume_confidence.amb": {"confidence_method": "ambiguity", "eta_max": 0.5},
                         "disparity": {"disparity_method": "wta"}}}
    out = PandoraMachine().check_conf(pipe)
    assert out["pipeline"]["cost_volume_confidence.amb"]["eta_step"] == 0.01
    # normalisation helpers (ambiguity.py:168-184, cost_volume_confidence.py:114-138)
    amb = np.array([[0.0, 10.0], [20.0, 1000.0]], np.float32)
    n = a.normalize_with_percentile(amb)
    assert n.min() == 0.0 and n.max() == 1.0
    ds = Dataset(attrs={"global_disparity": [-10, 10]})
    np.testing.assert_allclose(a.normalize_with_extremum(amb, ds, nbr_etas=70, subpix=2), amb / (20 * 70 * 2))
    # std_intensity's raster: float64 cumulative-sum box statistics (img_tools.py:834-952)
    from pandora_amd.cost_volume_confidence.std_intensity import compute_std_raster
    im = np.random.default_rng(1).random((9, 11)).astype(np.float32) * 100
    ref = np.array([[im[r:r + 3, c:c + 3].astype(np.float64).std() for c in range(9)] for r in range(7)])
    np.testing.assert_allclose(compute_std_raster(im, 3), ref, rtol=1e-6, atol=1e-4)


def test_repeated_steps_use_the_key_prefix():  # state_machine.py:706-717 ("refinement.again" -> refinement)
    cfg = {"pipeline": {"matching_cost": {"matching_cost_method": "sad"}, "disparity": {"disparity_method": "wta"},
                        "refinement": {"refinement_method": "vfit"}, "refinement.again": {"refinement_method": "quadratic"}}}
    out = PandoraMachine().check_conf(cfg)
    assert out["pipeline"]["refinement.again"]["refinement_method"] == "quadratic"


# ---- criteria: validity mask golden masks of the reference --------------------------------------------------
with open(os.path.join(ROOT, "tests", "golden", "validity_mask_cases.json")) as f:
    VM_CASES = json.load(f)["cases"]


@pytest.mark.parametrize("case", VM_CASES, ids=lambda c: c["id"])
def test_validity_mask_reference_goldens(oracle, case):
    """tests/test_criteria.py::test_validity_mask: validity_mask + compute_cost_volume + cv_masked.
    The all-NaN-pixel reduction (GPU in the product) is supplied by the oracle here."""
    L = np.array(case["left_data"], np.float32)
    R = np.array(case["right_data"], np.float32)
    left = make_image(L, disparity=case["disparity"], msk=np.array(case["left_msk"]), valid_pixels=case["left_valid"],
                      no_data_mask=case["left_nodata"])
    right = make_image(R, msk=np.array(case["right_msk"]), valid_pixels=case["right_valid"], no_data_mask=case["right_nodata"])
    m = matching_cost.AbstractMatchingCost(matching_cost_method="sad", window_size=case["window_size"], subpix=1)
    cv = m.allocate_cost_volume(left, (left["disparity"].sel(band_disp="min"), left["disparity"].sel(band_disp="max")))
    cv = criteria.validity_mask(left, right, cv)
    dmin, dmax = case["disparity"]
    vol = oracle.sad_ssd(L, R, dmax - dmin + 1, dmin, 1, case["window_size"], False)
    assert case["left_valid"] == case["right_valid"] and case["left_nodata"] == case["right_nodata"]
    oracle.cv_masked(vol, dmin, 1, case["window_size"], mskL=np.array(case["left_msk"]), mskR=np.array(case["right_msk"]),
                     valid=case["left_valid"], nodata=case["left_nodata"])
    criteria.mask_invalid_variable_disparity_range(cv, np.min(np.isnan(vol), axis=2))
    if cv.attrs["offset_row_col"] > 0:
        criteria.mask_border(cv)
    np.testing.assert_array_equal(cv["validity_mask"].data, np.array(case["gt_mask"]))


# ---- the C ABI surface ------------------------------------------------------------------------------------
def test_library_exports_every_declared_symbol():
    handle = ctypes.CDLL(_lib.LIB_PATH)
    declared = _lib.header_symbols()
    assert len(declared) >= 30
    for sym in declared:
        assert hasattr(handle, sym), f"{sym} declared in include/pandora_amd.h but not exported"
    assert set(declared) == set(_lib.SIGNATURES), "ctypes signature table out of sync with the header"


def test_header_cites_the_reference_interface():
    text = open(_lib.HEADER_PATH).read()
    for needle in ("census.cpp:97-180", "aggregation.cpp:224-321", "refinement.cpp:28-99", "disparity.py:399-516",
                   "optimization.py:104-123", "matching_cost.py:770-872"):
        assert needle in text


def test_no_gpu_means_loud_failure_not_fallback():
    if _lib.lib().pmx_device_count() > 0:
        pytest.skip("a GPU is visible")
    from pandora_amd.engine import Engine

    with pytest.raises(RuntimeError) as err:
        Engine(0)
    assert "no CPU fallback" in str(err.value)
    left = make_image(np.zeros((8, 8)), disparity=[-1, 1])
    m = matching_cost.AbstractMatchingCost(matching_cost_method="census", window_size=3)
    cv = m.allocate_cost_volume(left, (left["disparity"].sel(band_disp="min"), left["disparity"].sel(band_disp="max")))
    with pytest.raises(RuntimeError):
        m.compute_cost_volume(left, left, cv)


def test_product_never_touches_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use oracle/."""
    pkg = os.path.join(ROOT, "pandora_amd")
    bad = []
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, fn), errors="ignore").read()
                if re.search(r'(import\s+oracle|from\s+oracle\b|liboracle|include\s*[<"][^\n]*oracle)', text):
                    bad.append(os.path.join(dirpath, fn))
    assert not bad, f"product files reference the oracle: {bad}"
    assert "pandora_amd" in pandora_amd.__name__


# test_matching_cost.py:110-195 (census / sad / ssd) and test_matching_cost_zncc.py:315-397 (zncc), 6-column images, window 3
_PI = [(0, (0, 6), (0, 6)), (-2, (2, 6), (0, 4)), (2, (0, 4), (2, 6)), (-2.5, (3, 6), (0, 4)), (2.5, (0, 3), (2, 6)),
       (7, (6, 6), (6, 6)), (-7, (6, 6), (6, 6))]


@pytest.mark.parametrize("method", ["census", "sad", "ssd", "zncc"])
def test_point_interval_like_the_reference(method):
    img = make_image(np.zeros((5, 6)), disparity=[-2, 2])
    m = matching_cost.AbstractMatchingCost(matching_cost_method=method, window_size=3, subpix=1)
    last = [(5, (6, 6), (6, 6)), (-5, (6, 6), (6, 6))] if method == "zncc" else [(5, (0, 1), (5, 6)), (-5, (5, 6), (0, 1))]
    for disp, p, q in _PI + last:
        assert m.point_interval(img, img, disp) == (p, q), (method, disp)


def test_disparity_interval_helpers():  # tests/test_disparity.py:642-740
    from pandora_amd.dataset import Dataset
    from pandora_amd.disparity import disparity as dmod

    cv = Dataset({}, coords={"row": np.arange(2), "col": np.arange(3), "disp": np.array([-2.0, -1.5, -1.0, -0.5, 0.0, 0.5, 1.0])})
    interval = dmod.extract_disparity_interval_from_cost_volume(cv)
    np.testing.assert_array_equal(interval.data, [-2.0, 1.0])
    assert list(interval.coords["disparity"]) == ["min", "max"]
    disp = Dataset({"disparity_interval": interval})
    assert dmod.extract_interval_from_disparity_map(disp) == (-2, 1)
    np.testing.assert_array_equal(dmod.extract_disparity_range_from_disparity_map(disp), [-2, -1, 0, 1])


# ---- image datasets from files (tests/test_pandora_image.py:369-461) ---------------------------------------
def test_create_dataset_from_inputs_reference_vectors():
    from pandora_amd import check_datasets, img_tools

    gold = os.path.join(ROOT, "tests", "golden", "image")
    mask_gt = np.array([[1, 0, 2, 2, 1], [0, 0, 0, 0, 2], [1, 1, 0, 0, 2], [0, 0, 2, 0, 1]])
    left_img = np.array([[-9999.0, 1.0, 2.0, 3.0, -9999.0], [5.0, 6.0, 7.0, 8.0, 9.0], [-9999.0, -9999.0, 23.0, 5.0, 6.0],
                         [12.0, 5.0, 6.0, 3.0, -9999.0]], np.float32)
    for name, nodata in (("left_img.tif", -9999), ("left_img_nan.tif", np.nan)):
        ds = img_tools.create_dataset_from_inputs({"img": os.path.join(gold, name), "nodata": nodata,
                                                   "mask": os.path.join(gold, "mask_left.tif"), "disp": [-60, 0]})
        np.testing.assert_array_equal(ds["msk"].data, mask_gt)
        np.testing.assert_array_equal(ds["im"].data, left_img)
        assert ds["msk"].data.dtype == np.int16 and ds["im"].data.dtype == np.float32
        assert ds.attrs["no_data_img"] == -9999 and ds.attrs["disparity_source"] == [-60, 0]
        assert ds["disparity"].data.shape == (2, 4, 5) and (ds["disparity"].data[0] == -60).all() and (ds["disparity"].data[1] == 0).all()
        check_datasets(ds, ds)
    # no mask and no no-data pixel: no msk at all (img_tools.py:283-285); inf as no-data (test_pandora_image.py:631-668)
    ds = img_tools.create_dataset_from_inputs({"img": os.path.join(gold, "left_img.tif"), "nodata": 12345, "disp": [0, 2]})
    assert "msk" not in ds.data_vars
    with pytest.raises(FileNotFoundError):
        img_tools.create_dataset_from_inputs({"img": os.path.join(gold, "left_img.tif"), "nodata": 0, "classif": "x.tif"})
    with pytest.raises(AttributeError):
        check_datasets(img_tools.create_dataset_from_inputs({"img": os.path.join(gold, "left_img.tif"), "nodata": 0}), ds)


def test_save_results_writes_the_reference_tree(tmp_path):
    from PIL import Image

    from pandora_amd import common
    from pandora_amd.dataset import Dataset

    rng = np.random.default_rng(0)
    disp = rng.normal(size=(6, 7)).astype(np.float32)
    disp[0, 0] = np.nan
    conf = rng.random((6, 7, 2)).astype(np.float32)
    left = Dataset({"disparity_map": (("row", "col"), disp), "validity_mask": (("row", "col"), rng.integers(0, 4096, (6, 7))),
                    "confidence_measure": (("row", "col", "indicator"), conf)},
                   coords={"row": np.arange(6), "col": np.arange(7), "indicator": ["a", "b"]}, attrs={"crs": None, "transform": None})
    common.save_results(left, Dataset(), str(tmp_path))
    common.save_config(str(tmp_path), {"pipeline": {"x": {"y": np.int64(3)}}})
    assert sorted(os.listdir(tmp_path)) == ["cfg", "left_confidence_measure.tif", "left_disparity.tif", "left_validity_mask.tif"]
    np.testing.assert_array_equal(np.array(Image.open(tmp_path / "left_disparity.tif")), disp)
    vm = np.array(Image.open(tmp_path / "left_validity_mask.tif"))
    assert vm.dtype == np.uint16
    np.testing.assert_array_equal(vm, left["validity_mask"].data)
    from pandora_amd.tiff_reader import read_tiff

    bands, names = read_tiff(str(tmp_path / "left_confidence_measure.tif"))
    assert bands.shape == (2, 6, 7) and names == ["a", "b"] and bands.dtype == np.float32
    np.testing.assert_array_equal(bands, np.moveaxis(conf, 2, 0))
    assert json.load(open(tmp_path / "cfg" / "config.json"))["pipeline"]["x"]["y"] == 3


# ---- multiband images and the matching-cost "band" parameter ------------------------------------------------------------------
def _two_band_pair():
    data_l = np.zeros((2, 4, 4))
    data_l[0] = [[1, 1, 1, 3], [1, 3, 2, 5], [2, 1, 0, 1], [1, 5, 4, 3]]
    data_l[1] = [[2, 3, 4, 6], [8, 7, 0, 4], [4, 9, 1, 5], [6, 5, 2, 1]]
    data_r = np.zeros((2, 4, 4))
    data_r[0] = [[5, 1, 2, 3], [1, 3, 0, 2], [2, 3, 5, 0], [1, 6, 7, 5]]
    data_r[1] = [[6, 5, 2, 7], [8, 7, 6, 5], [5, 2, 3, 6], [0, 3, 4, 7]]
    return (make_image(data_l, disparity=[-1, 1], band_names=["red", "green"]), make_image(data_r, band_names=["red", "green"]))


def test_band_errors_like_the_reference():
    """test_matching_cost_census.py:226-376 (test_check_band_census, test_instantiate_band_with_monoband), same messages for
    every measure; state_machine.py:1042-1072 check_band_pipeline."""
    left, right = _two_band_pair()
    grids = (left["disparity"].sel(band_disp="min"), left["disparity"].sel(band_disp="max"))
    for method in ("census", "sad", "zncc"):
        m = matching_cost.AbstractMatchingCost(matching_cost_method=method, window_size=3, subpix=1, band="blue")
        with pytest.raises(AttributeError, match="Wrong band instantiate : blue not in img_left or img_right"):
            m.compute_cost_volume(left, right, m.allocate_cost_volume(left, grids))
        m = matching_cost.AbstractMatchingCost(matching_cost_method=method, window_size=3, subpix=1)
        with pytest.raises(AttributeError, match="Band must be instantiated in matching cost step"):
            m.compute_cost_volume(left, right, m.allocate_cost_volume(left, grids))
    mono = make_image(np.zeros((4, 4)), disparity=[-1, 1])
    m = matching_cost.AbstractMatchingCost(matching_cost_method="census", window_size=3, subpix=1, band="red")
    with pytest.raises(AttributeError, match="Right dataset is monoband: red band cannot be selected"):
        m.compute_cost_volume(left, mono, m.allocate_cost_volume(left, grids))
    with pytest.raises(AttributeError, match="Left dataset is monoband: red band cannot be selected"):
        m.compute_cost_volume(mono, right, m.allocate_cost_volume(mono, grids))
    pipe = {"pipeline": {"matching_cost": {"matching_cost_method": "census", "window_size": 3}, "disparity": {"disparity_method": "wta"}}}
    with pytest.raises(MachineError):  # "Missing band instantiate on census step : input image is multiband"
        PandoraMachine().check_conf(json.loads(json.dumps(pipe)), left, right)
    pipe["pipeline"]["matching_cost"]["band"] = "green"
    assert PandoraMachine().check_conf(json.loads(json.dumps(pipe)), left, right)["pipeline"]["matching_cost"]["band"] == "green"
    pipe["pipeline"]["matching_cost"]["band"] = "blue"
    with pytest.raises(MachineError):
        PandoraMachine().check_conf(json.loads(json.dumps(pipe)), left, right)


def test_sgm_use_confidence_configuration():  # plugin_libsgm.rst:38-47, :88-209
    from pandora_amd import optimization

    o = optimization.AbstractOptimization(None, optimization_method="sgm", use_confidence="cost_volume_confidence.before")
    assert o.cfg["use_confidence"] == "cost_volume_confidence.before"
    assert optimization.AbstractOptimization(None, optimization_method="sgm", use_confidence=False).cfg["use_confidence"] is False
    with pytest.raises(ConfigError):
        optimization.AbstractOptimization(None, optimization_method="sgm", use_confidence="ambiguity")
    with pytest.raises(ConfigError):
        optimization.AbstractOptimization(None, optimization_method="sgm", geometric_prior={"source": "superpixels"})


def test_tiff_reader_and_multiband_inputs(tmp_path):
    """pandora_amd/tiff_reader.py on the reference's tiny two-band grid (tests/pandora/tiny_left_disparity_grid.tif), on a
    multi-sample TIFF with GDAL band descriptions, and - when the reference tree is here - on its multiband / grid rasters;
    create_dataset_from_inputs with a multiband image and with a disparity-grid file (img_tools.py:388-398, :124-125)."""
    from PIL import Image

    from pandora_amd import img_tools
    from pandora_amd.tiff_reader import read_tiff

    gold = os.path.join(ROOT, "tests", "golden", "image")
    grid, names = read_tiff(os.path.join(gold, "tiny_left_disparity_grid.tif"))
    assert grid.shape == (2, 4, 4) and names == ["min", "max"] and (grid[0] == -27).all() and (grid[1] == -7).all()
    one, names = read_tiff(os.path.join(gold, "left_img.tif"))
    np.testing.assert_array_equal(one, np.array(Image.open(os.path.join(gold, "left_img.tif"))))
    assert names is None
    rgb = np.random.default_rng(0).integers(0, 255, (4, 4, 3)).astype(np.uint8)
    xml = ('<GDALMetadata>\n  <Item name="DESCRIPTION" sample="0" role="description">r</Item>\n  <Item name="DESCRIPTION" sample="1" '
           'role="description">g</Item>\n  <Item name="DESCRIPTION" sample="2" role="description">b</Item>\n</GDALMetadata>\n')
    Image.fromarray(rgb, mode="RGB").save(tmp_path / "rgb.tif", tiffinfo={42112: xml})
    ds = img_tools.create_dataset_from_inputs({"img": str(tmp_path / "rgb.tif"), "nodata": -9999,
                                               "disp": os.path.join(gold, "tiny_left_disparity_grid.tif")})
    assert ds["im"].dims == ("band_im", "row", "col") and list(ds.coords["band_im"]) == ["r", "g", "b"]
    np.testing.assert_array_equal(ds["im"].data, np.moveaxis(rgb, 2, 0).astype(np.float32))
    assert ds.attrs["disparity_source"].endswith("tiny_left_disparity_grid.tif") and ds["disparity"].data.dtype == np.float32
    np.testing.assert_array_equal(ds["disparity"].data, grid)
    ref = "/root/reference/tests/pandora"
    if os.path.exists(ref):  # build container only
        a, names = read_tiff(os.path.join(ref, "left_rgb.tif"))
        assert a.shape == (3, 375, 450) and names == ["red", "green", "blue"] and a.dtype == np.float32
        d, names = read_tiff(os.path.join(ref, "left_disparity_grid.tif"))
        assert d.shape == (2, 375, 450) and names == ["min", "max"] and d.min() == -65 and d.max() == 10
        for f in ("disp_left.tif", "disp_min_grid.tif", "mask_from_occlusion_left.tif"):
            np.testing.assert_array_equal(read_tiff(os.path.join(ref, f))[0], np.array(Image.open(os.path.join(ref, f))))


def test_tiff_writer_roundtrip(tmp_path):
    from PIL import Image

    from pandora_amd.tiff_reader import read_tiff, write_tiff

    rng = np.random.default_rng(0)
    for dt in (np.float32, np.uint16, np.uint8, np.int16, np.float64):
        a = (rng.random((3, 5, 7)) * 200).astype(dt)
        write_tiff(str(tmp_path / "m.tif"), a, ["x", "y y", "z"])
        b, names = read_tiff(str(tmp_path / "m.tif"))
        assert np.array_equal(a, b) and names == ["x", "y y", "z"] and b.dtype == dt
        write_tiff(str(tmp_path / "s.tif"), a[0])
        b, names = read_tiff(str(tmp_path / "s.tif"))
        assert np.array_equal(a[0], b) and names is None
        if dt in (np.float32, np.uint16, np.uint8):  # what Pillow decodes: an independent reader agrees
            np.testing.assert_array_equal(np.array(Image.open(tmp_path / "s.tif")), a[0])


def test_margins_like_the_reference():
    """margins/margins.py, descriptors.py; tests/test_pandora.py:150-210 (the margins written next to the configuration):
    matching_cost HalfWindowMargins, optimization UniformMargins(40), filters non-cumulative, the rest null."""
    from pandora_amd.margins import GlobalMargins, Margins, max_margins

    assert Margins(1, 2, 3, 4) + Margins(1, 1, 1, 1) == Margins(2, 3, 4, 5)
    assert max_margins([Margins(1, 5, 2, 0), Margins(3, 1, 2, 7)]) == Margins(3, 5, 2, 7)
    with pytest.raises(ValueError):
        Margins(-1, 0, 0, 0)
    g = GlobalMargins()
    g.add_cumulative("a", Margins(1, 1, 1, 1))
    with pytest.raises(KeyError):
        g.add_non_cumulative("a", Margins(0, 0, 0, 0))
    left, right = make_image(np.zeros((30, 40)), disparity=[-3, 0]), make_image(np.zeros((30, 40)), disparity=[0, 3])
    m = PandoraMachine()
    m.check_conf({"pipeline": {"matching_cost": {"matching_cost_method": "zncc", "window_size": 5, "subpix": 2},
                               "disparity": {"disparity_method": "wta", "invalid_disparity": -9999},
                               "refinement": {"refinement_method": "vfit"},
                               "filter": {"filter_method": "median", "filter_size": 3}}}, left, right)
    assert m.margins.to_dict() == {  # tests/test_pandora.py:198-209
        "cumulative margins": {"matching_cost": {"left": 2, "up": 2, "right": 2, "down": 2},
                               "disparity": {"left": 0, "up": 0, "right": 0, "down": 0},
                               "refinement": {"left": 0, "up": 0, "right": 0, "down": 0}},
        "non-cumulative margins": {"filter": {"left": 3, "up": 3, "right": 3, "down": 3}},
        "global margins": {"left": 3, "up": 3, "right": 3, "down": 3}}
    m.check_conf({"pipeline": {"matching_cost": {"matching_cost_method": "census", "window_size": 5},
                               "optimization": {"optimization_method": "sgm"},
                               "disparity": {"disparity_method": "wta"},
                               "filter": {"filter_method": "bilateral", "sigma_space": 6.0}}}, left, right)
    assert m.margins.global_margins == Margins(42, 42, 42, 42) and m.margins.get("filter") == Margins(19, 19, 19, 19)


def test_resident_pair_fingerprint_sees_every_pixel():
    """runtime._sample covers the whole buffer: an in-place edit of ONE pixel, anywhere (e.g. a column a strided sample with a
    power-of-two width never visits), changes the key, so ensure_pair uploads the pair again."""
    from pandora_amd import runtime

    a = np.zeros((2048, 2048), np.float32)
    before = runtime._sample(a)
    a[1234, 777] = 1.0  # 777 is not a multiple of 64
    assert runtime._sample(a) != before
    m = np.zeros((64, 64), np.int16)
    before = runtime._sample(m)
    m[63, 63] = 1
    assert runtime._sample(m) != before


def test_configuration_limits_of_the_device_kernels_are_refused_at_check_conf():
    """What the kernels cannot do is refused when the configuration is checked, not in the middle of a pipeline."""
    from pandora_amd import cost_volume_confidence as cvc
    from pandora_amd import filter as flt
    from pandora_amd.matching_cost.matching_cost import ConfigError

    with pytest.raises(ConfigError, match="1024 etas"):
        cvc.AbstractCostVolumeConfidence(**{"confidence_method": "risk", "eta_max": 0.7, "eta_step": 0.0005})
    with pytest.raises(ConfigError, match="1024 etas"):
        cvc.AbstractCostVolumeConfidence(**{"confidence_method": "ambiguity", "eta_max": 0.9, "eta_step": 0.0005})
    with pytest.raises(ConfigError, match="up to 15"):
        flt.AbstractFilter(cfg={"filter_method": "median", "filter_size": 17})


# ---- SURVEY 8f N5 remainder: ROI windows and georeferencing passthrough --------------------------------------------------
_ROI_SHAPE = (8, 11)  # tests/test_pandora_image.py:259-277


@pytest.mark.parametrize("roi,expected", [
    ({"col": {"first": 3, "last": 5}, "row": {"first": 3, "last": 5}, "margins": [2, 2, 2, 2]}, (1, 1, 7, 7)),    # :279-292
    ({"col": {"first": 0, "last": 2}, "row": {"first": 3, "last": 5}, "margins": [2, 2, 2, 2]}, (0, 1, 5, 7)),    # :294-326 left
    ({"col": {"first": 10, "last": 12}, "row": {"first": 3, "last": 5}, "margins": [2, 2, 2, 2]}, (8, 1, 3, 7)),  # right
    ({"col": {"first": 3, "last": 5}, "row": {"first": -1, "last": 5}, "margins": [2, 2, 2, 2]}, (1, 0, 7, 8)),   # up
    ({"col": {"first": 3, "last": 5}, "row": {"first": 9, "last": 11}, "margins": [2, 2, 2, 2]}, (1, 7, 7, 1)),   # down
])
def test_get_window_reference_vectors(roi, expected):
    from pandora_amd.img_tools import get_window

    assert tuple(get_window(roi, _ROI_SHAPE[1], _ROI_SHAPE[0])) == expected


@pytest.mark.parametrize("roi", [
    {"col": {"first": -10, "last": -12}, "row": {"first": 3, "last": 5}, "margins": [2, 2, 2, 2]},   # tests/test_pandora_image.py:328-357
    {"col": {"first": 100, "last": 120}, "row": {"first": 3, "last": 5}, "margins": [2, 2, 2, 2]},
    {"col": {"first": 3, "last": 5}, "row": {"first": -6, "last": -5}, "margins": [2, 2, 2, 2]},
    {"col": {"first": 3, "last": 5}, "row": {"first": 11, "last": 111}, "margins": [2, 2, 2, 2]},
])
def test_get_window_outside_the_image(roi):
    from pandora_amd.img_tools import get_window

    with pytest.raises(ValueError, match="Roi specified is outside the image"):
        get_window(roi, _ROI_SHAPE[1], _ROI_SHAPE[0])


@pytest.mark.parametrize("step,margin,coords,truth", [  # tests/test_matching_cost/test_matching_cost.py:198-236
    (2, 4, np.arange(0, 20, 2), np.arange(0, 20, 2)),
    (3, 2, np.arange(8, 24, 3), np.arange(10, 24, 3)),
    (2, 3, np.arange(7, 24, 2), np.arange(8, 24, 2)),
])
def test_get_coordinates_reference_vectors(step, margin, coords, truth):
    np.testing.assert_array_equal(matching_cost.AbstractMatchingCost.get_coordinates(margin, coords, step), truth)


def test_roi_dataset_and_georeferencing_passthrough(tmp_path):
    """create_dataset_from_inputs with a ROI (tests/test_pandora_image.py:669-699: the window's pixels, coordinates starting at its
    offset; mask and disparity-grid files cut the same way), and a GeoTIFF's georeferencing tags travelling from the input image
    through attrs["crs"] / attrs["transform"] into every file save_results writes (common.py:112-181)."""
    import struct

    from pandora_amd import common, img_tools
    from pandora_amd.tiff_reader import read_georeferencing, read_tags, write_tiff

    im = np.array([[np.inf, 1, 2, 5, 1, 3, 6, 4, 9, 7, 8], [5, 1, 2, 7, 1, 4, 7, 8, 5, 8, 0], [1, 2, 0, 3, 0, 4, 0, 6, 7, 4, 9],
                   [4, 9, 4, 0, 1, 3, 7, 4, 6, 9, 2], [2, 3, 5, 0, 1, 5, 9, 2, 8, 6, 7], [1, 2, 4, 5, 2, 6, 7, 7, 3, 7, 0],
                   [1, 2, 0, 3, 0, 4, 0, 6, 7, 4, 9], [np.inf, 9, 4, 0, 1, 3, 7, 4, 6, 9, 2]], np.float32)
    geo = {33550: (12, 3, struct.pack("<3d", 0.5, 0.5, 0.0)), 33922: (12, 6, struct.pack("<6d", 0, 0, 0, 600000.0, 4800000.0, 0)),
           34735: (3, 16, struct.pack("<16H", 1, 1, 0, 3, 1024, 0, 1, 1, 1025, 0, 1, 1, 3072, 0, 1, 32631)),  # projected, EPSG:32631
           34737: (2, 8, b"WGS 84|\0")}
    left_path, msk_path, grid_path = (str(tmp_path / n) for n in ("left.tif", "msk.tif", "grid.tif"))
    write_tiff(left_path, im, geo=geo)
    msk = np.zeros(im.shape, np.uint8)
    msk[4, 4] = 7
    write_tiff(msk_path, msk)
    write_tiff(grid_path, np.stack([np.full(im.shape, -3.0, np.float32) - np.arange(11, dtype=np.float32), np.full(im.shape, 2.0, np.float32)]))
    roi = {"col": {"first": 3, "last": 5}, "row": {"first": 3, "last": 5}, "margins": [2, 2, 2, 2]}
    ds = img_tools.create_dataset_from_inputs({"img": left_path, "nodata": np.inf, "mask": msk_path, "disp": grid_path}, roi=roi)
    np.testing.assert_array_equal(ds["im"].data, np.where(np.isinf(im), -9999, im)[1:8, 1:8])
    np.testing.assert_array_equal(ds.coords["row"], np.arange(1, 8))
    np.testing.assert_array_equal(ds.coords["col"], np.arange(1, 8))
    assert ds["msk"].data[3, 3] == 2 and ds["msk"].data[6, 0] == 0 and ds["msk"].data.shape == (7, 7)  # msk[4, 4] of the file
    np.testing.assert_array_equal(ds["disparity"].data[0, 0], -3.0 - np.arange(1, 8))
    crs, transform = ds.attrs["crs"], ds.attrs["transform"]
    assert crs is not None and transform == (0.5, 0.0, 600000.0, 0.0, -0.5, 4800000.0)
    # the ROI reaches grid_estimation through the configuration (matching_cost.py:353-354)
    mc = matching_cost.AbstractMatchingCost(matching_cost_method="census", window_size=3)
    grid = mc.grid_estimation(ds, {"ROI": roi}, (ds["disparity"].sel(band_disp="min"), ds["disparity"].sel(band_disp="max")))
    np.testing.assert_array_equal(grid.attrs["col_to_compute"], np.arange(1, 8))
    assert grid.attrs["crs"] is crs
    # ... and out again with the results
    from pandora_amd.dataset import Dataset

    res = Dataset({"disparity_map": (("row", "col"), np.zeros((7, 7), np.float32)), "validity_mask": (("row", "col"), np.zeros((7, 7), np.int64))},
                  coords={"row": np.arange(7), "col": np.arange(7)}, attrs={"crs": crs, "transform": transform})
    common.save_results(res, Dataset(), str(tmp_path / "out"))
    for name in ("left_disparity.tif", "left_validity_mask.tif"):
        tags = read_tags(str(tmp_path / "out" / name))
        assert {t: tags[t] for t in geo} == geo
        assert read_georeferencing(str(tmp_path / "out" / name))[1] == transform
    # an image without a GeoKey directory has neither crs nor transform (img_tools.py:400-403)
    write_tiff(left_path, im)
    plain = img_tools.create_dataset_from_inputs({"img": left_path, "nodata": np.inf, "disp": [-2, 2]})
    assert plain.attrs["crs"] is None and plain.attrs["transform"] is None


def test_sgm_penalty_configuration_like_the_plugin():
    """plugin_libsgm.rst:136-290: penalty_method sgm_penalty with p2_method constant / negativeGradient / inverseGradient and
    their documented defaults (P1 8, P2 32, alpha 1.0, beta 1, gamma 1); mc_cnn_fast_penalty belongs to another plugin."""
    from pandora_amd.matching_cost.matching_cost import ConfigError
    from pandora_amd.optimization.sgm import Sgm

    assert Sgm(optimization_method="sgm").cfg["penalty"] == {"penalty_method": "sgm_penalty", "p2_method": "constant", "P1": 8, "P2": 32}
    pen = Sgm(optimization_method="sgm", penalty={"p2_method": "negativeGradient"}).cfg["penalty"]
    assert (pen["alpha"], pen["gamma"], "beta" in pen) == (1.0, 1, False)
    pen = Sgm(optimization_method="sgm", penalty={"p2_method": "inverseGradient", "alpha": 3.0}).cfg["penalty"]
    assert (pen["alpha"], pen["beta"], pen["gamma"]) == (3.0, 1, 1)
    for bad in ({"p2_method": "cubic"}, {"penalty_method": "mc_cnn_fast_penalty"}, {"p2_method": "inverseGradient", "beta": 0},
                {"p2_method": "negativeGradient", "alpha": "1"}, {"P1": 8, "P2": 8}):
        with pytest.raises(ConfigError):
            Sgm(optimization_method="sgm", penalty=bad)
    # the maps: the configured P2 is the floor, a path's first pixel has no gradient
    s = Sgm(optimization_method="sgm", penalty={"p2_method": "negativeGradient", "P1": 4, "P2": 20, "alpha": 0.5, "gamma": 60})
    img = (np.arange(20, dtype=np.float32).reshape(4, 5) * 3)
    maps = s.p2_maps(img)
    np.testing.assert_array_equal(maps[0][:, 0], np.float32(60.0))          # (0,+1): column 0 starts the path
    np.testing.assert_array_equal(maps[0][:, 1:], np.float32(60.0 - 0.5 * 3))  # |I(p) - I(p - (0,1))| = 3
    np.testing.assert_array_equal(maps[2][1:], np.float32(60.0 - 0.5 * 15))    # (+1,0): 15 between rows
    s = Sgm(optimization_method="sgm", penalty={"p2_method": "inverseGradient", "P1": 4, "P2": 20, "alpha": 8.0, "beta": 1, "gamma": 2})
    np.testing.assert_array_equal(s.p2_maps(img)[1][:, :-1], np.float32(20.0))  # 8 / (3 + 1) + 2 = 4 < P2: the floor


def test_host_helpers_of_the_library():
    """pmx_host_minmax_i64 / pmx_host_fingerprint (threaded host passes, no GPU): extrema equal numpy's on sizes that do not
    divide into the chunks; the fingerprint is stable, sees any single byte - tail bytes included - and an empty buffer."""
    L = _lib.lib()
    rng = np.random.default_rng(5)
    for n in (1, 7, 131071, 131072, 1 << 20, (1 << 21) + 13):
        a = rng.integers(-2**40, 2**40, n).astype(np.int64)
        lo, hi = ctypes.c_int64(), ctypes.c_int64()
        assert L.pmx_host_minmax_i64(a.ctypes.data_as(_lib.c_i64_p), n, ctypes.byref(lo), ctypes.byref(hi)) == 0
        assert (lo.value, hi.value) == (a.min(), a.max())
    lo = ctypes.c_int64()
    assert L.pmx_host_minmax_i64(None, 0, ctypes.byref(lo), ctypes.byref(lo)) != 0  # refused, not a crash
    from pandora_amd.matching_cost.matching_cost import grid_extrema
    g = rng.integers(-60, 5, (300, 401))
    assert grid_extrema(g) == (g.min(), g.max()) and grid_extrema(g[:, ::2]) == (g[:, ::2].min(), g[:, ::2].max())
    assert grid_extrema(g.astype(np.int32)) == (g.min(), g.max())
    memo = {}
    assert grid_extrema(g, memo) == grid_extrema(g, memo) == (g.min(), g.max()) and len(memo) == 1
    for nbytes in (0, 1, 63, 64, 65, (1 << 20) + 5, (1 << 24) + 77):
        b = rng.integers(0, 256, nbytes).astype(np.uint8)
        h = L.pmx_host_fingerprint(b.ctypes.data, nbytes)
        assert h == L.pmx_host_fingerprint(b.ctypes.data, nbytes)
        for pos in {0, nbytes // 2, nbytes - 1} if nbytes else ():
            b[pos] ^= 0x10
            assert L.pmx_host_fingerprint(b.ctypes.data, nbytes) != h, (nbytes, pos)
            b[pos] ^= 0x10


def test_validity_mask_is_a_recipe_until_somebody_reads_it():
    """criteria.validity_mask without input masks gives a LazyValidity (one line of flags for every row); reading ``.data``
    carries the recipe out on the host - same values as the eager construction - and later in-place updates then work on the
    array like before; with masks the base is the full map."""
    left, right = make_image(np.zeros((6, 9), np.float32), disparity=[-3, 1]), make_image(np.zeros((6, 9), np.float32))
    mc = matching_cost.AbstractMatchingCost(matching_cost_method="census", window_size=3)
    cv = criteria.validity_mask(left, right, mc.allocate_cost_volume(left, (left["disparity"].sel(band_disp="min").data,
                                                                              left["disparity"].sel(band_disp="max").data)))
    lazy = cv["validity_mask"]
    assert isinstance(lazy, criteria.LazyValidity) and lazy.pending and lazy.shape == (6, 9) and cv.sizes["row"] == 6
    line = np.array(lazy._base)
    criteria.mask_border(cv)  # deferred
    assert lazy.pending and lazy.recipe(None) is not None and lazy.recipe(None)[1:] == (None, 1)
    want = np.tile(line, (6, 1))
    want[0], want[-1], want[:, 0], want[:, -1] = 1, 1, 1, 1
    np.testing.assert_array_equal(lazy.data, want)
    assert not lazy.pending and lazy.recipe(None) is None and lazy.data.dtype == np.int64
    criteria.mask_invalid_variable_disparity_range(cv, np.eye(6, 9, dtype=bool))  # now on the array
    assert (cv["validity_mask"].data[2, 2] & 2) and cv["validity_mask"].data is lazy.data
    msk = np.zeros((6, 9), np.int16)
    msk[2, 4] = 1
    cvm = criteria.validity_mask(make_image(np.zeros((6, 9), np.float32), disparity=[-3, 1], msk=msk), right,
                                 mc.allocate_cost_volume(left, (left["disparity"].sel(band_disp="min").data,
                                                                left["disparity"].sel(band_disp="max").data)))
    assert isinstance(cvm["validity_mask"], criteria.LazyValidity) and cvm["validity_mask"]._base.shape == (6, 9)
    assert cvm["validity_mask"].data[2, 4] & 1


def test_classification_segmentation_and_edge_layers_in_the_dataset(tmp_path):
    """img_tools.py:165-231 add_classif / add_segm / add_edges: the reference's own classification file (tests/pandora/
    left_classif.tif, test_pandora_image.py:466-484: bands "cornfields", "olive tree", "forest") next to the cones image, int16,
    also through a ROI window; segm / edges from single-band files."""
    from PIL import Image

    from pandora_amd import img_tools

    cones = os.path.join(ROOT, "tests", "golden", "cones")
    cfg = {"img": os.path.join(cones, "left.png"), "nodata": -9999, "classif": os.path.join(cones, "left_classif.tif"), "disp": [-60, 0]}
    ds = img_tools.create_dataset_from_inputs(cfg)
    assert list(ds.coords["band_classif"]) == ["cornfields", "olive tree", "forest"]
    assert ds["classif"].data.shape == (3, 375, 450) and ds["classif"].data.dtype == np.int16 and ds["classif"].dims == ("band_classif", "row", "col")
    assert [int(b.sum()) for b in ds["classif"].data] == [5643, 14973, 3468]
    roi = {"col": {"first": 10, "last": 100}, "row": {"first": 10, "last": 100}, "margins": [2, 3, 4, 5]}
    win = img_tools.create_dataset_from_inputs(cfg, roi)
    np.testing.assert_array_equal(win["classif"].data, ds["classif"].data[:, 7:106, 8:105])
    segm = np.zeros((375, 450), np.int16)
    segm[:, 200:] = 7
    edges = np.zeros((375, 450), np.float32)
    edges[100, :] = 0.5  # (a float edge map: "higher than zero" - but the reference reads it as int16, img_tools.py:228)
    Image.fromarray(segm).save(tmp_path / "segm.tif")
    Image.fromarray(edges).save(tmp_path / "edges.tif")
    ds2 = img_tools.create_dataset_from_inputs({"img": cfg["img"], "nodata": -9999, "segm": str(tmp_path / "segm.tif"),
                                                "edges": str(tmp_path / "edges.tif"), "disp": [-60, 0]})
    np.testing.assert_array_equal(ds2["segm"].data, segm)
    assert ds2["edges"].data.dtype == np.int16 and ds2["edges"].data.sum() == 0
    with pytest.raises(ValueError, match="dimensions"):
        Image.fromarray(segm[:10]).save(tmp_path / "small.tif")
        img_tools.create_dataset_from_inputs({"img": cfg["img"], "nodata": -9999, "segm": str(tmp_path / "small.tif"), "disp": [-60, 0]})


def test_geometric_prior_configuration_and_cuts():
    """plugin_libsgm.rst:49-78, :122-137: the geometric_prior option of the SGM step and the (pixel, direction) pairs at which a
    path starts again - hand-made layers, every direction of the definition's order."""
    sgm = lambda **kw: optimization.AbstractOptimization(None, optimization_method="sgm", **kw)
    assert sgm().cfg.get("geometric_prior") is None and sgm(geometric_prior={"source": "internal"})._prior_source == "internal"
    for bad in ({"source": "superpixels"}, {"source": "classif"}, {"source": "classif", "classes": []}, {"source": "segm", "classes": ["a"]}, "segm"):
        with pytest.raises(ConfigError):
            sgm(geometric_prior=bad)
    segm = np.array([[1, 1, 2, 2],
                     [1, 1, 2, 2],
                     [3, 3, 3, 2]], np.int16)
    img = make_image(np.zeros((3, 4), np.float32), segm=segm)
    cuts = sgm(geometric_prior={"source": "segm"}).path_cuts(img)
    assert cuts.shape == (8, 3, 4) and cuts.dtype == bool
    np.testing.assert_array_equal(cuts[0], [[0, 0, 1, 0], [0, 0, 1, 0], [0, 0, 0, 1]])  # (0,+1): coming from the left neighbour
    np.testing.assert_array_equal(cuts[1], [[0, 1, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0]])  # (0,-1): from the right neighbour
    np.testing.assert_array_equal(cuts[2], [[0, 0, 0, 0], [0, 0, 0, 0], [1, 1, 1, 0]])  # (+1,0): from above
    np.testing.assert_array_equal(cuts[5], [[0, 0, 0, 0], [1, 1, 1, 0], [0, 0, 0, 0]])  # (-1,0): from below
    np.testing.assert_array_equal(cuts[3], [[0, 0, 0, 0], [0, 0, 1, 0], [0, 1, 1, 0]])  # (+1,+1): from the upper left
    np.testing.assert_array_equal(cuts[7], [[0, 1, 0, 0], [1, 1, 0, 0], [0, 0, 0, 0]])  # (-1,-1): from the lower right
    edges = np.zeros((3, 4), np.int16)
    edges[1, 1] = 5
    ecuts = sgm(geometric_prior={"source": "edges"}).path_cuts(make_image(np.zeros((3, 4), np.float32), edges=edges))
    np.testing.assert_array_equal(ecuts[0], [[0, 0, 0, 0], [0, 1, 1, 0], [0, 0, 0, 0]])  # the edge pixel and the pixel behind it
    np.testing.assert_array_equal(ecuts[2], [[0, 0, 0, 0], [0, 1, 0, 0], [0, 1, 0, 0]])
    bands = np.stack([segm == 1, segm == 2, segm == 3]).astype(np.int16)
    img_c = make_image(np.zeros((3, 4), np.float32), classif=(bands, ["a", "b", "c"]))
    np.testing.assert_array_equal(sgm(geometric_prior={"source": "classif", "classes": ["a", "b", "c"]}).path_cuts(img_c), cuts)
    only_b = sgm(geometric_prior={"source": "classif", "classes": ["b"]}).path_cuts(img_c)  # a and c are one background now
    np.testing.assert_array_equal(only_b[2], [[0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 1, 0]])
    with pytest.raises(AttributeError):
        sgm(geometric_prior={"source": "classif", "classes": ["d"]}).path_cuts(img_c)
    with pytest.raises(AttributeError):
        sgm(geometric_prior={"source": "segm"}).path_cuts(img_c)
