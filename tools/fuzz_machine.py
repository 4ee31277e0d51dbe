"""Random pipelines through PandoraMachine (host glue included: validity criteria, cv_masked, lazy device maps) against the oracle
pipeline of tests/test_gpu_pipeline.py; FUZZ_FROM / FUZZ_TO select the seeds.  Prints the failing draws."""
import json
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402

import tests.test_gpu_pipeline as tp  # noqa: E402
from oracle import capi as orc  # noqa: E402


def draw(seed):
    rng = np.random.default_rng(seed)
    method = str(rng.choice(["census", "census", "sad", "ssd"]))
    win = int(rng.choice([3, 5, 7] if method == "census" else [1, 3, 5]))
    sp = int(rng.choice([1, 1, 2]))
    H = int(rng.integers(max(win, 6), 40))
    W = int(rng.integers(max(win, 8), 80))
    dmin = int(rng.integers(-min(W, 25), 3))
    dmax = dmin + int(rng.integers(0, 30 // sp + 1))
    cfg = {"pipeline": {"matching_cost": {"matching_cost_method": method, "window_size": win, "subpix": sp}}}
    if rng.random() < 0.35 and H - 2 * (win // 2) > 0 and W - 2 * (win // 2) - (1 if sp > 1 else 0) > 0:
        cfg["pipeline"]["aggregation"] = {"aggregation_method": "cbca", "cbca_intensity": float(rng.choice([5.0, 30.0])),
                                          "cbca_distance": int(rng.integers(2, 7))}
    if rng.random() < 0.6:
        P1 = int(rng.integers(1, 12))
        cfg["pipeline"]["optimization"] = {"optimization_method": "sgm", "penalty": {"P1": P1, "P2": P1 + int(rng.integers(1, 50))},
                                           "overcounting": bool(rng.random() < 0.2)}
    cfg["pipeline"]["disparity"] = {"disparity_method": "wta", "invalid_disparity": rng.choice(["NaN", -9999]).item() if True else None}
    if cfg["pipeline"]["disparity"]["invalid_disparity"] == "-9999":
        cfg["pipeline"]["disparity"]["invalid_disparity"] = -9999
    if rng.random() < 0.7:
        cfg["pipeline"]["refinement"] = {"refinement_method": str(rng.choice(["vfit", "quadratic"]))}
    integer = bool(rng.random() < 0.6)
    L, R = tp.pair(H, W, seed=seed, integer=integer)
    layers = None
    if "optimization" in cfg["pipeline"] and rng.random() < 0.35:  # penalty methods that follow the image, geometric priors
        pen = cfg["pipeline"]["optimization"]["penalty"]
        how = rng.random()
        if how < 0.6:
            pen["p2_method"] = str(rng.choice(["negativeGradient", "inverseGradient"]))
            pen["alpha"], pen["gamma"] = float(rng.choice([0.5, 1.0, 3.0])), float(rng.choice([1, 20, 60]))
            if pen["p2_method"] == "inverseGradient":
                pen["beta"] = float(rng.choice([0.5, 1, 4]))
        if how > 0.4:
            source = str(rng.choice(["segm", "edges", "classif"]))
            prior = {"source": source}
            if source == "segm":
                lab = rng.integers(0, 3, (H // 7 + 1, W // 9 + 1))
                layers = {"segm": np.kron(lab, np.ones((7, 9), int))[:H, :W]}
            elif source == "edges":
                layers = {"edges": (rng.random((H, W)) < 0.06).astype(np.int16) * rng.integers(1, 4, (H, W)).astype(np.int16)}
            else:
                bands = (rng.random((3, H, W)) < 0.3).astype(np.int16)
                layers = {"classif": (bands, ["a", "b", "c"])}
                prior["classes"] = [str(x) for x in rng.choice(["a", "b", "c"], int(rng.integers(1, 4)), replace=False)]
            cfg["pipeline"]["optimization"]["geometric_prior"] = prior
    mskL = mskR = None
    if rng.random() < 0.5:
        mskL = rng.choice([0, 0, 0, 0, 0, 0, 0, 1, 2], (H, W)).astype(np.int16)
        if rng.random() < 0.6:
            mskR = rng.choice([0, 0, 0, 0, 0, 0, 0, 1, 2], (H, W)).astype(np.int16)
    return cfg, L, R, dmin, dmax, mskL, mskR, layers


def one(seed):
    cfg, L, R, dmin, dmax, mskL, mskR, layers = draw(seed)
    sp = cfg["pipeline"]["matching_cost"]["subpix"]
    machine, got = tp.run_machine(L, R, cfg, dmin, dmax, mskL, mskR, layers)
    mc_only = {"pipeline": {"matching_cost": cfg["pipeline"]["matching_cost"], "disparity": {"disparity_method": "wta"}}}
    cv0, _, _, _ = tp.oracle_pipeline(orc, L, R, mc_only, dmin, dmax, mskL, mskR)
    val0 = tp.expected_validity(L, R, cfg, dmin, dmax, mskL, mskR, np.min(np.isnan(cv0), axis=2))
    ecv, edisp, eval_, eitp = tp.oracle_pipeline(orc, L, R, cfg, dmin, dmax, mskL, mskR, val0, layers)
    np.testing.assert_array_equal(machine.left_cv["cost_volume"].data, ecv)
    np.testing.assert_array_equal(got["disparity_map"].data, edisp)
    np.testing.assert_array_equal(got["validity_mask"].data, eval_)
    if eitp is not None:
        np.testing.assert_array_equal(got["interpolated_coeff"].data, eitp)


fails = 0
for seed in range(int(os.environ.get("FUZZ_FROM", "0")), int(os.environ.get("FUZZ_TO", "300"))):
    try:
        one(seed)
    except Exception as e:  # noqa: BLE001
        fails += 1
        cfg, L, R, dmin, dmax, mskL, mskR, _ = draw(seed)
        print("FAIL", seed, json.dumps(cfg), L.shape, dmin, dmax, mskL is not None, mskR is not None, type(e).__name__, str(e)[:300].replace("\n", " "))
        if fails > 10:
            break
print("done, failures:", fails)
