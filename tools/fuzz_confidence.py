"""Random volumes through the cost-volume confidence kernels (ambiguity, risk, interval bounds), device against oracle, bit for
bit: random shapes and sub-pixel factors, quantised costs (ties), NaN inside and outside the per-pixel ranges, pixels without any
cost, "max" measures, random eta grids and thresholds.  FUZZ_FROM / FUZZ_TO."""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402

from oracle import capi as orc  # noqa: E402
from pandora_amd.engine import Engine  # noqa: E402

eng = Engine(0)


def one(seed):
    rng = np.random.default_rng(seed)
    H, W = int(rng.integers(2, 40)), int(rng.integers(2, 90))
    sp = int(rng.choice([1, 1, 2, 4]))
    dmin = int(rng.integers(-40, 3))
    span = int(rng.integers(3, 60 // sp + 4))
    dmax = dmin + span
    D = span * sp + 1
    eng.set_images(np.zeros((H, W), np.float32), np.zeros((H, W), np.float32), sp)
    cv = eng.alloc_cv(D, dmin)
    quantised = rng.random() < 0.5
    vol = rng.integers(0, 14, (H, W, D)).astype(np.float32) if quantised else (rng.random((H, W, D)) * 50 - 10).astype(np.float32)
    vol[rng.random((H, W, D)) < rng.choice([0.0, 0.1, 0.4])] = np.nan
    vol[rng.integers(0, H), rng.integers(0, W), :] = np.nan
    cv.from_host(vol)
    gmin = rng.integers(dmin, dmin + max(span // 3, 1), (H, W)).astype(np.int64)
    gmax = np.minimum(dmax, gmin + rng.integers(1, span + 1, (H, W))).astype(np.int64)
    negate = bool(rng.random() < 0.3)
    eta_max, eta_step = float(rng.choice([0.3, 0.7, 0.9])), float(rng.choice([0.01, 0.05, 0.1]))
    etas = np.arange(0.0, eta_max, eta_step)
    disp_range = (dmin + np.arange(D) / sp).astype(np.float32)
    src = -vol if negate else vol
    np.testing.assert_array_equal(eng.ambiguity(cv, etas, gmin, gmax, negate), orc.ambiguity(src, etas, gmin, gmax, disp_range), err_msg="ambiguity")
    for g, e, name in zip(eng.risk(cv, etas, gmin, gmax, negate), orc.risk(src, etas, gmin, gmax, disp_range),
                          ("risk_max", "risk_min", "disp_sup", "disp_inf")):
        np.testing.assert_array_equal(g, e, err_msg=name)
    thr, tf = float(rng.choice([0.0, 0.5, 0.9, 1.0])), float(rng.choice([-1.0, 1.0]))
    for g, e, name in zip(eng.interval_bounds(cv, thr, tf, gmin, gmax), orc.interval_bounds(vol, thr, tf, gmin, gmax, disp_range), ("inf", "sup")):
        np.testing.assert_array_equal(g, e, err_msg="interval " + name)
    cv.free()


fails = 0
for seed in range(int(os.environ.get("FUZZ_FROM", "0")), int(os.environ.get("FUZZ_TO", "300"))):
    try:
        one(seed)
    except Exception as e:  # noqa: BLE001
        fails += 1
        print("FAIL", seed, type(e).__name__, str(e)[:400].replace("\n", " "))
        if fails > 6:
            break
print("done, failures:", fails)
