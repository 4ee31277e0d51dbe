"""Random maps through the disparity filters (median, bilateral, disparity_denoiser) and the cross-checking, device against
oracle.  FUZZ_FROM / FUZZ_TO."""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
from scipy.ndimage import gaussian_filter  # noqa: E402

from oracle import capi as orc  # noqa: E402
from pandora_amd.engine import Engine  # noqa: E402

eng = Engine(0)


def one(seed):
    rng = np.random.default_rng(seed)
    H, W = int(rng.integers(2, 70)), int(rng.integers(2, 120))
    disp = (rng.integers(-40, 10, (H, W)) + rng.choice([0.0, 0.25, 0.5, -0.125], (H, W))).astype(np.float32)
    val = np.where(rng.random((H, W)) < 0.2, rng.choice([1, 2, 64, 128, 256, 512, 4, 8], (H, W)), 0).astype(np.int64)
    if rng.random() < 0.5:
        disp[rng.integers(0, H), rng.integers(0, W)] = np.nan
    if rng.random() < 0.2:
        disp[rng.integers(0, H), rng.integers(0, W)] = np.inf
    what = []
    size = int(rng.choice([1, 3, 5, 7, 9, 11, 13, 15]))
    if size <= min(H, W):
        what.append(f"median {size}")
        np.testing.assert_array_equal(eng.median_filter_disparity(disp, val, size), orc.filter_median_disparity(disp, val, size))
    sc, ss = float(rng.choice([1.0, 2.0, 4.0])), float(rng.choice([0.7, 1.5, 3.0, 6.0]))
    what.append(f"bilateral {sc} {ss}")
    np.testing.assert_allclose(eng.bilateral_filter_disparity(disp, val, sc, ss), orc.filter_bilateral_disparity(disp, val, sc, ss),
                               rtol=1e-6, atol=2e-6, equal_nan=True)  # (expf of the device vs libm: an ulp of a weight, against values of tens)
    fin = np.where(np.isfinite(disp), disp, np.float32(0))
    fs = int(rng.choice([1, 3, 5, 11]))
    band = rng.integers(0, 2000, (H, W)).astype(np.float32)
    grad = np.gradient(gaussian_filter(fin, sigma=1.5)) if min(H, W) > 1 else [np.zeros_like(fin)] * 2
    what.append(f"denoiser {fs}")
    np.testing.assert_allclose(eng.denoise_disparity(fin, val, band, grad[0], grad[1], fs, 4.0, 100.0, 12.0),
                               orc.denoise_disparity(fin, val, band, grad[0], grad[1], fs, 4.0, 100.0, 12.0), rtol=1e-6, atol=1e-6)
    # cross-checking of two maps + the two interpolations
    dmin, dmax = -40, 10
    right = (-(disp) + rng.choice([0.0, 0.0, 0.0, 1.0, -2.0], (H, W))).astype(np.float32)
    right[~np.isfinite(right)] = 0
    thr = float(rng.choice([0.5, 1.0, 2.0]))
    gv, gc = eng.cross_checking(fin, val, right, dmin, dmax, thr)
    ev, ec = orc.cross_checking(fin, val, right, dmin, dmax, thr)
    what.append("cross_checking")
    np.testing.assert_array_equal(gv, ev)
    np.testing.assert_array_equal(gc, ec)
    return what


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
