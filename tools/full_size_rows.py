#!/usr/bin/env python3
"""The rows beyond the hot path and the sub-pixel matching costs at FULL size against the CPU oracle - the parity suite
runs them on small images: whole 4096 x 4096 maps for the 2-D steps (median, bilateral, cross-checking, disparity range), a 2.2 G-cell
volume (8.6 GB: offsets beyond 2^32 bytes) for the confidence kernels, the image's bottom rows for sub-pixel SAD / SSD / ZNCC / census
volumes.  Test infrastructure (uses oracle/).  Usage (GPU box): python tools/full_size_rows.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import capi as oracle  # noqa: E402
from pandora_amd.engine import Engine  # noqa: E402
from tests.test_gpu_full_size import big_pair, SIZES  # noqa: E402

oracle.set_threads(0)
bad = 0


def report(label, pairs, t0, tol=None):
    global bad
    diffs = []
    for a, b in pairs:
        a, b = np.asarray(a), np.asarray(b)
        if tol is None:
            diffs.append(int((~((a == b) | (np.isnan(a.astype(np.float64)) & np.isnan(b.astype(np.float64))))).sum()))
        else:
            with np.errstate(invalid="ignore"):
                diffs.append(int((~((np.abs(a - b) <= tol) | (np.isnan(a) & np.isnan(b)))).sum()))
    ok = not any(diffs)
    bad += 0 if ok else 1
    print(f"{'ok ' if ok else 'BAD'} {label}: mismatches {diffs}  ({time.time() - t0:.0f} s)", flush=True)


H, W = 4096, 4096
rng = np.random.default_rng(3)
eng = Engine(0)
eng.set_lazy(False)

# ---- 2-D steps on whole maps -------------------------------------------------------------------------------------------------------------
disp = (rng.integers(-40, 41, (H, W)) + rng.integers(0, 4, (H, W)) * 0.25).astype(np.float32)
val = np.zeros((H, W), np.int64)
val[rng.random((H, W)) < 0.03] = 1 << 0 | 1 << 9  # invalid pixels
val[rng.random((H, W)) < 0.03] |= 1 << 3          # an information bit
disp_r = (-disp + rng.integers(-2, 3, (H, W))).astype(np.float32)
L, R = (rng.random((H, W)) * 255).astype(np.float32), (rng.random((H, W)) * 255).astype(np.float32)
eng.set_images(L, R, 1)
t0 = time.time()
report("median filter 3 / 5 of a 4096^2 disparity map",
       [(eng.median_filter_disparity(disp, val, s), oracle.filter_median_disparity(disp, val, s)) for s in (3, 5)], t0)
t0 = time.time()
report("bilateral filter (sigma 4 / 2) of a 4096^2 map, 1e-6",
       [(eng.bilateral_filter_disparity(disp, val, 4.0, 2.0), oracle.filter_bilateral_disparity(disp, val, 4.0, 2.0))], t0, tol=1e-6 * 64)
t0 = time.time()
gv, gc = eng.cross_checking(disp, val, disp_r, -40, 40, 1.0)
ov, oc = oracle.cross_checking(disp, val, disp_r, -40, 40, 1.0)
report("cross-checking of 4096^2 maps", [(gv, ov), (gc, oc)], t0)
t0 = time.time()
glo, ghi = eng.disparity_range(disp, val, 5, 2, -45, 45)
olo, ohi = oracle.disparity_range(disp, val, 5, 2, -45, 45)
report("multiscale disparity range of a 4096^2 map", [(glo, olo), (ghi, ohi)], t0)

# ---- confidence on a volume beyond 2^32 bytes ----------------------------------------------------------------------------------------------
D, dmin = 129, -64
t0 = time.time()
vol = rng.integers(0, 40, (H, W, D), dtype=np.int8).astype(np.float32)
vol[rng.random((H, W)) < 0.01] = np.nan  # pixels without any cost
vol[:, :, 5][rng.random((H, W)) < 0.1] = np.nan
cv = eng.alloc_cv(D, dmin)
cv.from_host(vol)
gmin = rng.integers(dmin, dmin + 3, (H, W)).astype(np.int64)
gmax = (gmin + rng.integers(20, D - 5, (H, W))).astype(np.int64)
disp_range = (dmin + np.arange(D)).astype(np.float32)
etas = np.arange(0.0, 0.7, 0.01)
print(f"   (volume of {vol.nbytes / 1e9:.1f} GB made and uploaded in {time.time() - t0:.0f} s)", flush=True)
t0 = time.time()
report("ambiguity integral, 4096^2 x 129", [(eng.ambiguity(cv, etas, gmin, gmax), oracle.ambiguity(vol, etas, gmin, gmax, disp_range))], t0)
t0 = time.time()
report("risk (four maps), 4096^2 x 129", list(zip(eng.risk(cv, etas, gmin, gmax), oracle.risk(vol, etas, gmin, gmax, disp_range))), t0)
t0 = time.time()
report("interval bounds, 4096^2 x 129", list(zip(eng.interval_bounds(cv, 0.9, -1.0, gmin, gmax),
                                                  oracle.interval_bounds(vol, 0.9, -1.0, gmin, gmax, disp_range))), t0)
cv.free()
del vol

# ---- sub-pixel matching costs: the image's bottom rows -------------------------------------------------------------------------------------
SIZES["DBG"] = (H, W, -8, 8)
Lp, Rp = big_pair("DBG")
strip, skip = 40, 14
for label, sp, win, fn in (("census 5x5, subpix 4", 4, 5, "census"), ("SAD 5x5, subpix 2", 2, 5, "sad"), ("SSD 3x3, subpix 4", 4, 3, "ssd"),
                           ("ZNCC 7x7, subpix 2 (1e-5)", 2, 7, "zncc"), ("ZNCC 5x5, subpix 4 (1e-5)", 4, 5, "zncc")):
    t0 = time.time()
    Dk = 16 * sp + 1
    eng.set_images(Lp, Rp, sp)
    cvk = eng.alloc_cv(Dk, -8)
    Ls, Rs = np.ascontiguousarray(Lp[-strip:]), np.ascontiguousarray(Rp[-strip:])
    if fn == "census":
        eng.census(cvk, win)
        exp = oracle.census_cost(Ls, Rs, Dk, -8, sp, win)
    elif fn in ("sad", "ssd"):
        eng.sad_ssd(cvk, win, fn == "ssd")
        exp = oracle.sad_ssd(Ls, Rs, Dk, -8, sp, win, fn == "ssd")
    else:
        eng.zncc(cvk, win)
        exp = oracle.zncc(Ls, Rs, Dk, -8, sp, win)
    got = cvk.rows_to_host(H - strip + skip, H)
    cvk.free()
    report(f"{label}, 4096^2 x {Dk}: bottom rows of the volume", [(got, exp[skip:])], t0, tol=1e-5 if fn == "zncc" else None)
eng.close()
print("failures:", bad)
sys.exit(1 if bad else 0)
