#!/usr/bin/env python3
"""Census + SGM + WTA + vfit with per-pixel disparity grids (the fine scale of a multiscale run) at C3 size: the integer fast
path with per-pixel valid intervals (lazy) against the float32 path (eager).  Usage: python tools/bench_ranges.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pandora_amd.engine import Engine  # noqa: E402

H, W, dmin, dmax = 2048, 2048, 0, 128
L, R = bench.synthetic_pair(H, W, dmin, dmax)
rng = np.random.default_rng(0)
centre = rng.integers(10, 118, (H // 8, W // 8)).repeat(8, 0).repeat(8, 1)
lo, hi = (centre - 10).astype(np.float64), (centre + 10).astype(np.float64)
out = {}
eng = Engine(0)
for mode in ("lazy", "eager"):
    eng.set_lazy(mode == "lazy")
    eng.set_images(L, R, 1)
    eng.set_disparity_grids(lo, hi)
    cv = eng.alloc_cv(dmax - dmin + 1, dmin)

    def step():
        eng.census(cv, 5)
        eng.cv_masked(cv, 5)
        eng.sgm(cv, 8.0, 32.0, False, 26.0, False)
        eng.set_validity(None)
        eng.wta(cv, False, -9999.0)
        eng.refine(cv, "vfit", False)

    step()
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    eng.sync()
    out[mode] = {"ms_per_step": round((time.perf_counter() - t0) / 5 * 1e3, 3), "disp_checksum": float(np.nansum(eng.get_disparity()[0]))}
    cv.free()
print(json.dumps(out))
