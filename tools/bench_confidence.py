#!/usr/bin/env python3
"""Wall time of the cost-volume confidence reductions (ambiguity, risk, interval_bounds) on a BASELINE-sized float volume;
includes the host<->device copies of the grids and maps the C ABI performs.  Usage: python tools/bench_confidence.py [H W D]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pandora_amd.engine import Engine  # noqa: E402

H, W, D = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (2048, 2048, 129)
L, R = bench.synthetic_pair(H, W, 0, D - 1)
eng = Engine(0)
eng.set_images(R, L, 1)
cv = eng.alloc_cv(D, -(D - 1))
eng.census(cv, 5)
eng.cv_masked(cv, 5)
gmin, gmax = np.full((H, W), -(D - 1), np.int64), np.zeros((H, W), np.int64)
etas = np.arange(0.0, 0.7, 0.01)
CALLS = []
for label, g0, g1 in (("per-pixel grids", gmin, gmax), ("constant range (no grids)", None, None)):
    CALLS += [(f"ambiguity, {label}", lambda g0=g0, g1=g1: eng.ambiguity(cv, etas, g0, g1)),
              (f"risk, {label}", lambda g0=g0, g1=g1: eng.risk(cv, etas, g0, g1)),
              (f"interval_bounds, {label}", lambda g0=g0, g1=g1: eng.interval_bounds(cv, 0.9, -1.0, g0, g1))]
for name, fn in CALLS:
    fn()
    t = time.perf_counter()
    for _ in range(3):
        fn()
    print(f"{name:44s} {1e3 * (time.perf_counter() - t) / 3:8.2f} ms per call ({H}x{W}x{D})")
