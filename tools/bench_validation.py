#!/usr/bin/env python3
"""Stage times of the validation flows at a BASELINE-sized pair: left pipeline (census5 + SGM + WTA + vfit), then the
right side either re-indexed from the left volume (cross_checking_fast) or recomputed (accurate), then both cross-checks.
Usage: python tools/bench_validation.py [H W dmin dmax]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pandora_amd.engine import Engine  # noqa: E402

H, W, dmin, dmax = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (2048, 2048, 0, 128)
D = dmax - dmin + 1
L, R = bench.synthetic_pair(H, W, dmin, dmax)
eng = Engine(0)
out = {"shape": [H, W, D]}


def clock(fn, reps=3):
    fn()
    eng.sync()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    eng.sync()
    return round((time.perf_counter() - t) / reps * 1e3, 3)


def left():
    eng.set_images(L, R, 1)
    cv = eng.alloc_cv(D, dmin)
    eng.census(cv, 5)
    eng.sgm(cv, 8, 32, False, 26.0, False)
    eng.set_validity(None)
    eng.wta(cv, False, -9999.0)
    eng.refine(cv, "vfit", False)
    return cv


cv = left()
ldisp, lval = eng.get_disparity()
out["left_pipeline_ms (incl. image upload)"] = clock(lambda: left())
holder = {}


def right_fast():
    holder["r"] = eng.reverse_cost_volume(cv, -dmax)
    eng.set_validity(None)
    eng.wta(holder["r"], False, -9999.0)
    eng.refine(holder["r"], "vfit", False)


out["right_fast_ms (reverse + wta + vfit)"] = clock(right_fast)
rdisp, rval = eng.get_disparity()
out["cross_check_ms (one direction, host maps in/out)"] = clock(lambda: eng.cross_checking(ldisp, lval, rdisp, dmin, dmax, 1.0))
v, c = eng.cross_checking(ldisp, lval, rdisp, dmin, dmax, 1.0)
out["rejected_fraction"] = round(float(np.mean((v & 0x300) != 0)), 4)
print(json.dumps(out, indent=1))
