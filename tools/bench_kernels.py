#!/usr/bin/env python3
"""Per-kernel timings of the general (float32) path at a BASELINE-sized volume: every stage is run
eagerly (lazy representations off) and timed with the engine's HIP-event stage timers.
Usage: python tools/bench_kernels.py [H W dmin dmax]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pandora_amd.engine import Engine  # noqa: E402

H, W, dmin, dmax = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (2048, 2048, 0, 128)
D = dmax - dmin + 1
cells = H * W * D
L, R = bench.synthetic_pair(H, W, dmin, dmax)
eng = Engine(0)
eng.set_lazy(False)
eng.set_images(L, R, 1)
eng.set_profiling(True)
out = {"shape": [H, W, D], "GB_per_volume": round(cells * 4 / 1e9, 3)}


def timed(name, stages, fn, reps=3):
    fn()
    eng.sync()
    eng.reset_stage_times()
    for _ in range(reps):
        fn()
    eng.sync()
    t = {s: eng.stage_time(s) for s in stages}
    ms = sum(v[0] for v in t.values()) / reps
    out[name] = {"ms": round(ms, 3), "Gcell/s": round(cells / ms / 1e6, 1), "parts_ms": {s: round(v[0] / reps, 3) for s, v in t.items()}}


cv = eng.alloc_cv(D, dmin)
if os.environ.get("PMX_BENCH_ONLY") == "cbca":
    eng.census(cv, 5)
    timed("cbca (d=5, i=30)", ["cbca_arms", "cbca_h", "cbca_v"], lambda: eng.cbca(cv, 2, 30.0, 5), reps=2)
    print(json.dumps(out, indent=1))
    sys.exit(0)
if os.environ.get("PMX_BENCH_ONLY") == "census_cbca":  # as the pipeline runs them: the census costs stay implicit until pass H
    eng.set_lazy(True)

    def both():
        eng.census(cv, 5)
        eng.cbca(cv, 2, 30.0, 5)

    timed("census5 + cbca (d=5, i=30), lazy", ["census_transform", "census_cost", "cbca_arms", "cbca_h", "cbca_v"], both, reps=2)
    print(json.dumps(out, indent=1))
    sys.exit(0)
if os.environ.get("PMX_BENCH_ONLY") == "zncc":
    timed("zncc5", ["zncc"], lambda: eng.zncc(cv, 5), reps=2)
    timed("zncc11", ["zncc"], lambda: eng.zncc(cv, 11), reps=1)
    print(json.dumps(out, indent=1))
    sys.exit(0)
timed("census5 (float volume)", ["census_transform", "census_cost"], lambda: eng.census(cv, 5))
timed("wta (float)", ["wta"], lambda: eng.wta(cv, False, -9999.0))
timed("refine vfit", ["refine"], lambda: eng.refine(cv, "vfit", False))
timed("sgm 8 passes (float)", ["sgm_path"], lambda: eng.sgm(cv, 8, 32, False, 26.0, False), reps=2)
eng.census(cv, 5)
timed("cbca (d=5, i=30)", ["cbca_arms", "cbca_h", "cbca_v"], lambda: eng.cbca(cv, 2, 30.0, 5), reps=2)
timed("sad5", ["sad_ssd"], lambda: eng.sad_ssd(cv, 5, False), reps=2)
timed("zncc5", ["zncc"], lambda: eng.zncc(cv, 5), reps=2)
timed("zncc11", ["zncc"], lambda: eng.zncc(cv, 11), reps=1)
print(json.dumps(out, indent=1))
