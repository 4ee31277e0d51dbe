#!/usr/bin/env python3
"""float32 SGM schedules at BASELINE sizes: "seq" (one launch per path), "fam" (horizontal pair + two fused three-path marching
passes, csrc/k_sgmfam.hip), per-stage HIP-event times.  The volume holds census costs written as float32 (lazy mode off).
Usage: python tools/bench_sgm_sched.py [C3 C4 C5] [--sched seq,fam] [--reps 3]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pandora_amd.engine import Engine  # noqa: E402

SIZES = {"C2": (375, 450, -60, 0), "C3": (2048, 2048, 0, 128), "C4": (4096, 4096, 0, 256), "C5": (10000, 10000, -64, 64),
         "W8k": (2048, 8192, 0, 128),
         # the marching kernels' time against the number of rows: slope = time per row, intercept = pipeline fill
         "C4h1k": (1024, 4096, 0, 256), "C4h2k": (2048, 4096, 0, 256), "C4h3k": (3072, 4096, 0, 256),
         # the same lane map (3 disparities per lane) with 43 and with 64 active lanes: issue-bound or memory-bound?
         "K3a": (4096, 4096, 0, 128), "K3b": (4096, 4096, 0, 191)}

ap = argparse.ArgumentParser()
ap.add_argument("names", nargs="*", default=["C3", "C4"])
ap.add_argument("--sched", default="seq,fam")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--masks", default="0xff", help="comma list of direction masks (pmx_debug_sgm_directions)")
args = ap.parse_args()

eng = Engine(0)
eng.set_lazy(False)
eng.set_profiling(True)
for name in args.names:
    H, W, dmin, dmax = SIZES[name]
    D = dmax - dmin + 1
    rng = np.random.default_rng(1)
    base = rng.integers(0, 255, (64, W + 16)).astype(np.float32)
    L = np.tile(base[:, 8:8 + W], (-(-H // 64), 1))[:H].copy()
    R = np.tile(base[:, 5:5 + W], (-(-H // 64), 1))[:H].copy()
    eng.set_images(L, R, 1)
    cv = eng.alloc_cv(D, dmin)
    for sched, mask in [(s_, int(m_, 0)) for s_ in args.sched.split(",") for m_ in args.masks.split(",")]:
        eng.set_option("SGM_SCHED", sched)
        eng.census(cv, 5)
        eng.sgm(cv, 8.0, 32.0, False, 26.0, False, dir_mask=mask)  # warm-up: allocations, hand-off buffer
        eng.sync()
        eng.reset_stage_times()
        for _ in range(args.reps):
            eng.census(cv, 5)
            eng.sgm(cv, 8.0, 32.0, False, 26.0, False, dir_mask=mask)
        eng.sync()
        path_ms, path_n = eng.stage_time("sgm_path")
        fam_ms, fam_n = eng.stage_time("sgm_family")
        cells = H * W * D
        total = (path_ms + fam_ms) / args.reps
        print(json.dumps({"size": name, "shape": [H, W, D], "sched": sched, "mask": hex(mask), "sgm_ms": round(total, 3),
                          "line_kernel_ms_per_launch": round(path_ms / max(path_n, 1), 3), "line_launches": path_n // args.reps,
                          "family_ms_per_launch": round(fam_ms / max(fam_n, 1), 3), "family_launches": fam_n // args.reps,
                          "Gcell/s": round(cells / total / 1e6, 1)}), flush=True)
    cv.free()
eng.close()
