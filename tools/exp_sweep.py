#!/usr/bin/env python3
"""Experiment: the sweep schedule of the float32 SGM (two sweeps of four paths) against the oracle on integer-valued costs
(order of the float32 sum invisible) and timed against the family schedule.  Usage: python tools/exp_sweep.py [check] [time]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pandora_amd.engine import Engine  # noqa: E402


def run(eng, cvh, P1, P2, is_max, inv, over, mask=0xFF):
    H, W, D = cvh.shape
    z = np.zeros((H, W), np.float32)
    eng.set_images(z, z, 1)
    cv = eng.alloc_cv(D, 0)
    cv.from_host(cvh)
    eng.sgm(cv, P1, P2, is_max, inv, over, dir_mask=mask)
    out = cv.to_host()
    cv.free()
    return out


def check(eng):
    from oracle import capi as oracle
    bad = 0
    for (H, W, D) in [(5, 9, 10), (12, 20, 30), (40, 70, 30), (33, 100, 61), (70, 41, 100), (25, 130, 129), (37, 50, 257), (20, 90, 150),
                      (2, 17, 20), (50, 3, 40), (9, 2, 12), (64, 64, 257), (17, 300, 300)]:
        rng = np.random.default_rng(H * 1000 + W)
        cvh = np.floor(rng.random((H, W, D)) * 40).astype(np.float32)
        cvh[rng.random(cvh.shape) < 0.05] = np.nan
        for mask in (0x01, 0x04, 0x08, 0x10, 0x1d, 0x02, 0x20, 0x40, 0x80, 0xe2, 0xff):
            exp = oracle.sgm(cvh, 2.0, 9.0, False, 45.0, False, dir_mask=mask)
            eng.set_option("SGM_SCHED", "w")
            eng.set_option("SGM_SWEEP_SHAPE", SHAPE if SHAPE and int(SHAPE.split(",")[0]) * int(SHAPE.split(",")[1]) >= D else None)
            got = run(eng, cvh, 2.0, 9.0, False, 45.0, False, mask)
            eng.set_option("SGM_SCHED", None)
            ok = np.array_equal(got, exp, equal_nan=True)
            if not ok:
                bad += 1
                d = np.argwhere(~((got == exp) | (np.isnan(got) & np.isnan(exp))))
                print(f"MISMATCH {H}x{W}x{D} mask {mask:#x}: {len(d)} cells, first {d[:3].tolist()}", flush=True)
        print(f"shape {H}x{W}x{D} done", flush=True)
    print("check:", "OK" if not bad else f"{bad} failures")


def timing(eng, H, W, D, reps=3):
    import bench

    L, R = bench.synthetic_pair(H, W, 0, D - 1)
    eng.set_images(L, R, 1)
    cv = eng.alloc_cv(D, 0)
    eng.set_option("SGM_SWEEP_SHAPE", SHAPE)
    for sched in ("f", "w", "f", "w"):
        eng.set_option("SGM_SCHED", sched)
        ts = []
        for _ in range(reps):
            eng.census(cv, 5)
            eng.sync()
            t0 = time.perf_counter()
            eng.sgm(cv, 8.0, 32.0, False, 26.0, False)
            eng.sync()
            ts.append((time.perf_counter() - t0) * 1e3)
        print(f"{H}x{W}x{D} sched {sched}: " + " ".join(f"{t:.2f}" for t in ts) + " ms", flush=True)
    eng.set_option("SGM_SCHED", None)
    cv.free()


SHAPE = os.environ.get("SWEEP_SHAPE") or None

if __name__ == "__main__":
    eng = Engine(0)
    eng.set_lazy(False)
    what = sys.argv[1:] or ["check", "time"]
    if "check" in what:
        check(eng)
    if "band" in what:
        for hh in (8, 16, 64):
            timing(eng, hh, 60000, 257, reps=2)
    if "time" in what:
        if "c5" not in what:
            timing(eng, 4096, 4096, 257)
        if "c4" not in what:
            timing(eng, 10000, 10000, 129, reps=2)
    eng.close()
