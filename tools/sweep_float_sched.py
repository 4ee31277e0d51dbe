#!/usr/bin/env python3
"""float32 SGM on a list of shapes through its schedules (SGM_SCHED seq / par / fam against the library's choice): what the size rule of
`pmx_launch_sgm` (k_sgm.hip) is checked against.  Census costs as float32 (lazy mode off), ms per SGM + WTA step, a fresh context per
figure.  Usage: python tools/sweep_float_sched.py H W D [H W D ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pandora_amd.engine import Engine  # noqa: E402


def measure(L, R, D, sched, steps=3):
    eng = Engine(0)
    try:
        eng.set_lazy(False)
        if sched:
            eng.set_option("SGM_SCHED", sched)
        eng.set_images(L, R, 1)
        cv = eng.alloc_cv(D, 0)
        work = eng.alloc_cv(D, 0)

        def step():
            eng.census(work, 5)
            eng.sgm(work, 8.0, 32.0, False, 26.0, False)
            eng.set_validity(None)
            eng.wta(work, False, -9999.0)

        step()
        eng.sync()
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            eng.sync()
            ms = (time.perf_counter() - t0) / steps * 1e3
            best = ms if best is None or ms < best else best
        cv.free()
        work.free()
        return best
    except Exception as e:  # a schedule that does not take the shape
        return float("nan")
    finally:
        eng.close()


a = [int(x) for x in sys.argv[1:]]
for H, W, D in [tuple(a[i:i + 3]) for i in range(0, len(a), 3)]:
    L, R = bench.synthetic_pair(H, W, 0, D - 1)
    measure(L, R, D, None)  # (the shape's first context runs 5 - 9 % slow: not counted)
    out = [(measure(L, R, D, s), s or "default") for s in (None, "seq", "par", "fam")]
    print(f"{H} x {W} x {D}: " + "  ".join(f"[{n}] {ms:.2f}" for ms, n in out), flush=True)
