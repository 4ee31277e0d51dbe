#!/usr/bin/env python3
"""The integer path's family form on a list of shapes through every horizontal-pair mode (SGM8_HPAIR 1 / 2 / 3) with the marching
kernel on the cost volume or from the census words (SGM8_CODES 0 / default): what `pmx_launch_sgm8`'s rule for short images is
checked against.  ms per census + SGM + WTA + vfit step, a fresh context per measurement (allocation placement).
Usage: python tools/sweep_hpair.py H W D [H W D ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pandora_amd.engine import Engine  # noqa: E402


def measure(L, R, D, opts, steps=4):
    eng = Engine(0)
    try:
        for k, v in opts.items():
            eng.set_option(k, v)
        eng.set_images(L, R, 1)
        cv = eng.alloc_cv(D, 0)

        def step():
            eng.census(cv, 5)
            eng.sgm(cv, 8.0, 32.0, False, 26.0, False)
            eng.set_validity(None)
            eng.wta(cv, False, -9999.0)
            eng.refine(cv, "vfit", False)

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
        return best
    finally:
        eng.close()


def main():
    a = [int(x) for x in sys.argv[1:]]
    shapes = [tuple(a[i:i + 3]) for i in range(0, len(a), 3)]
    combos = [("default", {})] + [(f"hpair {hp} codes {c}", {"SGM8_HPAIR": hp, **({} if c == "-" else {"SGM8_CODES": c})})
                                  for hp in "123" for c in ("0", "-")]
    for H, W, D in shapes:
        L, R = bench.synthetic_pair(H, W, 0, D - 1)
        out = []
        for name, opts in combos:
            out.append((measure(L, R, D, dict(opts, SGM8_FAM="1")), name))
        base = out[0][0]
        print(f"{H} x {W} x {D}: " + "  ".join(f"[{n}] {ms:.2f}" for ms, n in out) + f"   best: {min(out)[1]} ({min(out)[0] / base:.2f} of default)", flush=True)


if __name__ == "__main__":
    main()
