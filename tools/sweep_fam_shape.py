#!/usr/bin/env python3
"""float32 SGM's marching schedule on a list of shapes through the lane maps that fit (SGM_FAM_SHAPE = lanes per pixel, disparities per
lane, compute wavefronts per workgroup) against `pick_shape`'s choice (k_sgmfam.hip).  ms per census (float32) + SGM + WTA step, a
fresh context per figure.  Usage: python tools/sweep_fam_shape.py H W D [H W D ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pandora_amd.engine import Engine  # noqa: E402

K16, K32 = (3, 5, 7, 9), (3, 5, 6, 9, 12, 16)


def measure(L, R, D, shape, steps=3):
    eng = Engine(0)
    try:
        eng.set_lazy(False)
        eng.set_option("SGM_SCHED", "fam")
        if shape:
            eng.set_option("SGM_FAM_SHAPE", shape)
        eng.set_images(L, R, 1)
        cv = eng.alloc_cv(D, 0)

        def step():
            eng.census(cv, 5)
            eng.sgm(cv, 8.0, 32.0, False, 26.0, False)
            eng.set_validity(None)
            eng.wta(cv, False, -9999.0)

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


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    for H, W, D in [tuple(a[i:i + 3]) for i in range(0, len(a), 3)]:
        L, R = bench.synthetic_pair(H, W, 0, D - 1)
        measure(L, R, D, None)
        k16 = next((k for k in K16 if 16 * k >= D), None)
        k32 = next((k for k in K32 if 32 * k >= D), None)
        shapes = [None] + [f"16,{k16},{nw}" for nw in (4, 8, 10) if k16] + [f"32,{k32},{nw}" for nw in (4, 8, 10) if k32]
        out = [(measure(L, R, D, s), s or "default") for s in shapes]
        print(f"{H} x {W} x {D}: " + "  ".join(f"[{n}] {ms:.2f}" for ms, n in out), flush=True)
