#!/usr/bin/env python3
"""One-GPU timings of the row tiles a `bench.py --gpus N` run hands to each rank (N = 1, 2, 4, 8): the projected strong-scaling
curve that the first run on real links can be compared with.  Two shapes:

* the headline pair, 4096 x 4096, d = [0, 256], Census 5x5 + SGM + WTA + vfit (integer path);
* BASELINE configs[4], 10000 x 10000, d = [-64, 64], Census 5x5 + CBCA + SGM + WTA + vfit (float32 volumes between the steps) -
  the configuration BASELINE words as "row-tiled 8 GPUs".

A tile = the interior rank's rows: H / N owned rows + the SGM margin (40 rows, optimization/optimization.py:43) on both sides.
The gather of the result maps runs under the next step's kernels and is not part of these numbers.
Usage: python tools/bench_tiles.py [--steps K] [--only headline|c5] [--ranks 1,2,4,8]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pandora_amd.dist import row_tile  # noqa: E402
from pandora_amd.engine import Engine  # noqa: E402


def time_tile(eng, L, R, dmin, dmax, cbca, steps):
    eng.set_images(L, R, 1)
    cv = eng.alloc_cv(dmax - dmin + 1, dmin)

    def step():
        eng.census(cv, 5)
        if cbca:
            eng.cbca(cv, 2, 30.0, 5)
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
    eng.set_profiling(True)
    eng.reset_stage_times()
    step()
    eng.sync()
    from pandora_amd import _lib

    stages = {k: round(eng.stage_time(k)[0], 3) for k in _lib.STAGES if eng.stage_time(k)[1]}
    eng.set_profiling(False)
    cv.free()
    return best, stages


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--only", default=None)
    ap.add_argument("--ranks", default="1,2,4,8", help="rank counts whose interior tile is timed")
    args = ap.parse_args()
    shapes = []
    if args.only in (None, "headline"):
        shapes.append(("headline 4096x4096 d=[0,256] census5+sgm+wta+vfit", 4096, 4096, 0, 256, False))
    if args.only in (None, "c5"):
        shapes.append(("configs[4] 10000x10000 d=[-64,64] census5+cbca+sgm+wta+vfit", 10000, 10000, -64, 64, True))
    for label, H, W, dmin, dmax, cbca in shapes:
        L, R = bench.synthetic_pair(H, W, dmin, dmax)
        print(f"# {label}: interior tile of an N-rank run on ONE MI355X, best of 3 x {args.steps} steps")
        print("ranks  tile rows  ms/step   projected speedup   stage ms (one more step)")
        whole = None
        for n in [int(x) for x in args.ranks.split(",")]:
            rank = 0 if n == 1 else n // 2  # an interior rank: margin on both sides
            (_, _), (tlo, thi) = row_tile(H, n, rank, bench.SGM_MARGIN if n > 1 else 0)
            # a context of its own per tile size, as a rank of an N-rank run has: its volumes are allocated (and placed) for THIS size -
            # a tile that re-used the whole pair's cached buffers measured 13 % slower (2.99 against 2.65 ms at 592 rows, round 6)
            eng = Engine(0)
            ms, stages = time_tile(eng, np.ascontiguousarray(L[tlo:thi]), np.ascontiguousarray(R[tlo:thi]), dmin, dmax, cbca,
                                   args.steps if not cbca else max(1, args.steps // 2))
            whole = ms if whole is None else whole
            eng.close()
            print(f"{n:5d}  {thi - tlo:9d}  {ms:7.3f}   {whole / ms:6.2f}x             {stages}", flush=True)
        del L, R


if __name__ == "__main__":
    main()
