#!/usr/bin/env python3
"""Device time of the five BASELINE.json configurations on ONE MI355X, through the engine (C ABI), inputs resident:
C1 cones a_local_block_matching.json as written (ZNCC 5x5 subpix 4 + WTA + quadratic) and its SAD variant, C2 cones
census+CBCA+SGM, C3 2048^2 census+SGM, C4 4096^2 ZNCC 11x11 + SGM (float32 path), C5 10000^2 census+CBCA+SGM (one scale, the
fine one of the 2-scale run).  Usage: python tools/bench_configs.py [--stages] [C1 C2 ...]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pandora_amd.engine import Engine  # noqa: E402


def cones():
    from PIL import Image

    d = os.path.join(ROOT, "tests", "golden", "cones")
    return (np.array(Image.open(os.path.join(d, "left.png"))).astype(np.float32),
            np.array(Image.open(os.path.join(d, "right.png"))).astype(np.float32))


def run(eng, L, R, dmin, dmax, subpix, cost, cbca, sgm, refine, steps):
    eng.set_images(L, R, subpix)
    D = (dmax - dmin) * subpix + 1
    cv = eng.alloc_cv(D, dmin)
    is_max = cost[0] == "zncc"

    def step():
        if cost[0] == "census":
            eng.census(cv, cost[1])
        elif cost[0] == "zncc":
            eng.zncc(cv, cost[1])
        else:
            eng.sad_ssd(cv, cost[1], cost[0] == "ssd")
        if cbca:
            eng.cbca(cv, cost[1] // 2, 30.0, 5)
        if sgm:
            eng.sgm(cv, 8.0, 32.0, is_max, float(cost[1] ** 2 + 1) if cost[0] == "census" else 2.0, False)
        eng.set_validity(None)
        eng.wta(cv, is_max, -9999.0)
        eng.refine(cv, refine, is_max)

    step()
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    eng.sync()
    ms = (time.perf_counter() - t0) / steps * 1e3
    stages = None
    if STAGES:  # one more step with the library's per-stage HIP events
        from pandora_amd import _lib

        eng.set_profiling(True)
        eng.reset_stage_times()
        step()
        eng.sync()
        stages = {k: round(eng.stage_time(k)[0], 3) for k in _lib.STAGES if eng.stage_time(k)[1]}
        eng.set_profiling(False)
    cv.free()
    cells = L.shape[0] * L.shape[1] * D
    out = {"shape": [L.shape[0], L.shape[1], D], "ms": round(ms, 3), "Gdisp/s": round(cells / ms / 1e6, 2)}
    if stages:
        out["stages_ms"] = stages
    return out


CONFIGS = {
    "C1 cones zncc5 subpix4 + wta + quadratic (a_local_block_matching.json)": lambda e: run(e, *cones(), -60, 0, 4, ("zncc", 5), False, False, "quadratic", 20),
    "C1' cones sad5 d=[-64,0] + wta": lambda e: run(e, *cones(), -64, 0, 1, ("sad", 5), False, False, "vfit", 20),
    "C2 cones census5 + cbca + sgm + wta + vfit": lambda e: run(e, *cones(), -60, 0, 1, ("census", 5), True, True, "vfit", 20),
    "C3 2048^2 d=[0,128] census5 + sgm + wta + vfit": lambda e: run(e, *bench.synthetic_pair(2048, 2048, 0, 128), 0, 128, 1, ("census", 5), False, True, "vfit", 5),
    "C4 4096^2 d=[0,256] zncc11 + sgm + wta + vfit (float32)": lambda e: run(e, *bench.synthetic_pair(4096, 4096, 0, 256), 0, 256, 1, ("zncc", 11), False, True, "vfit", 2),
    "C5 10000^2 d=[-64,64] census5 + cbca + sgm + wta + vfit (float32, one scale)": lambda e: run(e, *bench.synthetic_pair(10000, 10000, -64, 64), -64, 64, 1, ("census", 5), True, True, "vfit", 1),
}

STAGES = "--stages" in sys.argv

if __name__ == "__main__":
    want = [a for a in sys.argv[1:] if not a.startswith("--")]
    eng = Engine(0)
    if os.environ.get("PMX_BENCH_TRIALS"):  # candidates per volume-sized buffer (pmx_set_placement_trials)
        eng.set_placement_trials(int(os.environ["PMX_BENCH_TRIALS"]))
    out = {}
    for name, fn in CONFIGS.items():
        if want and name.split()[0] not in want:
            continue
        try:
            out[name] = fn(eng)
        except Exception as err:  # e.g. out of device memory at C5
            out[name] = {"error": str(err)[:200]}
        print(json.dumps({name: out[name]}), flush=True)
    eng.close()
