#!/usr/bin/env python3
"""Random pipelines through the direction-family form of the integer path (k_sgmfam8.hip + the horizontal-pair kernels, forced with
PMX_SGM8_FAM=1) against the CPU oracle: random image shapes (down to a few pixels, up to several windows in flight), disparity
ranges inside / across / outside the image, census windows 3 / 5 / 7, integer penalties, both window widths, optional per-pixel
disparity grids; the summed volume, the WTA and the refinement must be identical.  Usage: python tools/fuzz_fam8.py [seed0] [n]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PMX_SGM8_FAM"] = "1"
from oracle import capi  # noqa: E402  (checker)
from pandora_amd.engine import Engine  # noqa: E402


def main():
    seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    capi.lib()
    capi.set_threads(0)
    eng = Engine(0)
    t0, cells, ran = time.time(), 0, 0
    for seed in range(seed0, seed0 + n):
        rng = np.random.default_rng(seed)
        H = int(rng.choice([2, 3, 5, 17, 40, 97, 150, 260]))
        W = int(rng.choice([6, 16, 31, 33, 64, 100, 257, 420, 700]))
        win = int(rng.choice([3, 5, 5, 7]))
        D = int(rng.choice([1, 3, 16, 17, 61, 64, 65, 129, 200, 257, 300, 319]))
        dmin = int(rng.integers(-D - 5, 10))
        dmax = dmin + D - 1
        P1 = int(rng.integers(1, 12))
        P2 = int(rng.integers(P1 + 1, P1 + 40))
        ic = win * win + 1
        if 3 * (ic + P2) > 255:
            P2 = max(P1 + 1, 255 // 3 - ic)
            if P2 <= P1:
                continue
        os.environ["PMX_SGM8_FAM_NW"] = str(rng.choice([4, 8]))
        # the horizontal pair's one-sided / two-sided walk on the cost volume / row-per-wavefront walk from the census words; the
        # Hamming costs from a cost volume or made inside the SGM kernels (round 4), for the marching kernel alone or for both
        os.environ["PMX_SGM8_HPAIR"] = str(rng.choice([1, 2, 3, 3]))
        # ... and the WTA of the three volumes by the lean kernel (default) or the general one
        for k, v in (("PMX_SGM8_CODES", rng.choice(["", "0", "1"])), ("PMX_SGM8_FAMCODES", rng.choice(["", "", "0", "1"])),
                     ("PMX_WTA3", rng.choice(["", "", "", "0"]))):
            if v:
                os.environ[k] = str(v)
            else:
                os.environ.pop(k, None)
        eng.options_from_env()  # (the library reads its environment once, at pmx_create)
        base = rng.integers(0, 256, (H, W + 8)).astype(np.float32)
        base = np.floor((base + np.roll(base, 1, 1) + np.roll(base, 1, 0)) / 3.0)
        L = base[:, 4:4 + W].copy()
        R = (base[:, 1:1 + W] + rng.integers(-2, 3, (H, W))).astype(np.float32)
        grids = None
        if rng.random() < 0.3 and D > 4:
            gmin = rng.integers(dmin, dmin + D // 2, (H, W)).astype(np.float64)
            gmax = rng.integers(dmin + D // 2, dmax + 1, (H, W)).astype(np.float64)
            grids = (gmin, gmax)
        eng.set_images(L, R, 1)
        eng.set_disparity_grids(*(grids if grids else (None, None)))
        cv = eng.alloc_cv(D, dmin)
        eng.census(cv, win)
        if grids:
            eng.cv_masked(cv, win)
        eng.sgm(cv, P1, P2, False, float(ic), False)
        try:
            eng.debug_path_costs(cv, raw=True)
            took_family = False
        except Exception:
            took_family = True
        eng.set_validity(None)
        eng.wta(cv, False, -9999.0)
        eng.refine(cv, "vfit", False)
        disp, val, itp = eng.get_disparity(want_itp=True)
        vol = cv.to_host()
        cv.free()
        cpu = capi.census_cost(L, R, D, dmin, 1, win)
        if grids:
            capi.cv_masked(cpu, dmin, 1, win, dmin=grids[0], dmax=grids[1])
        ref = capi.sgm(cpu, P1, P2, False, float(ic), False)
        rdisp, rval = capi.wta(ref, dmin, 1, False, -9999.0)
        ritp, rdisp2, rval2 = capi.refine(ref, rdisp, rval, dmin, dmax, 1, False, "vfit")
        ok = (np.array_equal(vol, ref, equal_nan=True) and np.array_equal(disp, rdisp2, equal_nan=True) and np.array_equal(val, rval2)
              and np.array_equal(itp, ritp, equal_nan=True))
        if not ok:
            print(f"DIFFERENCE seed {seed}: H={H} W={W} win={win} d=[{dmin},{dmax}] P1={P1} P2={P2} nw={os.environ['PMX_SGM8_FAM_NW']} "
                  f"hpair={os.environ['PMX_SGM8_HPAIR']} codes={os.environ.get('PMX_SGM8_CODES')} famcodes={os.environ.get('PMX_SGM8_FAMCODES')} "
                  f"wta3={os.environ.get('PMX_WTA3')} "
                  f"grids={grids is not None} family={took_family} volume_equal={np.array_equal(vol, ref, equal_nan=True)}", flush=True)
            sys.exit(1)
        cells += H * W * D
        ran += took_family
    eng.set_disparity_grids(None, None)
    eng.close()
    print(f"fuzz_fam8: seeds {seed0}..{seed0 + n - 1}: {ran} pipelines through the family form, no difference "
          f"({cells / 1e6:.0f} Mcells, {time.time() - t0:.0f} s)")


if __name__ == "__main__":
    main()
