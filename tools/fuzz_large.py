#!/usr/bin/env python3
"""Random LARGE census pipelines (volumes of 0.5 - 20 GB: offsets beyond 2^31 and 2^32 bytes, sizes no test shape has) through the two
independent implementations of the path - lazy mode with a random set of forced kernel routes against eager mode with the library's
own choices - every map compared bit for bit.  The oracle is too slow at these sizes; two implementations that share no kernel on the
census -> CBCA -> SGM -> WTA -> refinement path are the check (tools/fuzz_mid.py does the same draw against the oracle at small sizes).
Usage (GPU box): python tools/fuzz_large.py [first seed] [count]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pandora_amd.engine import Engine  # noqa: E402
from tests.test_gpu_full_size import big_pair, SIZES, _PAIRS  # noqa: E402

seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40

ROUTES = {  # option -> values a lazy run may be forced onto
    "SGM8_FAM": ["0", "1"], "SGM8_HPAIR": ["1", "2", "3"], "SGM8_CODES": ["0", "1"], "COST5": ["0"], "WTA3": ["0"], "SGM8": ["0"],
    "SGM_SCHED": ["seq", "fam"], "SGM_HFUSED": ["0"], "SGM_PENDING": ["0"], "SGM_FAM_PAR": ["1"], "SGM_FAM_XCD": ["1", "4"],
    "SGM8_FAM_XCD": ["1", "4"], "CBCA_MARCH": ["0"], "CBCA_FAST": ["0", "2", "4"], "CBCA_VBUF": ["0"], "CBCA_SIGN": ["0"],
    "CBCA_ROWS": ["1", "2"], "CBCA_ARMS_FLAT": ["0"],
}


TRACE = bool(os.environ.get("FUZZ_TRACE"))  # a line (and a sync) after every step: which call a GPU memory fault belongs to


def run(lazy, opts, L, R, dmin, dmax, win, cbca, sgm, P, grids, mask, method):
    eng = Engine(0)

    def mark(name):
        if TRACE:
            eng.sync()
            print(f"    [{'lazy' if lazy else 'eager'}] {name}", flush=True)

    try:
        eng.set_lazy(lazy)
        for k, v in opts.items():
            eng.set_option(k, v)
        eng.set_images(L, R, 1)
        mark('set_images')
        if mask is not None:
            eng.set_masks(mask, None, 0, 1)
            mark('set_masks')
        if grids is not None:
            eng.set_disparity_grids(*grids)
            mark('set_grids')
        cv = eng.alloc_cv(dmax - dmin + 1, dmin)
        mark('alloc_cv')
        eng.census(cv, win)
        mark('census')
        if grids is not None or mask is not None:
            eng.cv_masked(cv, win)
            mark('cv_masked')
        if cbca:
            eng.cbca(cv, win // 2, 30.0, cbca)
            mark('cbca')
        if sgm:
            eng.sgm(cv, P[0], P[1], False, float(win * win + 1), False)
            mark('sgm')
        eng.set_validity(None)
        eng.wta(cv, False, -9999.0)
        mark('wta')
        eng.refine(cv, method, False)
        mark('refine')
        out = eng.get_disparity(want_itp=True)
        cv.free()
        mark('free')
        return out
    finally:
        eng.close()


bad = 0
t_all = time.time()
for seed in range(seed0, seed0 + count):
    rng = np.random.default_rng(seed)
    H, W = int(rng.integers(600, 3200)), int(rng.integers(900, 5200))
    D = int(rng.choice([rng.integers(20, 70), rng.integers(70, 140), rng.integers(140, 300), 257, 129, 65, 256, 128]))
    while H * W * D * 4 > 20e9:
        H = H * 3 // 4
    dmin = int(rng.integers(-D, 10))
    dmax = dmin + D - 1
    win = int(rng.choice([3, 5, 5, 5, 7, 9, 11, 13]))
    cbca = int(rng.choice([0, 0, 3, 5, 8]))
    sgm = bool(rng.random() < 0.75) or not cbca
    P = (8.0, 32.0) if rng.random() < 0.7 else (float(rng.integers(1, 9)) + 0.5, float(rng.integers(10, 40)) + 0.25)
    method = str(rng.choice(["vfit", "quadratic"]))
    SIZES["DBG"] = (H, W, dmin, dmax)
    _PAIRS.clear()  # (big_pair caches by name)
    L, R = big_pair("DBG")
    assert L.shape == (H, W)
    grids = mask = None
    if rng.random() < 0.25:
        lo = rng.integers(dmin, dmin + max(2, D // 6), (H, W)).astype(np.float64)
        hi = np.minimum(lo + rng.integers(max(2, D // 3), D, (H, W)), dmax).astype(np.float64)
        grids = (lo, hi)
    if rng.random() < 0.2:
        mask = (rng.random((H, W)) < 0.02).astype(np.int16)
    opts = {k: str(rng.choice(v)) for k, v in ROUTES.items() if rng.random() < 0.3}
    label = f"seed {seed}: {H} x {W} x {D} ({H * W * D * 4 / 1e9:.1f} GB) win {win} cbca {cbca} sgm {sgm} P {P} {method} grids {grids is not None} mask {mask is not None} routes {opts}"
    t0 = time.time()
    try:
        a = run(True, opts, L, R, dmin, dmax, win, cbca, sgm, P, grids, mask, method)
        b = run(False, {}, L, R, dmin, dmax, win, cbca, sgm, P, grids, mask, method)
        diffs = [int((~((x == y) | (np.isnan(x.astype(np.float64)) & np.isnan(y.astype(np.float64))))).sum()) for x, y in zip(a, b)]
        ok = not any(diffs)
    except Exception as e:  # noqa: BLE001 - reported
        diffs, ok = f"{type(e).__name__}: {e}"[:160], False
    bad += 0 if ok else 1
    print(f"{'ok ' if ok else 'BAD'} {label}: {diffs} ({time.time() - t0:.0f} s)", flush=True)
print(f"fuzz_large: seeds {seed0}..{seed0 + count - 1}, failures: {bad} ({time.time() - t_all:.0f} s)")
sys.exit(1 if bad else 0)
