#!/usr/bin/env python3
"""Census pipelines at full size through the two independent implementations of the path - lazy mode (census codes, integer kernels,
marching CBCA, pending SGM family) and eager mode (float32 volumes between the steps, the reference's way) - on combinations the
parity suite only runs at small sizes: multi-word census windows, per-pixel disparity grids + a left mask, D = 300, D = 512, CBCA with
D = 257, the fast right side (reverse_cost_volume) and its WTA.  Every map must be identical bit for bit.
Usage (GPU box): python tools/full_size_lazy_vs_eager.py [H W]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pandora_amd.engine import Engine  # noqa: E402
from tests.test_gpu_full_size import big_pair, SIZES, _PAIRS  # noqa: E402

H, W = (int(v) for v in sys.argv[1:3]) if len(sys.argv) > 2 else (4096, 4096)


def run(lazy, L, R, dmin, dmax, win, grids=None, mask=None, cbca=False, sgm=True, right=False, P=(8.0, 32.0)):
    eng = Engine(0)
    try:
        eng.set_lazy(lazy)
        eng.set_images(L, R, 1)
        if mask is not None:
            eng.set_masks(mask, None, 0, 1)
        if grids is not None:
            eng.set_disparity_grids(*grids)
        cv = eng.alloc_cv(dmax - dmin + 1, dmin)
        eng.census(cv, win)
        if grids is not None or mask is not None:
            eng.cv_masked(cv, win)
        if cbca:
            eng.cbca(cv, win // 2, 30.0, 5)
        if sgm:
            eng.sgm(cv, P[0], P[1], False, float(win * win + 1), False)
        out = []
        eng.set_validity(None)
        eng.wta(cv, False, -9999.0)
        eng.refine(cv, "vfit", False)
        out += list(eng.get_disparity(want_itp=True))
        if right:
            rcv = eng.reverse_cost_volume(cv, -dmax)
            eng.set_validity(None)
            eng.wta(rcv, False, -9999.0)
            out += list(eng.get_disparity())
            rcv.free()
        cv.free()
        return out
    finally:
        eng.close()


def same(a, b):
    return int((~((a == b) | (np.isnan(a.astype(np.float64)) & np.isnan(b.astype(np.float64))))).sum())


rng = np.random.default_rng(7)
cases = [
    ("census 9x9 (three code words), D = 257, SGM", dict(dmin=0, dmax=256, win=9)),
    ("census 13x13 (six code words), D = 129, SGM", dict(dmin=-64, dmax=64, win=13)),
    ("census 5x5, D = 300, SGM", dict(dmin=-150, dmax=149, win=5)),
    ("census 5x5, D = 512 (float32 SGM either way), SGM", dict(dmin=0, dmax=511, win=5, P=(8.0, 32.0))),
    ("census 5x5, per-pixel grids + left mask, D = 257, SGM", dict(dmin=0, dmax=256, win=5, grids=True, mask=True)),
    ("census 5x5 + CBCA, D = 257 (no marching CBCA: D > 256), SGM", dict(dmin=0, dmax=256, win=5, cbca=True)),
    ("census 5x5 + CBCA, D = 129, no SGM, fast right side", dict(dmin=-64, dmax=64, win=5, cbca=True, sgm=False, right=True)),
    ("census 5x5, D = 257, SGM, fast right side", dict(dmin=0, dmax=256, win=5, right=True)),
    ("census 3x3, fractional penalties (float32 SGM either way), D = 129", dict(dmin=-64, dmax=64, win=3, P=(1.5, 7.25))),
]
bad = 0
for label, kw in cases:
    dmin, dmax = kw["dmin"], kw["dmax"]
    SIZES["DBG"] = (H, W, dmin, dmax)
    _PAIRS.clear()  # (big_pair caches by name: a pair made for this case's range)
    L, R = big_pair("DBG")
    k = dict(kw)
    if k.pop("grids", False):
        lo = rng.integers(dmin, dmin + 40, (H, W)).astype(np.float64)
        hi = lo + rng.integers(60, dmax - dmin - 40, (H, W))
        hi = np.minimum(hi, dmax)
        k["grids"] = (lo, hi)
    if k.pop("mask", False):
        m = (rng.random((H, W)) < 0.02).astype(np.int16)
        k["mask"] = m
    t0 = time.time()
    try:
        a = run(True, L, R, **k)
        b = run(False, L, R, **k)
        diffs = [same(x, y) for x, y in zip(a, b)]
    except Exception as e:  # noqa: BLE001 - reported
        diffs = f"{type(e).__name__}: {e}"[:200]
    ok = isinstance(diffs, list) and not any(diffs)
    bad += 0 if ok else 1
    print(f"{'ok ' if ok else 'BAD'} {label}: mismatching (disp, validity, itp[, right disp, right validity]) = {diffs}  ({time.time() - t0:.0f} s)", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
