"""Mid-size random pipelines (images up to 200 x 500, D up to 140) with the kernel-choice hooks drawn at random, so that the paths
the small-image fuzz never reaches are compared with the oracle too: the whole-row CBCA pass H with 1..4 rows per workgroup, pass V
through pointers / buffers / 512-thread workgroups, census costs inside pass H with and without valid intervals, census + CBCA as one
marching kernel, the float32 SGM
schedules (one after the other, side by side, marching families), the integer path with odd lane maps.  FUZZ_FROM / FUZZ_TO."""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402

from oracle import capi as orc  # noqa: E402
from pandora_amd.engine import Engine  # noqa: E402
from tests.cbca_helpers import oracle_cross_supports  # noqa: E402

HOOKS = {"PMX_CBCA_VBUF": ["0", "1"], "PMX_CBCA_VBS": ["256", "512"], "PMX_CBCA_ROWS": [None, "1", "2", "3"],
         "PMX_CBCA_GEO": [None, "0"], "PMX_SGM_SCHED": [None, "seq", "par", "fam"], "PMX_SGM_PENDING": [None, "0"],
         "PMX_SGM_HFUSED": [None, "0"], "PMX_COST5": [None, "0"], "PMX_CBCA_MARCH": [None, None, "0"]}


def one(seed):
    rng = np.random.default_rng(seed)
    hooks = {k: rng.choice(np.array(v, dtype=object)) for k, v in HOOKS.items()}
    for k, v in hooks.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    method = str(rng.choice(["census", "census", "sad"]))
    win = int(rng.choice([3, 5, 7] if method == "census" else [1, 3, 5]))
    H, W = int(rng.integers(40, 200)), int(rng.integers(60, 500))
    D = int(rng.integers(10, 140))
    dmin = int(rng.integers(-D, 1))
    dmax = dmin + D - 1
    base = rng.integers(0, 255, (H, W + 8)).astype(np.float32)
    base = np.floor((base + np.roll(base, 1, 1) + np.roll(base, 1, 0)) / 3.0)
    L, R = base[:, 4:4 + W].copy(), base[:, 1:1 + W].copy() + rng.integers(-2, 3, (H, W)).astype(np.float32)
    mskL = rng.choice([0, 0, 0, 0, 0, 0, 0, 0, 1, 2], (H, W)).astype(np.int16) if rng.random() < 0.3 else None
    grids = None
    if rng.random() < 0.3:
        lo = rng.integers(dmin, dmin + 4, (H, W)).astype(np.float64)
        hi = rng.integers(dmax - 4, dmax + 1, (H, W)).astype(np.float64)
        lo[0, 0], hi[0, 0] = dmin, dmax
        grids = (lo, hi)
    cbca = rng.random() < 0.5
    dist = int(rng.choice([2, 3, 5, 5, 5, 9, 12]))
    sgm = rng.random() < 0.6
    P1 = float(rng.integers(1, 12))
    P2 = P1 + float(rng.integers(1, 40))
    lazy = bool(rng.random() < 0.6)
    eng = Engine(0)
    try:
        eng.set_lazy(lazy)
        eng.set_images(L, R, 1)
        eng.set_masks(mskL, None, 0, 1) if mskL is not None else eng.set_masks(None, None)
        eng.set_disparity_grids(*(grids if grids else (None, None)))
        cv = eng.alloc_cv(D, dmin)
        if method == "census":
            eng.census(cv, win)
            ocv = orc.census_cost(L, R, D, dmin, 1, win)
        else:
            eng.sad_ssd(cv, win, False)
            ocv = orc.sad_ssd(L, R, D, dmin, 1, win, False)
        eng.cv_masked(cv, win)
        kw = {}
        if mskL is not None:
            kw.update(mskL=mskL, mskR=None, valid=0, nodata=1)
        if grids:
            kw.update(dmin=grids[0], dmax=grids[1])
        orc.cv_masked(ocv, dmin, 1, win, **kw)
        if cbca:
            off = win // 2
            cl, crs = oracle_cross_supports(orc, L, R, mskL, None, 1, off, dist, 30.0, valid=0)
            orc.cbca(ocv, dmin, 1, off, cl, crs)
            eng.cbca(cv, off, 30.0, dist)
        if sgm:
            inv = float(win * win + 1) if method == "census" else float(np.nanmax(np.abs(ocv)) + 1)
            ocv = orc.sgm(ocv, P1, P2, False, inv, False)
            eng.sgm(cv, P1, P2, False, inv, False)
        val0 = np.zeros((H, W), np.int64)
        eng.set_validity(val0)
        eng.wta(cv, False, -9999.0)
        eng.refine(cv, "vfit", False)
        disp, val, itp = eng.get_disparity(want_itp=True)
        np.testing.assert_array_equal(cv.to_host(), ocv)
        odisp, oval = orc.wta(ocv, dmin, 1, False, -9999.0, val0)
        oitp, odisp, oval = orc.refine(ocv, odisp, oval, dmin, dmax, 1, False, "vfit")
        np.testing.assert_array_equal(disp, odisp)
        np.testing.assert_array_equal(val, oval)
        np.testing.assert_array_equal(itp, oitp)
    finally:
        eng.close()
    return dict(method=method, win=win, H=H, W=W, D=D, dmin=dmin, cbca=cbca, dist=dist, sgm=sgm, lazy=lazy, mask=mskL is not None,
                grids=grids is not None, hooks={k: v for k, v in hooks.items() if v is not None})


fails = 0
for seed in range(int(os.environ.get("FUZZ_FROM", "0")), int(os.environ.get("FUZZ_TO", "60"))):
    try:
        one(seed)
    except Exception as e:  # noqa: BLE001
        fails += 1
        print("FAIL", seed, type(e).__name__, str(e)[:400].replace("\n", " "))
        if fails > 6:
            break
print("done, failures:", fails)
