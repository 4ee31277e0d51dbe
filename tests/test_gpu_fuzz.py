"""Seeded random pipelines, GPU (through the C ABI) against the oracle: ragged shapes down to the window size, disparity
ranges inside / across / outside the image, every measure and window, sub-pixel volumes, masks with random conventions,
per-pixel disparity grids, optional CBCA and SGM, both extrema, both refinements - in the lazy (integer fast path) and eager
(float32) modes.  Bit-exact everywhere except ZNCC costs (1e-5, the float64 sliding sums)."""
import numpy as np
import pytest

from tests.cbca_helpers import oracle_cross_supports

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["lazy", "eager"])
def eng(request):
    from pandora_amd.engine import Engine

    e = Engine(0)
    e.set_lazy(request.param == "lazy")
    yield e
    e.close()


def draw(seed):
    rng = np.random.default_rng(seed)
    method = rng.choice(["census", "census", "sad", "ssd", "zncc"])
    win = int(rng.choice([3, 5, 7, 9, 11, 13] if method == "census" else [1, 3, 5, 7, 9, 11]))
    sp = int(rng.choice([1, 1, 2, 4]))
    big = rng.random() < 0.15  # now and then: wide images and disparity ranges beyond every fast-path limit (D up to ~340)
    H = int(rng.integers(win, 80 if big else 34))
    W = int(rng.integers(max(win, 3), 400 if big else 70))
    span = int(rng.integers(0, (340 if big else 40) // sp + 1))
    kind = rng.choice(["inside", "inside", "across", "outside"])
    if kind == "inside":
        dmin = int(rng.integers(-min(W, 30), 5))
    elif kind == "across":
        dmin = int(rng.choice([-W + 2, W - span - 2]))
    else:
        dmin = int(rng.choice([-W - 5 - span, W + 3]))
    dmax = dmin + span
    integer = bool(rng.random() < 0.6)
    base = rng.integers(0, 255, (H, W + 8)).astype(np.float32)
    base = np.floor((base + np.roll(base, 1, 1) + np.roll(base, 1, 0)) / 3.0)
    L = base[:, 4:4 + W].copy()
    R = base[:, 1:1 + W].copy() + rng.integers(-2, 3, (H, W)).astype(np.float32)
    if not integer:
        L += rng.random((H, W)).astype(np.float32)
        R += rng.random((H, W)).astype(np.float32)
    masks = None
    if rng.random() < 0.5:
        valid, nodata = (0, 1) if rng.random() < 0.5 else (5, 7)
        pool = [valid] * 7 + [nodata, valid + nodata + 1]
        masks = (rng.choice(pool, (H, W)).astype(np.int16), rng.choice(pool, (H, W)).astype(np.int16), valid, nodata)
        if rng.random() < 0.4:  # a left mask alone keeps census volumes on the integer fast path (per-pixel intervals)
            masks = (masks[0], None, valid, nodata)
    grids = None
    if rng.random() < 0.3 and span > 0:
        lo = rng.integers(dmin, dmax + 1, (H, W))
        hi = np.minimum(dmax, lo + rng.integers(0, span + 1, (H, W)))
        lo[0, 0], hi[0, 0] = dmin, dmax  # keep the global range
        grids = (lo.astype(np.float64), hi.astype(np.float64))
    cbca = bool(rng.random() < 0.3) and H - 2 * (win // 2) > 0 and W - 2 * (win // 2) - (1 if sp > 1 else 0) > 0
    sgm = bool(rng.random() < 0.5)
    P1 = float(rng.integers(1, 12)) if rng.random() < 0.7 else float(np.float32(rng.random() * 10))
    P2 = P1 + (float(rng.integers(1, 60)) if P1 == int(P1) else float(np.float32(rng.random() * 40)))
    c = dict(method=method, win=win, sp=sp, H=H, W=W, dmin=dmin, dmax=dmax, L=L.astype(np.float32), R=R.astype(np.float32), masks=masks,
             grids=grids, cbca=cbca, cbca_int=float(rng.choice([5.0, 30.0])), cbca_dist=int(rng.integers(2, 7)), sgm=sgm, P1=P1, P2=P2,
             refine=str(rng.choice(["vfit", "quadratic"])), invalid=float(rng.choice([-9999.0, np.nan])),
             overcounting=bool(rng.random() < 0.2))
    if rng.random() < 0.15:  # (drawn last: the other parameters of a seed stay what they were) long arms: 64- and 128-slot rings
        c["cbca_dist"] = int(rng.choice([9, 12, 18, 22, 32]))
    return c


@pytest.mark.parametrize("seed", range(400))
def test_random_pipeline_equals_oracle(eng, oracle, seed):
    c = draw(seed)
    method, win, sp, dmin, dmax, L, R = c["method"], c["win"], c["sp"], c["dmin"], c["dmax"], c["L"], c["R"]
    D = (dmax - dmin) * sp + 1
    is_max = method == "zncc"
    # ---- oracle ----------------------------------------------------------------------------------------------------
    if method == "census":
        ocv = oracle.census_cost(L, R, D, dmin, sp, win)
    elif method == "zncc":
        ocv = oracle.zncc(L, R, D, dmin, sp, win)
    else:
        ocv = oracle.sad_ssd(L, R, D, dmin, sp, win, method == "ssd")
    kw = {}
    if c["masks"]:
        kw.update(mskL=c["masks"][0], mskR=c["masks"][1], valid=c["masks"][2], nodata=c["masks"][3])
    if c["grids"]:
        kw.update(dmin=c["grids"][0], dmax=c["grids"][1])
    oracle.cv_masked(ocv, dmin, sp, win, **kw)
    # ---- device -----------------------------------------------------------------------------------------------------
    eng.set_images(L, R, sp)
    eng.set_masks(*(c["masks"] if c["masks"] else (None, None)))
    eng.set_disparity_grids(*(c["grids"] if c["grids"] else (None, None)))
    cv = eng.alloc_cv(D, dmin)
    if method == "census":
        eng.census(cv, win)
    elif method == "zncc":
        eng.zncc(cv, win)
    else:
        eng.sad_ssd(cv, win, method == "ssd")
    eng.cv_masked(cv, win)
    exact = method != "zncc"
    if not exact:  # continue both sides from the same float32 costs: only the cost kernel itself has a tolerance
        got = cv.to_host()
        np.testing.assert_array_equal(np.isnan(got), np.isnan(ocv))
        np.testing.assert_allclose(got, ocv, rtol=0, atol=1e-5, equal_nan=True)
        ocv = got.copy()
    if c["cbca"]:
        off = win // 2
        msk = c["masks"]
        cl, crs = oracle_cross_supports(oracle, L, R, msk[0] if msk else None, msk[1] if msk else None, sp, off, c["cbca_dist"],
                                        c["cbca_int"], valid=msk[2] if msk else 0)
        oracle.cbca(ocv, dmin, sp, off, cl, crs)
        eng.cbca(cv, off, c["cbca_int"], c["cbca_dist"])
    if c["sgm"]:
        invalid_cost = float(win * win + 1) if method == "census" else float(np.nanmax(np.abs(ocv)) + 1 if np.isfinite(ocv).any() else 1.0)
        ocv = oracle.sgm(ocv, c["P1"], c["P2"], is_max, invalid_cost, c["overcounting"])
        eng.sgm(cv, c["P1"], c["P2"], is_max, invalid_cost, c["overcounting"])
    val0 = np.zeros((c["H"], c["W"]), np.int64)
    eng.set_validity(val0)
    eng.wta(cv, is_max, c["invalid"])
    eng.refine(cv, c["refine"], is_max)
    disp, val, itp = eng.get_disparity(want_itp=True)
    np.testing.assert_array_equal(cv.to_host(), ocv)  # the volume after the last volume step (materialised if it was lazy)
    odisp, oval = oracle.wta(ocv, dmin, sp, is_max, c["invalid"], val0)
    oitp, odisp, oval = oracle.refine(ocv, odisp, oval, dmin, dmin + (D - 1) / sp, sp, is_max, c["refine"])
    np.testing.assert_array_equal(disp, odisp)
    np.testing.assert_array_equal(val, oval)
    np.testing.assert_array_equal(itp, oitp)
    eng.set_masks(None, None)
    eng.set_disparity_grids(None, None)
    cv.free()
