import numpy as np, sys, traceback
sys.path.insert(0, '.')
from pandora_amd.engine import Engine
from oracle import capi as orc
eng = Engine(0)
rng = np.random.default_rng(1)
def pair(H, W):
    base = rng.integers(0, 255, (H, W + 8)).astype(np.float32)
    return base[:, 4:4+W].copy(), base[:, 1:1+W].copy()
def run(name, fn):
    try:
        r = fn()
        print(name, "OK" if r else "MISMATCH")
    except Exception as e:
        print(name, "ERR", str(e)[:160])
for lazy in (True, False):
    eng.set_lazy(lazy)
    tag = "lazy" if lazy else "eager"
    def sgm_case(H, W, dmin, dmax, win, P1, P2):
        L, R = pair(H, W)
        D = dmax - dmin + 1
        eng.set_images(L, R, 1)
        cv = eng.alloc_cv(D, dmin); eng.census(cv, win)
        inv = float(win*win+1)
        eng.sgm(cv, P1, P2, False, inv, False)
        got = cv.to_host()
        exp = orc.sgm(orc.census_cost(L, R, D, dmin, 1, win), P1, P2, False, inv, False)
        cv.free()
        return np.array_equal(got, exp, equal_nan=True)
    run(f"{tag} sgm D=512", lambda: sgm_case(12, 600, -511, 0, 5, 8., 32.))
    run(f"{tag} sgm D=400 win13", lambda: sgm_case(20, 500, -399, 0, 13, 8., 32.))
    run(f"{tag} sgm P2=1e6", lambda: sgm_case(20, 90, -20, 5, 5, 3., 1e6))
    run(f"{tag} sgm P float", lambda: sgm_case(20, 90, -20, 5, 7, 0.5, 0.75))
    run(f"{tag} sgm tall thin", lambda: sgm_case(700, 9, -3, 3, 3, 8., 32.))
    run(f"{tag} sgm 1 row", lambda: sgm_case(5, 300, -30, 0, 5, 8., 32.))
    def med(size):
        H, W = 50, 70
        d = (rng.integers(-30, 5, (H, W)) + rng.random((H, W))).astype(np.float32)
        v = np.where(rng.random((H, W)) < 0.2, 1, 0).astype(np.int64)
        return np.array_equal(eng.median_filter_disparity(d, v, size), orc.filter_median_disparity(d, v, size), equal_nan=True)
    for s in (7, 15, 31):
        run(f"{tag} median {s}", lambda s=s: med(s))
    def bil(ss):
        H, W = 60, 80
        d = (rng.integers(-30, 5, (H, W)) + rng.random((H, W))).astype(np.float32)
        v = np.zeros((H, W), np.int64)
        return np.allclose(eng.bilateral_filter_disparity(d, v, 2.0, ss), orc.filter_bilateral_disparity(d, v, 2.0, ss), rtol=1e-6, equal_nan=True)
    for ss in (10.0, 19.0):
        run(f"{tag} bilateral sigma_space {ss}", lambda ss=ss: bil(ss))
    def zn(win, sp):
        L, R = pair(40, 120)
        D = 20*sp+1
        eng.set_images(L, R, sp); cv = eng.alloc_cv(D, -10); eng.zncc(cv, win)
        got = cv.to_host(); exp = orc.zncc(L, R, D, -10, sp, win); cv.free()
        return np.allclose(got, exp, atol=1e-5, equal_nan=True) and np.array_equal(np.isnan(got), np.isnan(exp))
    for w, sp in ((13, 1), (15, 1), (21, 2), (1, 4)):
        run(f"{tag} zncc win {w} sp {sp}", lambda w=w, sp=sp: zn(w, sp))
    def sad(win, sp):
        L, R = pair(40, 120)
        D = 20*sp+1
        eng.set_images(L, R, sp); cv = eng.alloc_cv(D, -10); eng.sad_ssd(cv, win, False)
        got = cv.to_host(); exp = orc.sad_ssd(L, R, D, -10, sp, win, False); cv.free()
        return np.array_equal(got, exp, equal_nan=True)
    for w, sp in ((9, 1), (15, 1), (21, 2)):
        run(f"{tag} sad win {w} sp {sp}", lambda w=w, sp=sp: sad(w, sp))
