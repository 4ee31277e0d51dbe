import sys, numpy as np
sys.path.insert(0, '.')
from pandora_amd.engine import Engine
from oracle import capi
from tests.test_gpu_parity import pair
DIRS = [(0, 1), (0, -1), (1, 0), (-1, 0), (1, 1), (-1, -1), (1, -1), (-1, 1)]
def path(C, dr, dc, P1, P2):
    H, W, D = C.shape
    L = np.zeros_like(C)
    rs = range(H) if dr >= 0 else range(H - 1, -1, -1)
    cs = range(W) if dc >= 0 else range(W - 1, -1, -1)
    for r in rs:
        for c in cs:
            pr, pc = r - dr, c - dc
            if pr < 0 or pr >= H or pc < 0 or pc >= W:
                L[r, c] = C[r, c]; continue
            q = L[pr, pc]; M = q.min()
            lo = np.concatenate([[np.inf], q[:-1]]); hi = np.concatenate([q[1:], [np.inf]])
            L[r, c] = C[r, c] + (np.minimum(np.minimum(q, np.minimum(lo, hi) + P1), M + P2) - M)
    return L
for (H, W, dmin, dmax, win) in [(9, 11, -2, 2, 3), (24, 37, -6, 3, 5)]:
    L, R = pair(H, W, seed=H + W)
    D = dmax - dmin + 1
    eng = Engine(0); eng.set_images(L, R, 1)
    cv = eng.alloc_cv(D, dmin); eng.census(cv, win)
    eng.sgm(cv, 8, 32, False, float(win * win + 1), False)
    got = eng.debug_path_costs(cv).astype(np.float32)
    C = capi.census_cost(L, R, D, dmin, 1, win); C = np.where(np.isnan(C), win * win + 1, C).astype(np.float32)
    for k, (dr, dc) in enumerate(DIRS):
        exp = path(C, dr, dc, 8, 32)
        bad = got[k] != exp
        print((H, W), "dir", k, (dr, dc), "bad", bad.sum(), "rows", np.unique(np.where(bad)[0])[:12], "cols", np.unique(np.where(bad)[1])[:12])
    eng.close()
H, W, dmin, dmax, win = 9, 11, -2, 2, 3
L, R = pair(H, W, seed=H + W); D = dmax - dmin + 1
eng = Engine(0); eng.set_images(L, R, 1); cv = eng.alloc_cv(D, dmin); eng.census(cv, win)
eng.sgm(cv, 8, 32, False, float(win * win + 1), False)
got = eng.debug_path_costs(cv).astype(np.float32)
C = capi.census_cost(L, R, D, dmin, 1, win); C = np.where(np.isnan(C), win * win + 1, C).astype(np.float32)
exp = path(C, 1, 1, 8, 32)
bad = (got[4] != exp).any(axis=2)
print(bad.astype(int))
for (r, c) in [(1, 0), (1, 1), (2, 1), (1, 5), (2, 6)]:
    print((r, c), "got", got[4][r, c], "exp", exp[r, c], "C", C[r, c], "prev exp", exp[r-1, c-1] if c > 0 else None)
