"""Debug helper: compare GPU zncc against the oracle on one small case and print the mismatch pattern."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pandora_amd.engine import Engine
from oracle import capi as oracle
sys.path.insert(0, "tests")
from test_gpu_parity import pair, gpu_cv, cpu_cv

eng = Engine(0)
for (H, W, dmin, dmax, sp, win, integer) in [(30, 44, -6, 3, 1, 5, True), (70, 300, -6, 20, 1, 11, False)]:
    L, R = pair(H, W, seed=5 * H + W, integer=integer)
    L[2:8, 3:12] = 7.0
    got = gpu_cv(eng, "zncc", L, R, dmin, dmax, sp, win).to_host()
    exp = cpu_cv(oracle, "zncc", L, R, dmin, dmax, sp, win)
    nanm = np.isnan(got) != np.isnan(exp)
    print("nan mismatches", nanm.sum())
    if nanm.sum():
        idx = np.argwhere(nanm)
        print(idx[:10], "rows", np.unique(idx[:, 0])[:20], "cols", np.unique(idx[:, 1])[:20], "d", np.unique(idx[:, 2]))
    diff = np.abs(np.nan_to_num(got) - np.nan_to_num(exp))
    print("max diff", diff.max())
    bad = np.argwhere(diff > 1e-5)
    print(len(bad), bad[:10])
    for b in bad[:5]:
        print(b, got[tuple(b)], exp[tuple(b)])
    if len(bad):
        print("rows", np.unique(bad[:, 0]), "cols", np.unique(bad[:, 1]), "d", np.unique(bad[:, 2]))
