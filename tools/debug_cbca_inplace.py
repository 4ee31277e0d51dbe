#!/usr/bin/env python3
"""census (float volume) + CBCA through the in-place passes (CBCA_FAST=4: what runs when the second volume cannot be had) against the
default passes, at a given size: where the WTA / quadratic maps differ.  Usage: python tools/debug_cbca_inplace.py [H W dmax]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pandora_amd.engine import Engine  # noqa: E402
from tests.test_gpu_full_size import big_pair, SIZES  # noqa: E402

H, W, dmax = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 4096, 256)
SIZES["DBG"] = (H, W, 0, dmax)
L, R = big_pair("DBG")
maps = {}
for fast in ("1", "4"):
    eng = Engine(0)
    eng.set_lazy(False)
    eng.set_option("CBCA_FAST", fast)
    eng.set_images(L, R, 1)
    cv = eng.alloc_cv(dmax + 1, 0)
    eng.census(cv, 5)
    eng.cbca(cv, 2, 30.0, 5)
    eng.set_validity(None)
    eng.wta(cv, False, -9999.0)
    eng.refine(cv, "quadratic", False)
    maps[fast] = eng.get_disparity(want_itp=True)
    cv.free()
    eng.close()
for name, a, b in zip(("disp", "validity", "itp"), maps["1"], maps["4"]):
    bad = ~((a == b) | (np.isnan(a.astype(np.float64)) & np.isnan(b.astype(np.float64))))
    print(name, "differ:", int(bad.sum()), [(int(r), int(c), a[r, c], b[r, c]) for r, c in zip(*np.nonzero(bad))][:12])
