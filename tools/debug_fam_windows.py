#!/usr/bin/env python3
"""Placement of the float32 marching kernel's windows (csrc/k_sgmfam.hip; tickets: csrc/pmx_buf.h pmx_take_window): which XCD every window ran on, and - in a
library built with -DPMX_FAM_STATS - how many rows were published with plain / write-through stores and consumed at once / re-read.
Usage: python tools/debug_fam_windows.py [H W D CW]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pandora_amd import _lib  # noqa: E402
from pandora_amd.engine import Engine  # noqa: E402

H, W, D = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 4096, 257)
CW = int(sys.argv[4]) if len(sys.argv) > 4 else 16  # columns per window (C4: 16, C5: 40)
eng = Engine(0)
eng.set_lazy(False)
rng = np.random.default_rng(1)
base = rng.integers(0, 255, (64, W + 16)).astype(np.float32)
L = np.tile(base[:, 8:8 + W], (-(-H // 64), 1))[:H].copy()
R = np.tile(base[:, 5:5 + W], (-(-H // 64), 1))[:H].copy()
eng.set_images(L, R, 1)
cv = eng.alloc_cv(D, 0)
eng.set_option("SGM_SCHED", "fam")
eng.census(cv, 5)
eng.sgm(cv, 8.0, 32.0, False, 26.0, False)
eng.sync()
buf = (C.c_uint * 8192)()
n = _lib.lib().pmx_debug_fam_windows(eng.ctx, buf, 8192)
t = np.array(buf[:n])
print("windows taken of the first eight chunks:", t[:8].tolist())
tab = t[8:]
nwin = (W + H - 2) // CW + 1
x = tab[:nwin].astype(int) - 1
print("windows:", nwin, " XCD of the first 72:", x[:72].tolist())
same = int(np.sum(x[1:] == x[:-1]))
print(f"borders with both sides on one XCD: {same} of {nwin - 1}")
stats = tab[nwin:nwin + 8]
print("stats (plain rows, write-through rows, rows consumed, extra reads):", stats[:4].tolist())
cv.free()
eng.close()
