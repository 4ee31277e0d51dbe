#!/usr/bin/env python3
"""Device-memory footprint over many pipelines of changing shapes (volumes allocated and freed, engines created and closed):
free memory reported by rocm-smi must come back.  Usage: python tools/leak_probe.py"""
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pandora_amd.engine import Engine


import ctypes

_hip = ctypes.CDLL("libamdhip64.so")


def used_mb():
    """device memory in use as the HIP runtime of THIS process sees it"""
    free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
    _hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total))
    return (total.value - free.value) / 2**20


rng = np.random.default_rng(0)
print("start", round(used_mb()), "MB")
for cycle in range(3):
    eng = Engine(0)
    for it in range(40):
        H, W, D = int(rng.integers(200, 1200)), int(rng.integers(200, 1500)), int(rng.integers(20, 200))
        L = rng.integers(0, 255, (H, W)).astype(np.float32)
        eng.set_lazy(bool(it & 1))
        eng.set_images(L, np.roll(L, 3, 1), 1)
        cv = eng.alloc_cv(D, -D + 1)
        eng.census(cv, 5)
        if it % 3 == 0:
            eng.cbca(cv, 2, 30.0, 5)
        eng.sgm(cv, 8.0, 32.0, False, 26.0, False)
        eng.set_validity(None)
        eng.wta(cv, False, -9999.0)
        eng.refine(cv, "vfit", False)
        eng.get_disparity(want_itp=True)
        if it % 4 == 0:
            r = eng.reverse_cost_volume(cv, -5)
            r.free()
        cv.free()
    print("cycle", cycle, "engine open:", round(used_mb()), "MB", flush=True)
    eng.close()
    print("cycle", cycle, "engine closed:", round(used_mb()), "MB", flush=True)
