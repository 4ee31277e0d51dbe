#!/usr/bin/env python3
"""Float32 SGM: the eight directions side by side (PMX_SGM_PAR=1) against one after the other (=0), per volume size.
Usage: python tools/bench_sgm_float.py   (spawns itself once per mode: the switch is read from the environment)"""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [(375, 450, 61), (768, 1024, 65), (1024, 1024, 97), (1536, 2048, 65), (2048, 2048, 129)]

if len(sys.argv) > 1:
    from pandora_amd.engine import Engine

    eng = Engine(0)
    eng.set_lazy(False)
    out = {}
    for H, W, D in SHAPES:
        rng = np.random.default_rng(0)
        L = rng.integers(0, 255, (H, W)).astype(np.float32)
        eng.set_images(L, np.roll(L, 3, 1), 1)
        cv = eng.alloc_cv(D, -D + 1)
        eng.census(cv, 5)
        eng.cbca(cv, 2, 30.0, 5)      # float costs: the general path
        eng.sgm(cv, 8.0, 32.0, False, 26.0, False)
        eng.sync()
        eng.census(cv, 5)
        eng.sync()
        t0 = time.perf_counter()
        eng.sgm(cv, 8.0, 32.0, False, 26.0, False)
        eng.sync()
        out[f"{H}x{W}x{D}"] = round((time.perf_counter() - t0) * 1e3, 3)
        cv.free()
    print(json.dumps(out))
else:
    for mode in ("0", "1"):
        r = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, PMX_SGM_PAR=mode), capture_output=True, text=True)
        print("PMX_SGM_PAR=" + mode, r.stdout.strip() or r.stderr[-500:])
