import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
from pandora_amd.engine import Engine
from pandora_amd import _lib
eng = Engine(0)
for dmax in (255, 256, 287):
    L, R = bench.synthetic_pair(4096, 4096, 0, dmax)
    eng.set_images(L, R, 1)
    D = dmax + 1
    cv = eng.alloc_cv(D, 0)
    def step():
        eng.zncc(cv, 11); eng.sgm(cv, 8.0, 32.0, True, 2.0, False); eng.set_validity(None); eng.wta(cv, True, -9999.0); eng.refine(cv, "vfit", True)
    step(); eng.sync()
    eng.set_profiling(True); eng.reset_stage_times()
    t0 = time.perf_counter()
    for _ in range(3): step()
    eng.sync()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    st = {k: round(eng.stage_time(k)[0] / 3, 2) for k in _lib.STAGES if eng.stage_time(k)[1]}
    eng.set_profiling(False)
    print(D, round(ms, 2), round(ms / D * 257, 2), st, flush=True)
    cv.free()
