import sys, os, json
sys.path.insert(0, "/root/repo")
import numpy as np
import bench
from pandora_amd.engine import Engine
eng = Engine(0); eng.set_profiling(True)
for (H, W, dmax) in ((2048, 2048, 128), (3000, 4096, 256)):
    L, R = bench.synthetic_pair(H, W, 0, dmax, seed=1)
    for subpix in (1, 2):
        eng.set_images(L, R, subpix)
        cv = eng.alloc_cv(dmax * subpix + 1, 0)
        out = {}
        for win in (3, 5, 7, 9, 11, 13):
            eng.zncc(cv, win); eng.sync()
            eng.reset_stage_times()
            for _ in range(3):
                eng.zncc(cv, win)
            eng.sync()
            out[win] = round(eng.stage_time("zncc")[0] / 3, 3)
        print(H, W, dmax, "subpix", subpix, out)
        cv.free()
