import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import bench
from pandora_amd.engine import Engine
eng = Engine(0)
for (H, W, dmax) in ((2048, 2048, 128), (4096, 4096, 256)):
    L, R = bench.synthetic_pair(H, W, 0, dmax, seed=1)
    eng.set_images(L, R, 1)
    cv = eng.alloc_cv(dmax + 1, 0)
    for _ in range(2):
        bench.run_pipeline(eng, cv, 5, 8.0, 32.0)
    host_out = eng.get_disparity(want_itp=True)
    eng.sync()
    for rep in range(3):
        t = [time.perf_counter()]
        eng.set_images(L, R, 1); t.append(time.perf_counter())
        eng.sync(); t.append(time.perf_counter())
        bench.run_pipeline(eng, cv, 5, 8.0, 32.0); t.append(time.perf_counter())
        eng.sync(); t.append(time.perf_counter())
        eng.get_disparity(want_itp=True, out=host_out); t.append(time.perf_counter())
        print(H, "set_images %.2f  (dma wait %.2f)  launch %.2f  device %.2f  download %.2f  total %.2f" % tuple(
            [(t[i + 1] - t[i]) * 1e3 for i in range(5)] + [(t[5] - t[0]) * 1e3]))
    for rep in range(2):
        t0 = time.perf_counter()
        eng.set_images(L, R, 1)
        bench.run_pipeline(eng, cv, 5, 8.0, 32.0)
        eng.get_disparity(want_itp=True, out=host_out)
        print(H, "unsynchronised total %.2f" % ((time.perf_counter() - t0) * 1e3))
    cv.free()
