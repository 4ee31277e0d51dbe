import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
import bench
from pandora_amd.engine import Engine
eng = Engine(0); eng.set_profiling(True)
H, W, dmin, dmax = (int(x) for x in sys.argv[1:5])
L, R = bench.synthetic_pair(H, W, 0, dmax - dmin, seed=1)
eng.set_images(L, R, 1)
cv = eng.alloc_cv(dmax - dmin + 1, dmin)
for rep in range(3):
    eng.census(cv, 5)
    eng.reset_stage_times()
    eng.cbca(cv, 2, 5.0, 5)
    eng.sync()
print(os.environ.get("PMX_CBCA_DBG"), {k: round(eng.stage_time(k)[0], 3) for k in ("cbca_arms", "cbca_h", "cbca_v")})
