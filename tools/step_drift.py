"""How the headline step's time moves over the life of a process: per-step wall times of N steps (one sync per step), printed as
means over windows.  Asks whether the spread between processes is the buffers' placement or the device warming up.
    python tools/step_drift.py [steps=300] [trials=1] [preheat_ms=0]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pandora_amd.engine import Engine  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 1
preheat = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
H = W = 4096
L, R = bench.synthetic_pair(H, W, 0, 256)
eng = Engine(0)
if trials > 1:
    eng.set_placement_trials(trials)
eng.set_images(L, R, 1)
t0 = time.perf_counter()
while preheat > 0 and (time.perf_counter() - t0) * 1e3 < preheat:
    eng.measure_hbm(1 << 30)
cv = eng.alloc_cv(257, 0)
ts = []
for i in range(steps):
    eng.sync()
    a = time.perf_counter()
    bench.run_pipeline(eng, cv, 5, 8.0, 32.0)
    eng.sync()
    ts.append((time.perf_counter() - a) * 1e3)
ts = np.array(ts)
edges = [0, 2, 7, 27, 50, 100, 200, 300, 500, 1000]
parts = []
for lo, hi in zip(edges[:-1], edges[1:]):
    if lo < len(ts):
        parts.append(f"[{lo}:{min(hi, len(ts))}] {ts[lo:hi].mean():.2f}")
print(f"trials {trials} preheat {preheat:.0f} ms | " + "  ".join(parts) + f" | min {ts.min():.2f}")
