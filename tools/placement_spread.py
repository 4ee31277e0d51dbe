#!/usr/bin/env python3
"""The headline step on fresh contexts of ONE process: how much of the run-to-run spread is where hipMalloc put the volumes, and what
pmx_set_placement_trials takes out of it.  Usage: python tools/placement_spread.py [contexts per setting] [trials ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pandora_amd.engine import Engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
settings = [int(x) for x in sys.argv[2:]] or [1, 6, 8]
H = W = 4096
dmin, dmax = 0, 256
L, R = bench.synthetic_pair(H, W, dmin, dmax)
for trials in settings:
    ms_all = []
    for i in range(n):
        eng = Engine(0)
        eng.set_placement_trials(trials)
        ms, stage, _ = bench.measure_shape(eng, H, W, dmin, dmax, 8, 2, 20260928, pcie=False)
        ms_all.append(ms)
        print(f"trials {trials} context {i}: {ms:.3f} ms  span {stage['sgm_span'][0] / 8:.3f}  wta {stage['wta'][0] / 8:.3f}  cost {stage['census_cost'][0] / 8:.3f}", flush=True)
        eng.close()
    print(f"== trials {trials}: min {min(ms_all):.3f}  median {np.median(ms_all):.3f}  max {max(ms_all):.3f}", flush=True)
