#!/usr/bin/env python3
"""End-to-end time of whole pipelines through the reference-shaped API (PandoraMachine + plugins) at a BASELINE-sized
pair, per step.  Usage: python tools/bench_machine.py [H W dmin dmax]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import pandora_amd  # noqa: E402
from pandora_amd import runtime  # noqa: E402
from pandora_amd.dataset import make_image  # noqa: E402
from pandora_amd.state_machine import PandoraMachine  # noqa: E402

H, W, dmin, dmax = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (2048, 2048, -128, 0)
L, R = bench.synthetic_pair(H, W, 0, dmax - dmin)
L, R = R, L  # Pandora's convention needs negative disparities for this pair
CFGS = {
    "census+sgm+wta+vfit": {"matching_cost": {"matching_cost_method": "census", "window_size": 5},
                            "optimization": {"optimization_method": "sgm", "penalty": {"P1": 8, "P2": 32}},
                            "disparity": {"disparity_method": "wta", "invalid_disparity": "NaN"},
                            "refinement": {"refinement_method": "vfit"}},
    "a_semi_global_matching.json": {"matching_cost": {"matching_cost_method": "census", "window_size": 5, "subpix": 1},
                                    "optimization": {"optimization_method": "sgm", "penalty": {"P1": 8, "P2": 32}},
                                    "disparity": {"disparity_method": "wta", "invalid_disparity": "NaN"},
                                    "refinement": {"refinement_method": "vfit"},
                                    "filter": {"filter_method": "median", "filter_size": 3},
                                    "validation": {"validation_method": "cross_checking_accurate", "cross_checking_threshold": 1},
                                    "filter.this_time_after_validation": {"filter_method": "median", "filter_size": 3}},
}
out = {"shape": [H, W, dmax - dmin + 1]}
for name, pipe in CFGS.items():
    best = None
    for rep in range(3):
        left, right = make_image(L, disparity=[dmin, dmax]), make_image(R, disparity=[-dmax, -dmin])
        machine = PandoraMachine()
        cfg = {"pipeline": json.loads(json.dumps(pipe))}
        cfg["pipeline"] = machine.check_conf(cfg, left, right)["pipeline"]
        runtime.get_engine().sync()
        t0 = time.perf_counter()
        steps = {}
        machine.run_prepare(cfg, left, right)
        for step in list(cfg["pipeline"]):
            t = time.perf_counter()
            machine.run(step, cfg)
            runtime.get_engine().sync()
            steps[step] = round((time.perf_counter() - t) * 1e3, 2)
        machine.run_exit()
        t = time.perf_counter()
        for side in (machine.left_disparity, machine.right_disparity):  # the caller reads its maps: lazy device maps come down now
            if side is not None and len(side.sizes):
                for k in ("disparity_map", "validity_mask", "interpolated_coeff"):
                    if k in side.data_vars:
                        side[k].data
        steps["(reading the result maps)"] = round((time.perf_counter() - t) * 1e3, 2)
        total = (time.perf_counter() - t0) * 1e3
        if best is None or total < best[0]:
            best = (total, steps)
    out[name] = {"total_ms": round(best[0], 2), "steps_ms": best[1]}
print(json.dumps(out, indent=1))
