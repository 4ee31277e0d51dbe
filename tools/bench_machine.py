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
MAPS = ("disparity_map", "validity_mask", "interpolated_coeff")


def read_maps(machine):
    """the caller reads its maps: device-resident maps come down now"""
    for side in (machine.left_disparity, machine.right_disparity):
        if side is not None and len(side.sizes):
            for k in MAPS:
                if k in side.data_vars:
                    side[k].data


alive = []  # (freeing the previous pair's 100 MB of arrays is the caller's business, not the run's)
for name, pipe in CFGS.items():
    best = None
    for rep in range(4):
        left, right = make_image(L, disparity=[dmin, dmax]), make_image(R, disparity=[-dmax, -dmin])
        alive.append((left, right))
        machine = PandoraMachine()
        cfg = {"pipeline": json.loads(json.dumps(pipe))}
        cfg["pipeline"] = machine.check_conf(cfg, left, right)["pipeline"]
        runtime.get_engine().sync()
        # (a) the run as a caller sees it: run the steps, read the maps - nothing waits for the GPU in between
        t0 = time.perf_counter()
        machine.run_prepare(cfg, left, right)
        for step in list(cfg["pipeline"]):
            machine.run(step, cfg)
        machine.run_exit()
        t1 = time.perf_counter()
        read_maps(machine)
        total, host = (time.perf_counter() - t0) * 1e3, (t1 - t0) * 1e3
        if best is None or total < best[0]:
            best = [total, host]
    # (b) where the time goes: the same run with a device synchronisation after every step (this serialises host and GPU, the
    #     steps add up to more than (a))
    left, right = make_image(L, disparity=[dmin, dmax]), make_image(R, disparity=[-dmax, -dmin])
    machine = PandoraMachine()
    cfg = {"pipeline": json.loads(json.dumps(pipe))}
    cfg["pipeline"] = machine.check_conf(cfg, left, right)["pipeline"]
    runtime.get_engine().sync()
    steps = {}
    machine.run_prepare(cfg, left, right)
    for step in list(cfg["pipeline"]):
        t = time.perf_counter()
        machine.run(step, cfg)
        runtime.get_engine().sync()
        steps[step] = round((time.perf_counter() - t) * 1e3, 2)
    machine.run_exit()
    t = time.perf_counter()
    read_maps(machine)
    steps["(reading the result maps)"] = round((time.perf_counter() - t) * 1e3, 2)
    out[name] = {"total_ms": round(best[0], 2), "host_ms_before_reading_the_maps": round(best[1], 2), "steps_ms_synchronised": steps}
print(json.dumps(out, indent=1))
