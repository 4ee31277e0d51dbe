#!/usr/bin/env python3
"""cProfile of one machine-level run (host-side Python + waits) at a BASELINE-sized pair: where the wall time of the
reference-shaped API goes beyond the kernels.  Usage: python tools/prof_machine.py [H W dmin dmax]"""
import cProfile
import json
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import pandora_amd  # noqa: E402
from pandora_amd.dataset import make_image  # noqa: E402
from pandora_amd.state_machine import PandoraMachine  # noqa: E402

H, W, dmin, dmax = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (2048, 2048, -128, 0)
L, R = bench.synthetic_pair(H, W, 0, dmax - dmin)
L, R = R, L
PIPE = {"matching_cost": {"matching_cost_method": "census", "window_size": 5},
        "optimization": {"optimization_method": "sgm", "penalty": {"P1": 8, "P2": 32}},
        "disparity": {"disparity_method": "wta", "invalid_disparity": "NaN"},
        "refinement": {"refinement_method": "vfit"}}


def once():
    left, right = make_image(L, disparity=[dmin, dmax]), make_image(R, disparity=[-dmax, -dmin])
    machine = PandoraMachine()
    cfg = {"pipeline": json.loads(json.dumps(PIPE))}
    cfg["pipeline"] = machine.check_conf(cfg, left, right)["pipeline"]
    return pandora_amd.run(machine, left, right, cfg)


once()
once()
pr = cProfile.Profile()
pr.enable()
once()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(30)
pstats.Stats(pr).sort_stats("cumtime").print_stats(45)
