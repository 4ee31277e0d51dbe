#!/usr/bin/env python3
"""Wall-clock trace of one machine-level run: every host function that touches a full-size array, in call order, with the time
it took (device waits included where the function synchronises).  Usage: python tools/trace_machine.py [H W dmin dmax]"""
import functools
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import pandora_amd  # noqa: E402
from pandora_amd import criteria, engine, runtime  # noqa: E402
from pandora_amd.dataset import make_image  # noqa: E402
from pandora_amd.matching_cost import matching_cost as mc  # noqa: E402
from pandora_amd.state_machine import PandoraMachine  # noqa: E402

LOG, DEPTH = [], [0]


def traced(owner, name):
    fn = getattr(owner, name)
    raw = fn.__func__ if isinstance(fn, staticmethod) else fn

    @functools.wraps(raw)
    def wrapper(*a, **k):
        DEPTH[0] += 1
        t = time.perf_counter()
        try:
            return raw(*a, **k)
        finally:
            DEPTH[0] -= 1
            LOG.append((DEPTH[0], f"{getattr(owner, '__name__', owner)}.{name}", (time.perf_counter() - t) * 1e3, t))

    setattr(owner, name, staticmethod(wrapper) if isinstance(owner.__dict__.get(name), staticmethod) else wrapper)


for owner, names in ((runtime, ("ensure_pair", "pair_engine", "_key", "_holders")), (engine, ("pinned_empty",)),
                     (engine._PinnedBlock, ("__init__", "__del__")),
                     (engine.Engine, ("set_images", "set_masks", "set_disparity_grids", "census", "cv_masked", "mark_missing", "sgm",
                                      "compose_validity", "wta", "refine", "new_maps", "fetch_map", "alloc_cv", "sync", "set_disparity",
                                      "set_validity", "maps_restore", "median_filter_maps", "median_filter_disparity",
                                      "cross_checking_maps", "cross_checking", "read_snapshot", "snapshot", "reverse_disp_range",
                                      "free_snapshot", "alloc_snapshot")),
                     (mc, ("grid_extrema",)),
                     (mc.AbstractMatchingCost, ("allocate_cost_volume", "cv_masked", "grid_estimation")),
                     (criteria, ("validity_mask", "mask_invalid_variable_disparity_range", "mask_border")),
                     (PandoraMachine, ("run_prepare", "matching_cost_prepare", "matching_cost_run", "optimization_run", "disparity_run",
                                       "refinement_run", "filter_run", "validation_run"))):
    for n in names:
        traced(owner, n)

H, W, dmin, dmax = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (2048, 2048, -128, 0)
L, R = bench.synthetic_pair(H, W, 0, dmax - dmin)
L, R = R, L
PIPE = {"matching_cost": {"matching_cost_method": "census", "window_size": 5},
        "optimization": {"optimization_method": "sgm", "penalty": {"P1": 8, "P2": 32}},
        "disparity": {"disparity_method": "wta", "invalid_disparity": "NaN"},
        "refinement": {"refinement_method": "vfit"}}
if os.environ.get("TRACE_FULL"):  # the sample configuration with filters and accurate cross-checking
    PIPE = {"matching_cost": {"matching_cost_method": "census", "window_size": 5, "subpix": 1},
            "optimization": {"optimization_method": "sgm", "penalty": {"P1": 8, "P2": 32}},
            "disparity": {"disparity_method": "wta", "invalid_disparity": "NaN"},
            "refinement": {"refinement_method": "vfit"},
            "filter": {"filter_method": "median", "filter_size": 3},
            "validation": {"validation_method": "cross_checking_accurate", "cross_checking_threshold": 1},
            "filter.this_time_after_validation": {"filter_method": "median", "filter_size": 3}}
keep = []
for rep in range(4):
    left, right = make_image(L, disparity=[dmin, dmax]), make_image(R, disparity=[-dmax, -dmin])
    keep.append((left, right))  # (freeing the previous pair's arrays is not part of the run)
    machine = PandoraMachine()
    cfg = {"pipeline": json.loads(json.dumps(PIPE))}
    cfg["pipeline"] = machine.check_conf(cfg, left, right)["pipeline"]
    runtime.get_engine().sync()
    LOG.clear()
    t0 = time.perf_counter()
    out = pandora_amd.run(machine, left, right, cfg)
    t1 = time.perf_counter()
    for side in out:
        for k in ("disparity_map", "validity_mask", "interpolated_coeff"):
            if side is not None and k in side.data_vars:
                side[k].data
    t2 = time.perf_counter()
print(f"run {1e3 * (t1 - t0):.2f} ms, reading the maps {1e3 * (t2 - t1):.2f} ms")
for depth, name, ms, t in sorted(LOG, key=lambda e: e[3]):
    if ms >= 0.02:
        print(f"{1e3 * (t - t0):7.2f}  {'  ' * depth}{name:<60s}{ms:7.2f}")
