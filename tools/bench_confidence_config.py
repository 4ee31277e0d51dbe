import json, os, sys, time, cProfile, pstats
sys.path.insert(0, "/root/repo")
import bench, pandora_amd
from pandora_amd import runtime
from pandora_amd.dataset import make_image
from pandora_amd.state_machine import PandoraMachine
H, W, dmin, dmax = 2048, 2048, -128, 0
L, R = bench.synthetic_pair(H, W, 0, dmax - dmin)
L, R = R, L
PIPE = {"matching_cost": {"matching_cost_method": "census", "window_size": 5, "subpix": 1},
        "cost_volume_confidence.before": {"confidence_method": "ambiguity", "eta_max": 0.7, "eta_step": 0.01},
        "optimization": {"optimization_method": "sgm", "use_confidence": "cost_volume_confidence.before", "overcounting": False,
                         "penalty": {"penalty_method": "sgm_penalty", "P1": 8, "P2": 32, "p2_method": "constant"}},
        "cost_volume_confidence.after": {"confidence_method": "ambiguity", "eta_max": 0.7, "eta_step": 0.01},
        "disparity": {"disparity_method": "wta", "invalid_disparity": "NaN"},
        "refinement": {"refinement_method": "vfit"},
        "filter": {"filter_method": "median", "filter_size": 3},
        "validation": {"validation_method": "cross_checking_accurate", "cross_checking_threshold": 1}}
def once(profile=False):
    left, right = make_image(L, disparity=[dmin, dmax]), make_image(R, disparity=[-dmax, -dmin])
    machine = PandoraMachine()
    cfg = {"pipeline": json.loads(json.dumps(PIPE))}
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
    for side in (machine.left_disparity, machine.right_disparity):
        for k in ("disparity_map", "validity_mask", "interpolated_coeff"):
            if k in side.data_vars: side[k].data
    return round((time.perf_counter() - t0) * 1e3, 2), steps
once(); print(once())
pr = cProfile.Profile(); pr.enable(); once(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
