import json, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import tests.test_gpu_pipeline as tp
from oracle import capi as orc
sys.path.insert(0, "tools")
import importlib.util
spec = importlib.util.spec_from_file_location("fm", "tools/fuzz_machine.py")
src = open("tools/fuzz_machine.py").read().split("fails = 0")[0]
ns = {}
exec(compile(src, "fm", "exec"), ns)
seed = int(sys.argv[1])
cfg, L, R, dmin, dmax, mskL, mskR, layers = ns["draw"](seed)
print(json.dumps(cfg), L.shape, dmin, dmax, mskL is not None, mskR is not None, "integer" if (L == np.round(L)).all() else "float")
steps = list(cfg["pipeline"])
for n in range(1, len(steps) + 1):
    sub = {"pipeline": {k: cfg["pipeline"][k] for k in steps[:n]}}
    if "disparity" not in sub["pipeline"]:
        sub["pipeline"]["disparity"] = {"disparity_method": "wta"}
    for lazy in (True, False):
        from pandora_amd import runtime
        runtime.get_engine().set_lazy(lazy)
        machine, got = tp.run_machine(L, R, json.loads(json.dumps(sub)), dmin, dmax, mskL, mskR, layers)
        mc_only = {"pipeline": {"matching_cost": cfg["pipeline"]["matching_cost"], "disparity": {"disparity_method": "wta"}}}
        cv0, _, _, _ = tp.oracle_pipeline(orc, L, R, mc_only, dmin, dmax, mskL, mskR)
        val0 = tp.expected_validity(L, R, sub, dmin, dmax, mskL, mskR, np.min(np.isnan(cv0), axis=2))
        ecv, edisp, eval_, eitp = tp.oracle_pipeline(orc, L, R, sub, dmin, dmax, mskL, mskR, val0, layers)
        g = machine.left_cv["cost_volume"].data
        bad = ~((g == ecv) | (np.isnan(g) & np.isnan(ecv)))
        print(steps[:n][-1], "lazy" if lazy else "eager", "cv mismatches", int(bad.sum()), "disp mismatches",
              int((~((got["disparity_map"].data == edisp) | (np.isnan(got["disparity_map"].data) & np.isnan(edisp)))).sum()))
        if bad.any():
            idx = np.argwhere(bad)[:8]
            for r, c, d in idx:
                print("   ", (r, c, d), g[r, c, d], ecv[r, c, d])

# SGM path by path on the aggregated volume of this seed, every schedule
if "optimization" in cfg["pipeline"]:
    from pandora_amd import runtime
    eng = runtime.get_engine()
    eng.set_lazy(False)
    pre = {"pipeline": {k: cfg["pipeline"][k] for k in steps[:steps.index("optimization")]}}
    pre["pipeline"]["disparity"] = {"disparity_method": "wta"}
    machine, _ = tp.run_machine(L, R, json.loads(json.dumps(pre)), dmin, dmax, mskL, mskR)
    vol = np.array(machine.left_cv["cost_volume"].data)
    cmax = float(machine.left_cv.attrs["cmax"])
    opt = cfg["pipeline"]["optimization"]
    P1, P2 = float(opt["penalty"]["P1"]), float(opt["penalty"]["P2"])
    print("cmax", cmax, "invalid", np.float32(cmax + 1.0), "NaN cells", int(np.isnan(vol).sum()), "of", vol.size)
    dcv = machine.left_cv["cost_volume"].device_cv
    for sched in ("seq", "par"):
        eng.set_option("SGM_SCHED", sched)
        for k in range(8):
            dcv.from_host(vol)
            eng.sgm(dcv, P1, P2, False, cmax + 1.0, False, dir_mask=1 << k)
            g = dcv.to_host()
            e = orc.sgm(vol, P1, P2, False, cmax + 1.0, False, dir_mask=1 << k)
            bad = ~((g == e) | (np.isnan(g) & np.isnan(e)))
            print(sched, "dir", k, "mismatches", int(bad.sum()), [(tuple(int(x) for x in i), float(g[tuple(i)]), float(e[tuple(i)])) for i in np.argwhere(bad)[:3]])
    for sched in ("seq", "par"):
        eng.set_option("SGM_SCHED", sched)
        for mask in (0xFF, 0x03, 0x1C, 0xE0, 0xFC, 0x1F):
            dcv.from_host(vol)
            eng.sgm(dcv, P1, P2, False, cmax + 1.0, False, dir_mask=mask)
            g = dcv.to_host()
            e = orc.sgm(vol, P1, P2, False, cmax + 1.0, False, dir_mask=mask)
            bad = ~((g == e) | (np.isnan(g) & np.isnan(e)))
            print(sched, "mask", hex(mask), "mismatches", int(bad.sum()), [(tuple(int(x) for x in i), float(g[tuple(i)]), float(e[tuple(i)])) for i in np.argwhere(bad)[:3]])
