#!/usr/bin/env python3
"""Stage times of the integer fast path (census -> fused SGM -> WTA -> vfit) at a chosen volume size; the lane map
can be forced with PMX_FUSED_MAP=<gl>x<kpl>.  Usage: python tools/bench_fused.py [H W dmin dmax [reps]]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pandora_amd.engine import Engine  # noqa: E402

H, W, dmin, dmax = (int(x) for x in sys.argv[1:5]) if len(sys.argv) >= 5 else (2048, 2048, 0, 128)
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
D = dmax - dmin + 1
L, R = bench.synthetic_pair(H, W, dmin, dmax)
eng = Engine(0)
eng.set_images(L, R, 1)
eng.set_profiling(True)
cv = eng.alloc_cv(D, dmin)


def once():
    eng.census(cv, 5)
    eng.sgm(cv, 8, 32, False, 26.0, False)
    eng.set_validity(None)
    eng.wta(cv, False, -9999.0)
    eng.refine(cv, "vfit", False)


once()
eng.sync()
eng.reset_stage_times()
for _ in range(reps):
    once()
eng.sync()
st = {s: round(eng.stage_time(s)[0] / reps, 4) for s in ("census_transform", "sgm_fused", "wta", "refine")}
raw, gl, kpl = eng.debug_path_costs(cv, raw=True) if H * W * D < 2e8 else (None, None, None)
print(json.dumps({"shape": [H, W, D], "map": os.environ.get("PMX_FUSED_MAP", "auto"), "stage_ms": st,
                  "Gdisp/s": round(H * W * D / sum(st.values()) / 1e6, 1)}))
