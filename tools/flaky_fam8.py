#!/usr/bin/env python3
"""Repeats small direction-family runs (every route of the integer path's family form) against the CPU oracle, many times: a
hand-off or look-ahead that is only wrong under some timing shows up as a handful of mismatching runs out of a hundred (round 4:
asm loads hidden from the compiler's wait counts in the marching kernel - 1 to 20 of 100 runs at 50 x 45 x 201, nothing at the
other shapes; taken out).  Test infrastructure: uses oracle/.  Usage: python tools/flaky_fam8.py [runs]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import capi  # noqa: E402
from pandora_amd.engine import Engine  # noqa: E402
from tests.test_gpu_parity import pair  # noqa: E402

eng = Engine(0)


def run(H, W, dmin, dmax, win, P1, P2):
    D = dmax - dmin + 1
    L, R = pair(H, W, seed=3 * H + W)
    eng.set_images(L, R, 1)
    cv = eng.alloc_cv(D, dmin)
    eng.census(cv, win)
    eng.sgm(cv, P1, P2, False, float(win * win + 1), False)
    vol = cv.to_host()
    cv.free()
    ref = capi.sgm(capi.census_cost(L, R, D, dmin, 1, win), P1, P2, False, float(win * win + 1), False)
    return int((~((vol == ref) | (np.isnan(vol) & np.isnan(ref)))).sum())


cases = [(50, 45, -100, 100, 5, 8, 32), (70, 16, -5, 5, 5, 1, 2), (45, 67, -20, 20, 5, 8, 30)]
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 100
os.environ["PMX_SGM8_FAM"] = "1"
os.environ["PMX_SGM8_FAM_NW"] = "8"
routes = (("row walk + marching kernel on the cost volume", {"PMX_SGM8_HPAIR": "3", "PMX_SGM8_CODES": "0"}),
          ("two-sided walk + marching kernel from the words", {"PMX_SGM8_HPAIR": "2", "PMX_SGM8_FAMCODES": "1"}),
          ("two-sided walk + marching kernel on the cost volume", {"PMX_SGM8_HPAIR": "2", "PMX_SGM8_CODES": "0"}),
          ("one-sided walk + marching kernel on the cost volume", {"PMX_SGM8_HPAIR": "1", "PMX_SGM8_CODES": "0"}),
          ("one-sided walk from the words + marching kernel on the cost volume", {"PMX_SGM8_HPAIR": "1", "PMX_SGM8_CODES": "1", "PMX_SGM8_FAMCODES": "0"}),
          ("row walk + marching kernel from the words (short images' default until round 6)", {"PMX_SGM8_HPAIR": "3", "PMX_SGM8_CODES": "1"}))
bad_total = 0
for label, env in routes:
    for k in ("PMX_SGM8_HPAIR", "PMX_SGM8_CODES", "PMX_SGM8_FAMCODES"):
        os.environ.pop(k, None)
    os.environ.update(env)
    eng.options_from_env()  # (the library reads its environment once, at pmx_create)
    bad = []
    for it in range(runs):
        for c in cases:
            n = run(*c)
            if n:
                bad.append((it, c[:2], n))
    bad_total += len(bad)
    print(f"{label}: {len(bad)} mismatching runs of {runs * len(cases)}", bad[:6])
sys.exit(1 if bad_total else 0)
