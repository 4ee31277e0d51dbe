cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run () { echo "== $*"; env "$@" timeout 900 python -m pytest tests/test_gpu_full_size.py -q -p no:cacheprovider -k "agree_at_full_size and (C5 or X16K)" 2>&1 | grep -E "passed|failed|^FAILED|skipped" | tail -3; }
run PMX_SGM8_FAM=0
run PMX_SGM8_HPAIR=2
run PMX_SGM8_CODES=1
run PMX_COST5=0
run PMX_WTA3=0
run PMX_SGM8_FAM_NW=4
run PMX_SGM_SCHED=seq
run PMX_SGM_HFUSED=0
run PMX_SGM_PENDING=0
run PMX_SGM_FAM_PAR=1
run PMX_SGM_FAM_XCD=1
echo "== census + CBCA at 10000 x 10000 x 129, lazy: the marching kernel against passes H + V (whole maps)"
timeout 900 python - <<'PY'
import sys
sys.path.insert(0, '.')
import numpy as np
from pandora_amd.engine import Engine
from tests.test_gpu_full_size import big_pair, SIZES
L, R = big_pair("C5")
H, W, dmin, dmax = SIZES["C5"]
maps = {}
for name, opts in (("march", {}), ("rows", {"CBCA_MARCH": "0"}), ("rows,1 row", {"CBCA_MARCH": "0", "CBCA_ROWS": "1"})):
    eng = Engine(0)
    eng.set_lazy(True)
    for k, v in opts.items():
        eng.set_option(k, v)
    eng.set_images(L, R, 1)
    cv = eng.alloc_cv(dmax - dmin + 1, dmin)
    eng.census(cv, 5)
    eng.cbca(cv, 2, 30.0, 5)
    eng.set_validity(None)
    eng.wta(cv, False, -9999.0)
    eng.refine(cv, "vfit", False)
    maps[name] = eng.get_disparity(want_itp=True)
    cv.free()
    eng.close()
ref = maps["march"]
for name, m in maps.items():
    print(name, [int((~((a == b) | (np.isnan(a.astype(np.float64)) & np.isnan(b.astype(np.float64))))).sum()) for a, b in zip(ref, m)])
PY
