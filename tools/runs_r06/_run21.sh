cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06h
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r06h -o t8 -- python tools/bench_tiles.py --only headline --ranks 8 > gpurun_out/r06h/t8.log 2>&1
python tools/rocpd_summary.py gpurun_out/r06h/t8*.db | head -20 | cut -c1-200
tail -3 gpurun_out/r06h/t8.log | cut -c1-300
for g in 16 10 8 4; do echo "G=$g"; PMX_SGM8_FAM_XCD=$g timeout 300 python tools/bench_tiles.py --only headline --ranks 4,8 2>&1 | grep "^ *[48] " | cut -c1-220; done
echo "trials=1"; PMX_BENCH_TRIALS=1 timeout 300 python - <<'PY'
import sys, time
sys.path.insert(0, '.')
import numpy as np, bench
from pandora_amd.engine import Engine
from pandora_amd.dist import row_tile
L, R = bench.synthetic_pair(4096, 4096, 0, 256)
for trials in (1, 6):
    eng = Engine(0); eng.set_placement_trials(trials)
    (olo, ohi), (tlo, thi) = row_tile(4096, 8, 3, 40)
    eng.set_images(np.ascontiguousarray(L[tlo:thi]), np.ascontiguousarray(R[tlo:thi]), 1)
    cv = eng.alloc_cv(257, 0)
    def step():
        eng.census(cv, 5); eng.sgm(cv, 8.0, 32.0, False, 26.0, False); eng.set_validity(None); eng.wta(cv, False, -9999.0); eng.refine(cv, "vfit", False)
    step(); eng.sync()
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(8): step()
        eng.sync()
        print("trials", trials, round((time.perf_counter() - t0) / 8 * 1e3, 3), "ms")
    cv.free(); eng.close()
PY
rm -f gpurun_out/r06h/*.db
