set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sgm_family.py -x -q 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "sgm" 2>&1 | tail -5
for i in 1 2; do
for par in 0 1; do
  echo "== PAR=$par"; PMX_SGM_FAM_PAR=$par timeout 600 python tools/bench_configs.py --stages C4 C5 2>&1 | tail -4
done; done
