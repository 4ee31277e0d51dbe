cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06b
timeout 900 python -m pytest tests/test_gpu_full_size.py -x -q 2>&1 | tail -60 > gpurun_out/r06b/fullsize.txt
cat gpurun_out/r06b/fullsize.txt | cut -c1-300
for i in 1 2; do
for par in 0 1; do
  echo "== PAR=$par"; PMX_SGM_FAM_PAR=$par timeout 600 python tools/bench_configs.py --stages C4 C5 2>&1 | tail -2 | sed 's/"shape.*"ms"/"ms"/' | cut -c1-420
done; done
