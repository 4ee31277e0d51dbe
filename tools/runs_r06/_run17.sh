cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_pipeline.py tests/test_gpu_fuzz.py -x -q -k "zncc or fuzz" 2>&1 | tail -5
for i in 1 2 3; do
timeout 600 python tools/bench_configs.py --stages C4 2>&1 | tail -1 | sed 's/"shape.*"ms"/"ms"/' | cut -c1-330
done
timeout 300 python tools/ubench/zncc_windows.py 2>&1 | tail -12
