cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fam8.py tests/test_gpu_bench_contract.py tests/test_gpu_tiled.py tests/test_gpu_baseline_configs.py -q 2>&1 | tail -2
bash tools/profile_round.sh r06_h > gpurun_out/profile_round_r06_h.log 2>&1
tail -2 gpurun_out/profile_round_r06_h.log | cut -c1-200
cat gpurun_out/r06_h/tiles.txt | cut -c1-60
