cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06l
echo "## chunk sizes forced through the environment"
for g in 1 4; do
  PMX_SGM_FAM_XCD=$g PMX_SGM8_FAM_XCD=$g timeout 1200 python -m pytest tests/test_gpu_sgm_family.py tests/test_gpu_fam8.py tests/test_gpu_full_size.py tests/test_gpu_baseline_configs.py -q 2>&1 | grep -E "passed|failed|error" | tail -2
done
echo "## four processes sharing the GPU"
start=$(date +%s)
for p in 1 2 3 4; do
  timeout 1500 python -m pytest tests/test_gpu_sgm_family.py tests/test_gpu_fam8.py tests/test_gpu_full_size.py -q -p no:cacheprovider > gpurun_out/r06l/proc$p.txt 2>&1 &
done
wait
echo "elapsed $(( $(date +%s) - start )) s"
for p in 1 2 3 4; do grep -E "passed|failed|error" gpurun_out/r06l/proc$p.txt | tail -1; done
grep -l "gave up\|Error" gpurun_out/r06l/proc*.txt
