cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_options.py -q 2>&1 | tail -2
timeout 3000 python tools/fuzz_large.py 0 60 > gpurun_out/fuzz_large_0.txt 2>&1
grep -c "^ok" gpurun_out/fuzz_large_0.txt; grep "^BAD\|fuzz_large:\|fault" gpurun_out/fuzz_large_0.txt | cut -c1-600
