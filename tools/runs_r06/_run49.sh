cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 3300 python tools/fuzz_large.py 60 240 > gpurun_out/fuzz_large_60.txt 2>&1
grep -c "^ok" gpurun_out/fuzz_large_60.txt; grep "^BAD\|fuzz_large:\|fault" gpurun_out/fuzz_large_60.txt | cut -c1-700
