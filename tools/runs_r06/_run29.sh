cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3
bash tools/profile_round.sh r06_c > gpurun_out/profile_round_r06_c.log 2>&1
tail -2 gpurun_out/profile_round_r06_c.log | cut -c1-200
cat gpurun_out/r06_c/tiles.txt | cut -c1-60
