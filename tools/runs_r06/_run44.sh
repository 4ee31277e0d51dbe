cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3
bash tools/profile_round.sh r06_f > gpurun_out/profile_round_r06_f.log 2>&1
tail -2 gpurun_out/profile_round_r06_f.log | cut -c1-200
cat gpurun_out/r06_f/tiles.txt | cut -c1-60
