cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python tools/debug_cbca_inplace.py 4096 4096 256 2>&1 | tail -3 | cut -c1-300
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3
bash tools/profile_round.sh r06_d > gpurun_out/profile_round_r06_d.log 2>&1
tail -2 gpurun_out/profile_round_r06_d.log | cut -c1-200
cat gpurun_out/r06_d/tiles.txt | cut -c1-60
