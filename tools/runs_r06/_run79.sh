cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "## pytest -m gpu"
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
echo "## sweep after"
timeout 400 python tools/sweep_float_sched.py 2048 2600 33 2048 4096 49 2048 3600 65 2048 2600 129 2>&1 | cut -c1-200
bash tools/profile_round.sh r06_l > gpurun_out/profile_round_r06_l.log 2>&1
tail -2 gpurun_out/profile_round_r06_l.log | cut -c1-160
