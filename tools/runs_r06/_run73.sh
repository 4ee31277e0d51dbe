cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "## pytest -m gpu"
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
echo "## sweep after"
timeout 400 python tools/sweep_float_sched.py 2048 2600 129 3000 3000 129 2048 2600 65 2>&1 | cut -c1-200
echo "## mid-width float32 routes against each other (fuzz_large: lazy with forced routes against eager)"
timeout 600 python tools/fuzz_large.py 900 30 2>&1 | grep "fuzz_large:\|^BAD" | cut -c1-300
bash tools/profile_round.sh r06_k > gpurun_out/profile_round_r06_k.log 2>&1
tail -2 gpurun_out/profile_round_r06_k.log | cut -c1-160
