cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "## pytest -m gpu"
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
echo "## sweep after"
timeout 600 python tools/sweep_float_sched.py 2048 2600 129 1500 3000 257 400 4096 257 2048 2048 129 2>&1 | cut -c1-200
echo "## fuzz_large 700 30 (float routes at mid sizes ride in it)"
timeout 600 python tools/fuzz_large.py 700 30 2>&1 | grep "fuzz_large:\|^BAD" | cut -c1-300
bash tools/profile_round.sh r06_i > gpurun_out/profile_round_r06_i.log 2>&1
tail -2 gpurun_out/profile_round_r06_i.log | cut -c1-200
