cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "## pytest -m gpu"
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
echo "## sweeps after"
timeout 400 python tools/sweep_fam_rows.py 300 4096 257 300 4096 129 200 4096 129 300 3000 129 2>&1 | cut -c1-200
timeout 400 python tools/sweep_float_sched.py 800 1000 129 600 800 129 2>&1 | cut -c1-200
echo "## short shapes against the oracle through the default route (families below 480 rows)"
timeout 600 python tools/fuzz_large.py 800 20 2>&1 | grep "fuzz_large:\|^BAD" | cut -c1-300
bash tools/profile_round.sh r06_j > gpurun_out/profile_round_r06_j.log 2>&1
tail -2 gpurun_out/profile_round_r06_j.log | cut -c1-160
