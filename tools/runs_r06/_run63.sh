cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "## pytest -m gpu"
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3
echo "## tiles"
timeout 600 python tools/bench_tiles.py 2>&1 | cut -c1-250
echo "## flaky_fam8 30"
timeout 600 python tools/flaky_fam8.py 30 2>&1 | tail -6 | cut -c1-200
echo "## sweep after"
timeout 600 python tools/sweep_hpair.py 592 4096 257 1104 4096 257 1536 4096 257 2088 4096 257 1500 2600 65 2>&1 | cut -c1-400
