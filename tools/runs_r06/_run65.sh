cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "## pytest -m gpu -x -q (the driver's command)"
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -2
echo "## bench.py as the driver runs it"
t0=$(date +%s.%N); python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; t1=$(date +%s.%N); echo "wall seconds: $(echo "$t1 - $t0" | bc)"; cut -c1-400 gpurun_out/bench_final.json
