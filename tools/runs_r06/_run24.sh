cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06k
echo "## patience A/B (float32 marching kernel), C4 C5 ms per step"
CMD="python tools/bench_configs.py C4 C5" REPS=2 bash tools/ab_variants.sh pat8 pat32 pat512 2>&1 | sed 's/"shape[^}]*"ms"/"ms"/' | cut -c1-200
echo "## fuzz_fam8 800000 4000"
timeout 900 python tools/fuzz_fam8.py 800000 4000 2>&1 | tail -2
echo "## smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "## pytest -m gpu"
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3
