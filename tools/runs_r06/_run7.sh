cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_sgm_family.py -x -q 2>&1 | tail -2
export PMX_SGM_FAM_PAR=0
CMD="python tools/bench_configs.py --stages C4 C5" REPS=2 bash tools/ab_variants.sh publast pubfirst_nolook 2>&1 | sed 's/"shape.*"ms"/"ms"/' | cut -c1-330
echo "#### PAR=1 base"
for i in 1 2; do PMX_SGM_FAM_PAR=1 python tools/bench_configs.py --stages C4 C5 2>&1 | tail -2 | sed 's/"shape.*"ms"/"ms"/' | cut -c1-330; done
