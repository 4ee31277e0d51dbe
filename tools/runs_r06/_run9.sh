cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_sgm_family.py -x -q 2>&1 | tail -2
export PMX_SGM_FAM_PAR=0
CMD="python tools/bench_configs.py --stages C4 C5" REPS=2 bash tools/ab_variants.sh allsc1 sfirst 2>&1 | sed 's/"shape.*"ms"/"ms"/' | cut -c1-330
