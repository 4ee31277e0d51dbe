cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PMX_SGM_FAM_PAR=0
for g in 32 8; do
echo "##### G=$g"
PMX_SGM_FAM_XCD=$g CMD="python tools/bench_configs.py --stages C4 C5" REPS=2 bash tools/ab_variants.sh nolook nolook_nosent nosent 2>&1 | sed 's/"shape.*"ms"/"ms"/' | cut -c1-330
done
