cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do
echo "== default"; timeout 600 python tools/bench_configs.py --stages C4 C5 2>&1 | tail -2 | sed 's/"shape[^}]*"ms"/"ms"/' | cut -c1-300
echo "== C4 32,9,10"; PMX_SGM_FAM_SHAPE=32,9,10 timeout 600 python tools/bench_configs.py --stages C4 2>&1 | tail -1 | sed 's/"shape[^}]*"ms"/"ms"/' | cut -c1-300
echo "== C5 16,9,8"; PMX_SGM_FAM_SHAPE=16,9,8 timeout 600 python tools/bench_configs.py --stages C5 2>&1 | tail -1 | sed 's/"shape[^}]*"ms"/"ms"/' | cut -c1-300
echo "== C5 32,5,10"; PMX_SGM_FAM_SHAPE=32,5,10 timeout 600 python tools/bench_configs.py --stages C5 2>&1 | tail -1 | sed 's/"shape[^}]*"ms"/"ms"/' | cut -c1-300
done
