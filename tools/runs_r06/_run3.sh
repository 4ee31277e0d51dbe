cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_sgm_family.py -x -q 2>&1 | tail -5
for g in 1 4 16; do echo "G=$g"; PMX_SGM_FAM_XCD=$g timeout 900 python -m pytest tests/test_gpu_sgm_family.py -x -q 2>&1 | tail -2; done
export PMX_SGM_FAM_PAR=0
for i in 1 2; do
for g in 1 8 16 32; do
  echo "== G=$g"; PMX_SGM_FAM_XCD=$g timeout 600 python tools/bench_configs.py --stages C4 C5 2>&1 | tail -2 | cut -c1-420
done; done
echo "== PAR=1 G=8"; PMX_SGM_FAM_PAR=1 timeout 600 python tools/bench_configs.py --stages C4 C5 2>&1 | tail -2 | cut -c1-420
