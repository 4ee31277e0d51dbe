# ZNCC kernels at C4 size (4096 x 4096 x 257, windows 5 and 11): base library against variants, alternated on one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
REPS=${REPS:-3} CMD="env PMX_BENCH_ONLY=zncc python tools/bench_kernels.py 4096 4096 0 256" bash tools/ab_variants.sh "$@" 2>&1 | grep -E "^==|\"ms\"" 
