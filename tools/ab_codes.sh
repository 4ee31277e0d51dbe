# A/B of the integer headline's routes on one box: bash tools/ab_codes.sh   (ms per step + stage times)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run () { # label, env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --no-configs --no-c3 --cpu-rows 0 --steps 6 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']; print('$label', d['ms_per_step'], {k:v for k,v in s.items() if v})"
}
for i in 1 2; do
run nocodes PMX_SGM8_CODES=0
run nocodes_costasync PMX_SGM8_CODES=0 PMX_SGM8_COST_ASYNC=1
run codes PMX_X=1
done
