# A/B of the integer headline's routes on one box (ms per step + stage times): bash tools/ab_codes.sh
#   volume: cost volume + both SGM kernels reading it (the default for tall images)
#   codes:  no cost volume, both SGM kernels make the Hamming costs from the census words (PMX_SGM8_CODES=1)
#   *_inline: the SGM kernels one after the other (PMX_SGM8_OVERLAP=0): every kernel's time is its own
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run () { # label, env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --no-configs --no-c3 --cpu-rows 0 --steps 6 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']; print('$label', d['ms_per_step'], {k:v for k,v in s.items() if v})"
}
for i in 1 2; do
run volume PMX_X=1
run volume_inline PMX_SGM8_OVERLAP=0
run codes PMX_SGM8_CODES=1
run codes_inline PMX_SGM8_CODES=1 PMX_SGM8_OVERLAP=0
done
