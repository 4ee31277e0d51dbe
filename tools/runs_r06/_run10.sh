cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_fam8.py -x -q 2>&1 | tail -3
for i in 1 2; do
for g in 1 8 16 32; do
  echo "== G=$g"; PMX_SGM8_FAM_XCD=$g timeout 600 python bench.py --steps 20 --warmup 3 --no-configs --no-c3 --cpu-rows 0 2>&1 | tail -1 | python -c "
import sys, json
o = json.loads(sys.stdin.read()); print(o['ms_per_step'], {k: v for k, v in o['stage_ms_per_step'].items() if v})"
done; done
