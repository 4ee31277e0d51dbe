cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run () { python bench.py --steps 10 --warmup 3 --cpu-rows 0 --no-c3 --no-configs --height $1 --width $2 --dmax 128 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], {k:v for k,v in d['stage_ms_per_step'].items() if v})"; }
for shape in "2048 2048" "3000 2400"; do
  echo "== $shape default"; run $shape
  echo "== $shape family form forced"; PMX_SGM8_FAM=1 run $shape
done
