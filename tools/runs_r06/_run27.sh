cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run () { python bench.py --steps 10 --warmup 3 --cpu-rows 0 --no-c3 --no-configs --height $1 --width $2 --dmax $3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], {k:v for k,v in d['stage_ms_per_step'].items() if v})"; }
for shape in "2048 2048 128" "2048 2400 128" "3000 2400 128" "1024 4096 256" "4096 2048 256"; do
  for rep in 1 2; do
  echo "== $shape eight volumes"; PMX_SGM8_FAM=0 run $shape
  echo "== $shape families"; PMX_SGM8_FAM=1 run $shape
  echo "== $shape families nw4"; PMX_SGM8_FAM=1 PMX_SGM8_FAM_NW=4 run $shape
  done
done
