cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run () { python bench.py --steps 10 --warmup 3 --cpu-rows 0 --no-c3 --no-configs --height $1 --width $2 --dmax $3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], {k:v for k,v in d['stage_ms_per_step'].items() if v})"; }
for shape in "2048 2048 128" "2048 2200 128" "2048 2400 128" "4096 2048 256" "3000 3000 64" "4096 4096 64" "2048 2048 191" "600 2100 256" "3000 2300 95"; do
  for rep in 1 2; do
  echo "== $shape default"; run $shape
  echo "== $shape eight volumes"; PMX_SGM8_FAM=0 run $shape
  echo "== $shape families"; PMX_SGM8_FAM=1 run $shape
  done
done
