# run-to-run spread of the headline pipeline on one box: plain hipMalloc against placement-aware allocation (4 candidates)
for i in 1 2 3 4 5 6; do
  for t in 1 6; do
    python bench.py --cpu-rows 0 --no-north-star --steps 20 --placement-trials $t 2> /dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']; print('trials $t', d['ms_per_step'], s['sgm_fused'], s['wta'], s['census_cost'])"
  done
done
