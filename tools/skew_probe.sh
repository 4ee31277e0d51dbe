for skew in 0 4352 69888 1052928; do
  for i in 1 2 3 4; do
    PMX_DIR_SKEW=$skew python bench.py --cpu-rows 0 --no-north-star --steps 20 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']; print('skew $skew', d['ms_per_step'], s['sgm_fused'], s['wta'])"
  done
done
