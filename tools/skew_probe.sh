# run-to-run spread of the headline pipeline within one box, next to the GPU's clocks / power / temperature
for i in 1 2 3 4 5 6 7 8; do
  python bench.py --cpu-rows 0 --no-north-star --steps 40 2> /dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']; print(d['ms_per_step'], s['sgm_fused'], s['wta'], s['census_cost'])"
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -i "sclk\|mclk\|Power (W)\|Socket Power\|Temperature (Sensor junction)\|memory) (C)" | sed 's/^GPU\[0\]\s*: //' | tr '\n' ';' ; echo
done
