cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06c
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -150 > gpurun_out/r06c/pytest.txt
tail -5 gpurun_out/r06c/pytest.txt
grep -n "Error\|assert\|Mismatch\|differ" gpurun_out/r06c/pytest.txt | head -20
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r06c/bench.json 2> gpurun_out/r06c/bench.err
tail -c 600 gpurun_out/r06c/bench.err
python - <<'PY'
import json
o = json.loads(open('gpurun_out/r06c/bench.json').read().strip().splitlines()[-1])
print(o['ms_per_step'], o['value'], o['stage_ms_per_step'])
for k in ('c3_shape','c4_as_stated','c5_as_stated','plain_hipmalloc'):
    if k in o: print(k, o[k].get('ms_per_step'), o[k].get('stage_ms_per_step'))
print(o.get('disparity_linf_vs_cpu'), o.get('disparity_linf_vs_cpu_kernels'), o.get('leg_errors'))
PY
