cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06a
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r06a/pytest.txt
cat gpurun_out/r06a/pytest.txt
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r06a/bench.json 2> gpurun_out/r06a/bench.err
tail -c 600 gpurun_out/r06a/bench.err
python - <<'PY'
import json
o = json.loads(open('gpurun_out/r06a/bench.json').read().strip().splitlines()[-1])
print(o['ms_per_step'], o['value'], o.get('plain_hipmalloc'), o['stage_ms_per_step'])
print(o['roofline'])
for k in ('c3_shape','c4_as_stated','c5_as_stated','plain_hipmalloc'):
    if k in o: print(k, o[k].get('ms_per_step'), o[k].get('stage_ms_per_step'))
print(o.get('disparity_linf_vs_cpu'))
PY
