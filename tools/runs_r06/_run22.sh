cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06i
for i in 1 2; do
for t in 6 1; do
PMX_TRIALS=$t timeout 300 python - > gpurun_out/r06i/m_${t}_$i.json 2>/dev/null <<'PY'
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
from pandora_amd import runtime
runtime.get_engine().set_placement_trials(int(os.environ["PMX_TRIALS"]))
import runpy
sys.argv = ["bench_machine.py"]
runpy.run_path("tools/bench_machine.py", run_name="__main__")
PY
python -c "
import json; o=json.load(open('gpurun_out/r06i/m_${t}_$i.json')); a=o['census+sgm+wta+vfit']; print('trials $t', a['total_ms'], a['steps_ms_synchronised'], o['a_semi_global_matching.json']['total_ms'])"
done; done
