cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06e
timeout 900 python -m pytest tests/test_gpu_sgm_family.py tests/test_gpu_fam8.py -x -q 2>&1 | tail -5
for i in 1 2; do
timeout 600 python tools/bench_configs.py --stages C4 C5 2>&1 | tail -2 | sed 's/"shape.*"ms"/"ms"/' | cut -c1-330
timeout 600 python bench.py --steps 20 --warmup 3 --no-configs --no-c3 --cpu-rows 0 2>&1 | tail -1 | python -c "
import sys, json
o = json.loads(sys.stdin.read()); print(o['ms_per_step'], {k: v for k, v in o['stage_ms_per_step'].items() if v})"
done
export BENCH_TRACE=1 HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29561 WORLD_SIZE=8
for it in 1 2; do
start=$(date +%s)
for r in 0 1 2 3 4 5 6 7; do
  RANK=$r LOCAL_RANK=$r timeout 900 python bench.py --gpus 8 --steps 2 --warmup 1 --height 300 --width 256 --dmax 40 --placement-trials 1 --c5-height 640 --c5-width 700 --test-comm tests.transports:TcpComm --test-device 0 > gpurun_out/r06e/it${it}_rank$r.out 2> gpurun_out/r06e/it${it}_rank$r.err &
done
wait
echo "iteration $it: $(( $(date +%s) - start )) s"
grep -h "bench rank 0\] .* <-" gpurun_out/r06e/it${it}_rank0.err | tail -12
grep -l "Error" gpurun_out/r06e/it${it}_rank*.err
python -c "
import json; o=json.loads(open('gpurun_out/r06e/it${it}_rank0.out').read().strip().splitlines()[-1]); print(o.get('leg_seconds'), o.get('leg_errors')); print({k: o[k].get('gathered_maps_vs_one_gpu') for k in ('c4_row_tiled','c5_row_tiled') if k in o})"
done
