cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06d
export BENCH_TRACE=1 HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29561 WORLD_SIZE=8
for it in 1 2; do
start=$(date +%s)
for r in 0 1 2 3 4 5 6 7; do
  RANK=$r LOCAL_RANK=$r timeout 900 python bench.py --gpus 8 --steps 2 --warmup 1 --height 300 --width 256 --dmax 40 --placement-trials 1 --c5-height 640 --c5-width 700 --test-comm tests.transports:TcpComm --test-device 0 > gpurun_out/r06d/it${it}_rank$r.out 2> gpurun_out/r06d/it${it}_rank$r.err &
done
wait
echo "iteration $it: $(( $(date +%s) - start )) s"
grep -h "bench rank 0\]" gpurun_out/r06d/it${it}_rank0.err | tail -12
grep -l "Error" gpurun_out/r06d/it${it}_rank*.err
python -c "
import json; o=json.loads(open('gpurun_out/r06d/it${it}_rank0.out').read().strip().splitlines()[-1]); print(o.get('leg_seconds'), o.get('leg_errors'))"
done
