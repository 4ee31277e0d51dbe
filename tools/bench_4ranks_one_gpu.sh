cd $GRAFT_REPO_ROOT
for rank in 0 1 2 3; do
  RANK=$rank LOCAL_RANK=$rank WORLD_SIZE=4 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 python bench.py --gpus 4 --test-comm tests.transports:TcpComm --test-device 0 --steps 3 --warmup 1 --height 2048 --width 2048 --dmax 128 --placement-trials 1 > gpurun_out/b4_$rank.log 2>&1 &
done
wait
