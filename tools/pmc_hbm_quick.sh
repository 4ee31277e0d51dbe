#!/bin/bash
# HBM bytes (FETCH_SIZE / WRITE_SIZE, separate passes) of every kernel of the bench: bash tools/pmc_hbm_quick.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/hbmq
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c -d gpurun_out/hbmq -o $c -- python bench.py --steps 3 --warmup 1 --cpu-rows 0 > gpurun_out/hbmq/log_$c.txt 2>&1
  python tools/rocpd_pmc.py gpurun_out/hbmq/${c}*.db | rev | cut -d, -f2-4 | rev | paste -d' ' - <(python tools/rocpd_pmc.py gpurun_out/hbmq/${c}*.db | cut -c1-40)
done
rm -f gpurun_out/hbmq/*.db
