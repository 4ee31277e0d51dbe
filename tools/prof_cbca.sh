#!/bin/bash
# instruction counters of the CBCA kernels at C3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/cb
export PMX_BENCH_ONLY=cbca
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS -d gpurun_out/cb -o q1 -- python tools/bench_kernels.py > gpurun_out/cb/log1.txt 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/cb -o q2 -- python tools/bench_kernels.py > gpurun_out/cb/log2.txt 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/cb -o q3 -- python tools/bench_kernels.py > gpurun_out/cb/log3.txt 2>&1
for f in gpurun_out/cb/q1*.db gpurun_out/cb/q2*.db gpurun_out/cb/q3*.db; do python tools/rocpd_pmc.py $f | grep "cbca_[hv]" | rev | cut -d, -f2-4 | rev | paste -d' ' - <(python tools/rocpd_pmc.py $f | grep "cbca_[hv]" | cut -c1-20); done
rm -f gpurun_out/cb/*.db
