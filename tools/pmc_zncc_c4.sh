#!/bin/bash
# instruction / wait counters of zncc_march_kernel at C4 size (4096 x 4096 x 257, windows 5 and 11)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/zn4 && mkdir -p gpurun_out/zn4
C="env PMX_BENCH_ONLY=zncc python tools/bench_kernels.py 4096 4096 0 256"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS -d gpurun_out/zn4 -o q1 -- $C > gpurun_out/zn4/log1.txt 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT -d gpurun_out/zn4 -o q2 -- $C > gpurun_out/zn4/log2.txt 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC -d gpurun_out/zn4 -o q3 -- $C > gpurun_out/zn4/log3.txt 2>&1
for f in gpurun_out/zn4/q1*.db gpurun_out/zn4/q2*.db gpurun_out/zn4/q3*.db; do python tools/pmc_print.py "$f" zncc_march 2>&1 | head -30; done
rm -f gpurun_out/zn4/*.db
