#!/bin/bash
# kernel trace + instruction counters of the ZNCC kernels at C3 (tools/bench_kernels.py only_zncc mode)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/zn
PMX_BENCH_ONLY=zncc timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/zn -o kt -- python tools/bench_kernels.py > gpurun_out/zn/log0.txt 2>&1
PMX_BENCH_ONLY=zncc timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS -d gpurun_out/zn -o q1 -- python tools/bench_kernels.py > gpurun_out/zn/log1.txt 2>&1
PMX_BENCH_ONLY=zncc timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT -d gpurun_out/zn -o q2 -- python tools/bench_kernels.py > gpurun_out/zn/log2.txt 2>&1
python tools/rocpd_summary.py gpurun_out/zn/kt*.db 2>&1 | head -20
for f in gpurun_out/zn/q1*.db gpurun_out/zn/q2*.db; do python tools/pmc_print.py $f zncc_march 2>&1 | head -30; done
