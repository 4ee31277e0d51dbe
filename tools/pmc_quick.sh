#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmcq
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH -d gpurun_out/pmcq -o q1 -- python bench.py --steps 2 --warmup 1 --cpu-rows 0 > gpurun_out/pmcq/log1.txt 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM -d gpurun_out/pmcq -o q2 -- python bench.py --steps 2 --warmup 1 --cpu-rows 0 > gpurun_out/pmcq/log2.txt 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_TA_BUSY TA_TA_BUSY TCP_PENDING_STALL_CYCLES -d gpurun_out/pmcq -o q3 -- python bench.py --steps 2 --warmup 1 --cpu-rows 0 > gpurun_out/pmcq/log3.txt 2>&1
ls gpurun_out/pmcq
