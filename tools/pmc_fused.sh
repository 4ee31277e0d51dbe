#!/bin/bash
# PMC passes over the bench (separate runs, counters only) -> gpurun_out/pmcf/*.db
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmcf
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY" \
           "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LEVEL_WAVES" \
           "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_ADDR_STALLED_BY_TD_CYCLES TCP_PENDING_STALL_CYCLES" \
           "TCP_TOTAL_CACHE_ACCESSES TCP_TCP_TA_DATA_STALL_CYCLES TD_TD_BUSY TCP_READ_TAGCONFLICT_STALL_CYCLES" \
           "GRBM_GUI_ACTIVE GRBM_TA_BUSY SQ_INSTS_BRANCH SQ_IFETCH SQ_INST_CYCLES_SALU SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d gpurun_out/pmcf -o p$i -- python bench.py --steps 2 --warmup 1 --cpu-rows 0 > gpurun_out/pmcf/log$i.txt 2>&1
done
ls gpurun_out/pmcf
