#!/bin/bash
# counters of the CBCA scan kernels at C3 (2048^2 x 129): instruction mix, wait / active cycles, texture-addresser busy, HBM bytes.
# Usage: bash tools/prof_cbca.sh [tag]   (PMX_CBCA_FAST / PMX_CBCA_SIGN select the variant)
TAG=${1:-cb}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
export PMX_BENCH_ONLY=cbca
run () { timeout 300 rocprofv3 --pmc "$@" -d $OUT -o q -- python tools/bench_kernels.py > $OUT/log.txt 2>&1; python tools/rocpd_pmc.py $OUT/q*.db | grep "cbca_[hv]" | sed 's/(cbca_args)//; s/void //' | cut -d, -f1,2,4; rm -f $OUT/q*.db; }
{
run SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS
run SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
run SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_BRANCH SQ_WAIT_INST_LDS
run GRBM_GUI_ACTIVE TA_TA_BUSY TCP_PENDING_STALL_CYCLES
run FETCH_SIZE
run WRITE_SIZE
} > $OUT/counters.csv 2>&1
cat $OUT/counters.csv
