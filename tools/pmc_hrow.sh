#!/bin/bash
# SQ counters of the short tiles' kernels (row walk + marching kernel from the words), kernels in line: bash tools/pmc_hrow.sh [ranks]
R=${1:-8}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_hrow
mkdir -p $OUT
export PMX_SGM8_OVERLAP=0
CMD="python tools/bench_tiles.py --only headline --ranks $R --steps 2"
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $OUT/sq_counters.txt
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o kt -- $CMD > $OUT/kt.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS -d $OUT -o mix -- $CMD > $OUT/mix.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC -d $OUT -o act -- $CMD > $OUT/act.log 2>&1
timeout 600 rocprofv3 --pmc SQ_IFETCH SQ_WAIT_IFETCH SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM -d $OUT -o misc -- $CMD > $OUT/misc.log 2>&1
python tools/rocpd_summary.py $OUT/kt*.db | head -8
for n in mix act misc; do python tools/rocpd_pmc.py $OUT/${n}*.db | grep -E "hrow|fam8|counter" ; done
find $OUT -name "*.db" -delete
