#!/bin/bash
# SQ counters of the float32 horizontal pair at C5.  Usage: bash tools/pmc_h.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmch
mkdir -p $OUT
CMD="python tools/bench_configs.py C5"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS -d $OUT -o mix -- $CMD > $OUT/mix.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM -d $OUT -o act -- $CMD > $OUT/act.log 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE TA_TA_BUSY_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT -o ta -- $CMD > $OUT/ta.log 2>&1
for k in sgm_h_checkpoint sgm_h_backward; do echo "== $k"; for f in mix act ta; do python tools/pmc_print.py "$OUT/$f*.db" $k; done; done > $OUT/summary.txt 2>&1
find $OUT -name "*.db" -delete
cat $OUT/summary.txt
