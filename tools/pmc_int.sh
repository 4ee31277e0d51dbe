#!/bin/bash
# Kernel trace, HBM counters and SQ counters of the integer headline pipeline (family form), kernels in line (no overlap) so that
# every kernel's numbers are its own.  Usage: bash tools/pmc_int.sh <tag> [extra bench.py args]
TAG=${1:-pmcint}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
export PMX_SGM8_OVERLAP=${PMX_SGM8_OVERLAP:-0}
CMD="python bench.py --steps 3 --warmup 1 --cpu-rows 0 --no-c3 --no-configs --placement-trials 1 $*"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o kt -- $CMD > $OUT/kt.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT -o fetch -- $CMD > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT -o write -- $CMD > $OUT/write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS -d $OUT -o mix -- $CMD > $OUT/mix.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM -d $OUT -o act -- $CMD > $OUT/act.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE TA_TA_BUSY_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT -o ta -- $CMD > $OUT/ta.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- $CMD   (MI355X, $TAG, PMX_SGM8_OVERLAP=$PMX_SGM8_OVERLAP)"; python tools/rocpd_summary.py $OUT/kt*.db; } > $OUT/kernel_stats.csv
{ echo "# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- $CMD   (MI355X, $TAG)"
  python tools/rocpd_pmc.py $OUT/fetch*.db; python tools/rocpd_pmc.py $OUT/write*.db | tail -n +2; } > $OUT/pmc_hbm.csv
{ echo "# rocprofv3 --pmc <SQ counters> (three passes) -- $CMD   (MI355X, $TAG)"
  python tools/rocpd_pmc.py $OUT/mix*.db; python tools/rocpd_pmc.py $OUT/act*.db | tail -n +2; python tools/rocpd_pmc.py $OUT/ta*.db | tail -n +2; } > $OUT/pmc_sq.csv
find $OUT -name "*.db" -delete
head -14 $OUT/kernel_stats.csv; grep -E "fam8|hpair|hrow|sum8_wta|sum3_wta|census_cost" $OUT/pmc_hbm.csv; grep -E "fam8|hpair" $OUT/pmc_sq.csv
