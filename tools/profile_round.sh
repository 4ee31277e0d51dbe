#!/bin/bash
# Regenerates the judged profile artefacts for one round on the GPU box: bench line, rocprofv3 kernel-trace stats,
# and the separate PMC passes (HBM bytes, instruction mix).  Usage: bash tools/profile_round.sh r01_c
# Outputs land in gpurun_out/<tag>/ (merged back by gpurun); copy the summaries into profiles/.
TAG=${1:-r01_c}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
python bench.py --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
BENCH="python bench.py --steps 5 --warmup 2 --cpu-rows 0 --no-north-star"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT -o kt -- $BENCH > $OUT/kt.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT -o fetch -- $BENCH > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT -o write -- $BENCH > $OUT/write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS -d $OUT -o mix -- $BENCH > $OUT/mix.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS -d $OUT -o act -- $BENCH > $OUT/act.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- $BENCH   (MI355X, $TAG)"; python tools/rocpd_summary.py $OUT/kt*.db; } > $OUT/kernel_stats.csv
{ echo "# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- $BENCH   (MI355X, $TAG)"
  python tools/rocpd_pmc.py $OUT/fetch*.db; python tools/rocpd_pmc.py $OUT/write*.db | tail -n +2; } > $OUT/pmc_hbm.csv
{ echo "# rocprofv3 --pmc <SQ instruction mix> (two passes) -- $BENCH   (MI355X, $TAG)"
  python tools/rocpd_pmc.py $OUT/mix*.db; python tools/rocpd_pmc.py $OUT/act*.db | tail -n +2; } > $OUT/pmc_sq.csv
python tools/bench_kernels.py > $OUT/general_path_kernels.json 2> $OUT/general.err
rm -f $OUT/*.db
cat $OUT/bench.json; head -8 $OUT/kernel_stats.csv; head -12 $OUT/pmc_hbm.csv
