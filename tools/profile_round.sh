#!/bin/bash
# Regenerates the judged profile artefacts of one round on the GPU box.  Usage: bash tools/profile_round.sh r02_a
#   1. the bench line (bench.py as the driver runs it: 4096x4096x257 headline + the 2048x2048x129 leg) - run LAST, after the counter
#      passes have produced <round>_pmc_traffic.json on these very sources (copy it into profiles/ for the line to quote it);
#   2. rocprofv3 --kernel-trace --stats of the integer pipeline at both shapes, FETCH_SIZE / WRITE_SIZE (separate passes) and the SQ
#      instruction mix of the same commands -> *_kernel_stats.csv, *_pmc_hbm.csv, *_pmc_sq.csv, <round>_pmc_traffic.json;
#   3. the float32 SGM schedules at BASELINE configs[3]'s size (ZNCC-less: census costs as float32) and census + CBCA at
#      2048^2 x 129: kernel trace + HBM counters + SQ mix;
#   4. all five BASELINE configurations on one GPU (tools/bench_configs.py).
# Outputs land in gpurun_out/<tag>/ (merged back by gpurun); copy the summaries into profiles/.
TAG=${1:-r02_a}
ROUND=${TAG%%_*}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
prof () {  # prof <name> <command...>: kernel trace + HBM counters + SQ mix of one command
  local name=$1; shift
  timeout 900 rocprofv3 --kernel-trace --stats -d $OUT -o ${name}_kt -- "$@" > $OUT/${name}_kt.log 2>&1
  timeout 900 rocprofv3 --pmc FETCH_SIZE -d $OUT -o ${name}_fetch -- "$@" > $OUT/${name}_fetch.log 2>&1
  timeout 900 rocprofv3 --pmc WRITE_SIZE -d $OUT -o ${name}_write -- "$@" > $OUT/${name}_write.log 2>&1
  timeout 900 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS -d $OUT -o ${name}_mix -- "$@" > $OUT/${name}_mix.log 2>&1
  timeout 900 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS -d $OUT -o ${name}_act -- "$@" > $OUT/${name}_act.log 2>&1
  { echo "# rocprofv3 --kernel-trace --stats -- $*   (MI355X, $TAG)"; grep -h '^{' $OUT/${name}_kt.log | head -3 | sed 's/^/# /'; python tools/rocpd_summary.py $OUT/${name}_kt*.db; } > $OUT/${name}_kernel_stats.csv
  { echo "# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- $*   (MI355X, $TAG)"
    python tools/rocpd_pmc.py $OUT/${name}_fetch*.db; python tools/rocpd_pmc.py $OUT/${name}_write*.db | tail -n +2; } > $OUT/${name}_pmc_hbm.csv
  { echo "# rocprofv3 --pmc <SQ instruction mix> (two passes) -- $*   (MI355X, $TAG)"
    python tools/rocpd_pmc.py $OUT/${name}_mix*.db; python tools/rocpd_pmc.py $OUT/${name}_act*.db | tail -n +2; } > $OUT/${name}_pmc_sq.csv
  find $OUT -name "${name}_*.db" -delete
}
prof northstar python bench.py --steps 5 --warmup 2 --cpu-rows 0 --no-c3 --no-configs
# the shader clock the headline's kernels run at: GRBM_GUI_ACTIVE (cycles the GPU was busy) per dispatch / the dispatch's duration
timeout 900 rocprofv3 --pmc GRBM_GUI_ACTIVE -d $OUT -o northstar_clk -- python bench.py --steps 5 --warmup 2 --cpu-rows 0 --no-c3 --no-configs > $OUT/northstar_clk.log 2>&1
{ echo "# rocprofv3 --pmc GRBM_GUI_ACTIVE -- python bench.py --steps 5 --warmup 2 --cpu-rows 0 --no-c3 --no-configs   (MI355X, $TAG)"; python tools/rocpd_pmc.py $OUT/northstar_clk*.db; } > $OUT/northstar_pmc_clk.csv
find $OUT -name "northstar_clk*.db" -delete
prof c3 python bench.py --steps 5 --warmup 2 --cpu-rows 0 --no-c3 --no-configs --height 2048 --width 2048 --dmax 128
prof float_sgm_c4 python tools/bench_sgm_sched.py C4 --sched fam,seq --reps 2
prof census_cbca_c3 env PMX_BENCH_ONLY=census_cbca python tools/bench_kernels.py
prof c4 python tools/bench_configs.py --stages C4
prof c5 python tools/bench_configs.py --stages C5
prof c2 python tools/bench_configs.py --stages C2
python tools/pmc_traffic.py headline $OUT/northstar_pmc_hbm.csv 4096 4096 257 c3 $OUT/c3_pmc_hbm.csv 2048 2048 129 c4 $OUT/c4_pmc_hbm.csv 4096 4096 257 c5 $OUT/c5_pmc_hbm.csv 10000 10000 129 c2 $OUT/c2_pmc_hbm.csv 375 450 61 > $OUT/pmc_traffic.json
python tools/bench_tiles.py > $OUT/tiles.txt 2> $OUT/tiles.err
bash tools/ab_codes.sh > $OUT/ab_codes.txt 2>&1
cp $OUT/pmc_traffic.json profiles/${ROUND}_pmc_traffic.json
python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
python tools/bench_configs.py --stages > $OUT/baseline_configs.json 2> $OUT/baseline_configs.err
python tools/bench_kernels.py > $OUT/general_path_kernels.json 2> $OUT/general.err
python tools/bench_machine.py > $OUT/machine_level.json 2> $OUT/machine.err
cat $OUT/bench.json; head -12 $OUT/northstar_kernel_stats.csv; cat $OUT/pmc_traffic.json; cat $OUT/baseline_configs.json
