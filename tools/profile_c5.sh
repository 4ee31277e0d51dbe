TAG=r02_f
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
prof () {
  local name=$1; shift
  timeout 900 rocprofv3 --kernel-trace --stats -d $OUT -o ${name}_kt -- "$@" > $OUT/${name}_kt.log 2>&1
  timeout 900 rocprofv3 --pmc FETCH_SIZE -d $OUT -o ${name}_fetch -- "$@" > $OUT/${name}_fetch.log 2>&1
  timeout 900 rocprofv3 --pmc WRITE_SIZE -d $OUT -o ${name}_write -- "$@" > $OUT/${name}_write.log 2>&1
  { echo "# rocprofv3 --kernel-trace --stats -- $*   (MI355X, $TAG)"; grep -h '^{' $OUT/${name}_kt.log | head -3 | sed 's/^/# /'; python tools/rocpd_summary.py $OUT/${name}_kt*.db; } > $OUT/${name}_kernel_stats.csv
  { echo "# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- $*   (MI355X, $TAG)"
    python tools/rocpd_pmc.py $OUT/${name}_fetch*.db; python tools/rocpd_pmc.py $OUT/${name}_write*.db | tail -n +2; } > $OUT/${name}_pmc_hbm.csv
  find $OUT -name "${name}_*.db" -delete
}
prof c5 python tools/bench_configs.py --stages C5
python tools/bench_configs.py --stages > $OUT/baseline_configs.json 2> $OUT/baseline_configs.err
python bench.py --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
