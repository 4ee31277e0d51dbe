cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "=== hbm probe"; timeout 300 tools/ubench/_bin/hbm_probe 4096
echo "=== halo experiments (PAR=0)"
export PMX_SGM_FAM_PAR=0
CMD="python tools/bench_configs.py --stages C4 C5" REPS=1 bash tools/ab_variants.sh halo1 halo2 halo3
echo "=== PMC C4 float"
mkdir -p gpurun_out/pmc2
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c -d gpurun_out/pmc2 -o $c -- python tools/bench_configs.py C4 > gpurun_out/pmc2/log_$c.txt 2>&1
  python tools/rocpd_pmc.py gpurun_out/pmc2/${c}*.db
done
rm -f gpurun_out/pmc2/*.db
