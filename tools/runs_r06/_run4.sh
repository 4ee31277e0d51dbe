cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PMX_SGM_FAM_PAR=0
mkdir -p gpurun_out/pmc4
for g in 1 8 32; do
for c in FETCH_SIZE WRITE_SIZE; do
  PMX_SGM_FAM_XCD=$g timeout 300 rocprofv3 --pmc $c -d gpurun_out/pmc4 -o ${c}_$g -- python tools/bench_configs.py C4 > gpurun_out/pmc4/log_${c}_$g.txt 2>&1
  echo "G=$g $c"; python tools/rocpd_pmc.py gpurun_out/pmc4/${c}_${g}*.db | grep family | cut -c1-200
done; done
rm -f gpurun_out/pmc4/*.db
