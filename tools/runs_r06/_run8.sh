cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_sgm_family.py -x -q 2>&1 | tail -2
export PMX_SGM_FAM_PAR=0
CMD="python tools/bench_configs.py --stages C4 C5" REPS=2 bash tools/ab_variants.sh nolocalrd nolook 2>&1 | sed 's/"shape.*"ms"/"ms"/' | cut -c1-330
mkdir -p gpurun_out/pmc8
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c -d gpurun_out/pmc8 -o ${c} -- python tools/bench_configs.py C4 > gpurun_out/pmc8/log_${c}.txt 2>&1
  echo "$c"; python tools/rocpd_pmc.py gpurun_out/pmc8/${c}*.db | grep "family\|sgm_h" | cut -c1-200
done
rm -f gpurun_out/pmc8/*.db
cp pandora_amd/libpandora_amd.so /tmp/base.so; cp pandora_amd/libvar_stats.so pandora_amd/libpandora_amd.so
timeout 120 python tools/debug_fam_windows.py | tail -2
cp /tmp/base.so pandora_amd/libpandora_amd.so
