cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06g
cp pandora_amd/libpandora_amd.so /tmp/base.so
for v in base zx1 zx256; do
  if [ $v = base ]; then cp /tmp/base.so pandora_amd/libpandora_amd.so; else cp pandora_amd/libvar_$v.so pandora_amd/libpandora_amd.so; fi
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE -d gpurun_out/r06g -o clk_$v -- env PMX_BENCH_ONLY=zncc python tools/bench_kernels.py 4096 4096 0 256 > gpurun_out/r06g/log_$v.txt 2>&1
  echo "== $v"; python tools/rocpd_pmc.py gpurun_out/r06g/clk_${v}*.db | grep -i "zncc_march\|kernel" | cut -c1-250
done
cp /tmp/base.so pandora_amd/libpandora_amd.so
rm -f gpurun_out/r06g/*.db
