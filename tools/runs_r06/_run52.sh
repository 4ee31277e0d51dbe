cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# census_cost_u8_kernel: a workgroup on a run of consecutive quads (one division per wavefront, addresses advanced) against the grid-stride form
REPS=2 bash tools/ab_variants.sh runs16 runs64 runs256 2>&1 | cut -c1-330
echo "## parity with runs64 as the library"
cp pandora_amd/libpandora_amd.so /tmp/base.so; cp pandora_amd/libvar_runs64.so pandora_amd/libpandora_amd.so
timeout 1200 python -m pytest tests/test_gpu_fam8.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_full_size.py -q -x 2>&1 | tail -3
cp /tmp/base.so pandora_amd/libpandora_amd.so
