cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# census costs as bytes: sixteen pixels per wavefront and step (overlapping right-word windows: 8 wide loads where four quads take 20)
REPS=3 bash tools/ab_variants.sh x4 runs16 2>&1 | cut -c1-330
echo "## parity with x4 as the library"
cp pandora_amd/libpandora_amd.so /tmp/base.so; cp pandora_amd/libvar_x4.so pandora_amd/libpandora_amd.so
timeout 1200 python -m pytest tests/test_gpu_fam8.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_full_size.py tests/test_gpu_pipeline.py -q -x 2>&1 | tail -3
echo "## c3 shape"
timeout 300 python tools/bench_configs.py --stages C3 2>&1 | tail -3 | cut -c1-400
cp /tmp/base.so pandora_amd/libpandora_amd.so
timeout 300 python tools/bench_configs.py --stages C3 2>&1 | tail -3 | cut -c1-400
