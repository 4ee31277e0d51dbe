cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# census costs as bytes: the group's bytes through LDS, stored as whole line-aligned kilobytes
REPS=3 bash tools/ab_variants.sh x4 x4l 2>&1 | cut -c1-330
echo "## parity with x4l as the library"
cp pandora_amd/libpandora_amd.so /tmp/base.so; cp pandora_amd/libvar_x4l.so pandora_amd/libpandora_amd.so
timeout 1200 python -m pytest tests/test_gpu_fam8.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_full_size.py tests/test_gpu_pipeline.py -q -x 2>&1 | tail -3
echo "## c3 shape"
timeout 300 python tools/bench_configs.py --stages C3 2>&1 | tail -3 | cut -c1-400
cp /tmp/base.so pandora_amd/libpandora_amd.so
timeout 300 python tools/bench_configs.py --stages C3 2>&1 | tail -3 | cut -c1-400
