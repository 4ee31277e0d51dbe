cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_sgm_family.py -x -q 2>&1 | tail -1
cp pandora_amd/libpandora_amd.so /tmp/base.so; cp pandora_amd/libvar_pk.so pandora_amd/libpandora_amd.so
timeout 600 python -m pytest tests/test_gpu_sgm_family.py -x -q 2>&1 | tail -1
cp /tmp/base.so pandora_amd/libpandora_amd.so
CMD="python tools/bench_configs.py --stages C4 C5" REPS=2 bash tools/ab_variants.sh pk 2>&1 | sed 's/"shape[^}]*"ms"/"ms"/' | cut -c1-330
