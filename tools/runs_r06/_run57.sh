cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
REPS=2 bash tools/ab_variants.sh units units4 units8 units32 2>&1 | grep -o "^== .*\|'census_cost': [0-9.]*\|^ *1 *4096 *[0-9.]*"
for v in units4 units units32; do cp pandora_amd/libpandora_amd.so /tmp/base.so; cp pandora_amd/libvar_$v.so pandora_amd/libpandora_amd.so; echo "== $v C3"; timeout 300 python tools/bench_configs.py --stages C3 2>&1 | tail -1 | grep -o "'census_cost[^,]*\|\"census_cost[^,]*\|\"ms\": [0-9.]*"; cp /tmp/base.so pandora_amd/libpandora_amd.so; done
