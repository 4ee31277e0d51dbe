cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cp pandora_amd/libpandora_amd.so /tmp/base.so; cp pandora_amd/libvar_stats.so pandora_amd/libpandora_amd.so
for g in 1 8 32; do echo "G=$g"; PMX_SGM_FAM_XCD=$g PMX_SGM_FAM_PAR=0 timeout 120 python tools/debug_fam_windows.py | tail -2; done
echo C5; PMX_SGM_FAM_XCD=8 PMX_SGM_FAM_PAR=0 timeout 120 python tools/debug_fam_windows.py 10000 10000 129 40 | tail -2
cp /tmp/base.so pandora_amd/libpandora_amd.so
