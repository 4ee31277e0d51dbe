cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python tools/debug_cbca_inplace.py 4096 4096 256 2>&1 | tail -4 | cut -c1-600
timeout 600 python tools/debug_cbca_inplace.py 1024 4096 256 2>&1 | tail -4 | cut -c1-600
timeout 600 python tools/debug_cbca_inplace.py 4096 1024 256 2>&1 | tail -4 | cut -c1-600
timeout 600 python tools/debug_cbca_inplace.py 2048 2048 128 2>&1 | tail -4 | cut -c1-600
