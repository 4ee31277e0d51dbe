cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 2400 python tools/full_size_rows.py 2>&1 | tail -22 | cut -c1-220
