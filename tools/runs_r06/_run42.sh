cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 tools/ubench/_bin/store_stream 40 1 1 | tail -16
