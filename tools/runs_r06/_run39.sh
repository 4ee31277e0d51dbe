cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 2400 python tools/full_size_lazy_vs_eager.py 4096 4096 2>&1 | tail -12 | cut -c1-260
