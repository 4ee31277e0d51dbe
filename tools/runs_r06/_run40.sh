cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 2400 python tools/full_size_lazy_vs_eager.py 3001 5003 2>&1 | tail -11 | cut -c1-200
timeout 2400 python tools/full_size_lazy_vs_eager.py 6007 2571 2>&1 | tail -11 | cut -c1-200
