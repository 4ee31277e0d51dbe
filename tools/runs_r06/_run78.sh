cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python tools/sweep_float_sched.py 2048 4096 33  2048 6000 33  2048 4096 49  1024 8000 33  2048 3600 65 2>&1 | cut -c1-200
