cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python tools/sweep_float_sched.py 1000 3000 65  2048 3500 65  3000 3000 65  2048 2600 33  3000 3400 33  2048 3000 80  1200 2500 100 2>&1 | cut -c1-200
