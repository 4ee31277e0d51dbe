cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python tools/sweep_float_sched.py 600 800 65  700 900 80  600 800 129  800 1000 90  500 700 257  400 600 129 900 1200 65 2>&1 | cut -c1-200
