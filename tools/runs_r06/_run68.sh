cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python tools/sweep_float_sched.py 375 450 61  500 500 65  800 1000 129  1000 1000 100  600 800 257  1000 1500 129  1200 1600 65  700 2000 129 2>&1 | cut -c1-200
