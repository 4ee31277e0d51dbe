cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# which horizontal-pair mode the family form wants on short images, re-measured on round 6's kernels
timeout 1500 python tools/sweep_hpair.py 480 4096 257  592 4096 257  800 4096 257  1104 4096 257  1536 4096 257  2088 4096 257  2560 4096 257  3072 4096 257  4096 4096 257 2>&1 | cut -c1-400
timeout 900 python tools/sweep_hpair.py 600 3000 129  1200 3000 129  2000 3000 129  600 2600 65  1500 2600 65  1000 6000 193  2000 6000 193 2>&1 | cut -c1-400
