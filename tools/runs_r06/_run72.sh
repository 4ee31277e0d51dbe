cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python tools/sweep_fam_shape.py 2048 2600 129  3000 3000 129  1500 3000 257  2048 3072 257  1024 3500 129  2048 2600 65  2048 4096 129  2048 4096 65 2>&1 | cut -c1-260
