cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_options.py -x -q 2>&1 | tail -15 | cut -c1-220
