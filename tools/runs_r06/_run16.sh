cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06f
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -300 > gpurun_out/r06f/pytest.txt
tail -5 gpurun_out/r06f/pytest.txt
grep -n "^FAILED\|^ERROR" gpurun_out/r06f/pytest.txt | head
