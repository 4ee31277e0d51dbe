cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_inner_face.py -x -q 2>&1 | tail -3
bash tools/profile_round.sh r06_a > gpurun_out/profile_round_r06_a.log 2>&1
tail -5 gpurun_out/profile_round_r06_a.log | cut -c1-300
ls gpurun_out/r06_a | head -50
