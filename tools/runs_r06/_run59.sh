cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r06_g > gpurun_out/profile_round_r06_g.log 2>&1
tail -2 gpurun_out/profile_round_r06_g.log | cut -c1-200
cat gpurun_out/r06_g/tiles.txt | cut -c1-60
