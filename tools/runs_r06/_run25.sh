cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r06_b > gpurun_out/profile_round_r06_b.log 2>&1
tail -3 gpurun_out/profile_round_r06_b.log | cut -c1-300
cat gpurun_out/r06_b/tiles.txt | cut -c1-60
