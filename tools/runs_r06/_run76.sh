cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python tools/full_size_lazy_vs_eager.py > gpurun_out/lazy_vs_eager_final.txt 2>&1; tail -12 gpurun_out/lazy_vs_eager_final.txt | cut -c1-260
