cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06m
echo "## strip tests alone, 4 times in a row"
for i in 1 2 3 4; do timeout 600 python -m pytest tests/test_gpu_full_size.py -q -k "bottom_strip" -p no:cacheprovider 2>&1 | grep -E "passed|failed" | tail -1; done
echo "## three processes at once, strip tests only, twice"
for it in 1 2; do
for p in 1 2 3; do
  timeout 900 python -m pytest tests/test_gpu_full_size.py -q -k "bottom_strip" -p no:cacheprovider > gpurun_out/r06m/s${it}_$p.txt 2>&1 &
done
wait
for p in 1 2 3; do grep -E "passed|failed" gpurun_out/r06m/s${it}_$p.txt | tail -1; done
done
grep -l "Mismatch\|memory" gpurun_out/r06m/*.txt
