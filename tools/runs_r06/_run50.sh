cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# four fuzz_large processes sharing the one GPU (the window tickets and the allocation fallbacks under contention)
for k in 0 1 2 3; do
  a=$((300 + 40 * k)); b=$((a + 40))
  timeout 1500 python tools/fuzz_large.py $a 40 > gpurun_out/fuzz_large_par_$k.txt 2>&1 &
done
wait
for k in 0 1 2 3; do grep -c "^ok" gpurun_out/fuzz_large_par_$k.txt; grep "^BAD\|fuzz_large:\|fault\|Error\|error" gpurun_out/fuzz_large_par_$k.txt | cut -c1-600 | head -5; done
