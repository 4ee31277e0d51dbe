cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06n
for it in 1 2; do
start=$(date +%s)
for p in 1 2 3; do
  timeout 1500 python -m pytest tests/test_gpu_sgm_family.py tests/test_gpu_fam8.py tests/test_gpu_full_size.py tests/test_gpu_parity.py -q -p no:cacheprovider -k "not C5 and not X16K" > gpurun_out/r06n/it${it}_proc$p.txt 2>&1 &
done
wait
echo "iteration $it: $(( $(date +%s) - start )) s"
for p in 1 2 3; do grep -E "passed|failed|error" gpurun_out/r06n/it${it}_proc$p.txt | tail -1; done
done
grep -l "gave up\|Mismatch\|memory" gpurun_out/r06n/*.txt
echo done
