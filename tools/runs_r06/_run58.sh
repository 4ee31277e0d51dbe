cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# the cost kernel's unit map + the code guards as long as the range: the suite, the family-form fuzzer, large random pipelines, the mid-size fuzzer
echo "## pytest -m gpu"
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3
echo "## fuzz_fam8 910000 3000"
timeout 600 python tools/fuzz_fam8.py 910000 3000 2>&1 | tail -2
echo "## fuzz_mid"
FUZZ_FROM=920000 FUZZ_TO=921500 timeout 420 python tools/fuzz_mid.py 2>&1 | tail -2
echo "## fuzz_large 500 40"
timeout 900 python tools/fuzz_large.py 500 40 > gpurun_out/fuzz_large_500.txt 2>&1; grep -c "^ok" gpurun_out/fuzz_large_500.txt; grep "^BAD\|fuzz_large:\|fault" gpurun_out/fuzz_large_500.txt | cut -c1-400
echo "## smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
