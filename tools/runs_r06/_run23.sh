cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06j
for i in 1 2; do timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/r06j/pytest_$i.txt; tail -2 gpurun_out/r06j/pytest_$i.txt; done
{
echo "# round 6, final sources: tools/flaky_fam8.py 200, tools/fuzz_fam8.py (4000 seeds from 800000), tools/fuzz_campaign.sh FUZZ_BASE=800000"
echo "## flaky_fam8.py 200"; timeout 900 python tools/flaky_fam8.py 200 2>&1 | tail -12
echo "## fuzz_fam8.py"; FUZZ_FROM=800000 FUZZ_TO=804000 timeout 900 python tools/fuzz_fam8.py 2>&1 | tail -4
echo "## fuzz_campaign.sh"; FUZZ_BASE=800000 bash tools/fuzz_campaign.sh; for f in machine validation mid filters confidence more; do echo "### $f"; cat gpurun_out/camp/$f.log; done
} > gpurun_out/r06j/fuzz.txt 2>&1
tail -40 gpurun_out/r06j/fuzz.txt | cut -c1-200
