cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# the round's last native sources (size rules re-measured): the campaign, the family-form fuzzers, large random pipelines in four processes
FUZZ_BASE=960000 bash tools/fuzz_campaign.sh
for f in machine validation mid filters confidence more; do echo "== $f"; tail -1 gpurun_out/camp/$f.log | cut -c1-200; done
echo "## fuzz_fam8 970000 2000"; timeout 400 python tools/fuzz_fam8.py 970000 2000 2>&1 | tail -1
echo "## flaky_fam8 50"; timeout 600 python tools/flaky_fam8.py 50 2>&1 | tail -6 | cut -c1-160
for k in 0 1 2 3; do a=$((1000 + 30 * k)); timeout 1200 python tools/fuzz_large.py $a 30 > gpurun_out/fuzz_large_par3_$k.txt 2>&1 & done; wait
for k in 0 1 2 3; do grep "fuzz_large:\|^BAD\|fault" gpurun_out/fuzz_large_par3_$k.txt | cut -c1-300; done
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
