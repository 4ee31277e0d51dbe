cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# the fuzz campaign on the round's last native sources (the cost kernel's unit map, the code guards)
FUZZ_BASE=940000 bash tools/fuzz_campaign.sh
for f in machine validation mid filters confidence more; do echo "== $f"; tail -2 gpurun_out/camp/$f.log | cut -c1-300; done
echo "## flaky_fam8 100"
timeout 900 python tools/flaky_fam8.py 100 2>&1 | tail -7 | cut -c1-200
echo "## four processes sharing the GPU: fuzz_large"
for k in 0 1 2 3; do a=$((600 + 25 * k)); timeout 1200 python tools/fuzz_large.py $a 25 > gpurun_out/fuzz_large_par2_$k.txt 2>&1 & done; wait
for k in 0 1 2 3; do grep "fuzz_large:\|^BAD\|fault" gpurun_out/fuzz_large_par2_$k.txt | cut -c1-300; done
