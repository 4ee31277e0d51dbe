cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# the fuzz campaign once more on the round's last native sources (the size rule, the CBCA descriptor and the sticky-error change came after the first one)
FUZZ_BASE=900000 bash tools/fuzz_campaign.sh
for f in machine validation mid filters confidence more; do echo "== $f"; tail -3 gpurun_out/camp/$f.log | cut -c1-300; done
echo "## pytest -m gpu"
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3
