cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python tools/fuzz_large.py 1200 40 2>&1 | grep "fuzz_large:\|^BAD" | cut -c1-300
FUZZ_FROM=980000 FUZZ_TO=981000 timeout 300 python tools/fuzz_machine.py 2>&1 | tail -1
