cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python tools/leak_probe.py 2>&1 | tail -12 | cut -c1-200
timeout 900 python tools/leak_probe_machine.py 2>&1 | tail -6 | cut -c1-200
