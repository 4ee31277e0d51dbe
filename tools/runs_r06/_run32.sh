cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for f in 4 2 0; do echo "== CBCA_FAST=$f"; PMX_CBCA_FAST=$f timeout 600 python -m pytest tests/test_gpu_full_size.py -q -k "bottom_strip and cbca" -p no:cacheprovider 2>&1 | grep -E "passed|failed|Mismatch|Max abs" | tail -3; done
