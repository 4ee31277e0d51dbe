cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# ablations of the sixteen-pixel cost kernel (results wrong): ns = nothing stored, so = nothing loaded or computed, no = neither
REPS=2 bash tools/ab_variants.sh x4 x4ns x4so x4no 2>&1 | grep -o "^== .*\|'census_cost': [0-9.]*"
