cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# the 2-rank and 4-rank tiles of the headline (2088 / 1104 rows) through every horizontal-pair mode, marching kernel from the cost volume / from the words
for hp in default 1 2 3; do for codes in default 0 1; do
  export_hp=""; [ $hp != default ] && export PMX_SGM8_HPAIR=$hp || unset PMX_SGM8_HPAIR
  [ $codes != default ] && export PMX_SGM8_CODES=$codes || unset PMX_SGM8_CODES
  echo "== hpair $hp codes $codes"; timeout 300 python tools/bench_tiles.py --only headline --ranks 2,4 2>&1 | grep "^ *[24] " | cut -c1-260
done; done
