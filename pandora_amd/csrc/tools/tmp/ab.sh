cd $GRAFT_REPO_ROOT/pandora_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result"
for v in old new old new; do
  if [ $v = old ]; then X="-DPMX_ZNCC_TAIL_OLD"; else X=""; fi
  /opt/rocm/bin/hipcc $FLAGS $X -c k_matching.hip -o k_matching.o && make > /dev/null 2>&1
  cd $GRAFT_REPO_ROOT; echo "== $v"; python tools/ubench/zncc_align.py 257 257 255 129; cd pandora_amd/csrc
done
