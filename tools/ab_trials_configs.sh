# C4 / C5 as stated with one candidate per volume-sized buffer against several, alternated on one box
for i in 1 2 3; do for t in 1 4; do echo "== trials $t"; PMX_BENCH_TRIALS=$t python tools/bench_configs.py --stages C4 C5 2>/dev/null | cut -c1-330; done; done
