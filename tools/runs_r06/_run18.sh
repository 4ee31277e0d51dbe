REPS=3 bash tools/ab_zncc.sh r5 2>&1 | cut -c1-200
