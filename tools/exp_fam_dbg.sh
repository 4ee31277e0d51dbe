for d in 0 1 2 3 4 8 15; do echo "DBG=$d"; PMX_FAM_DBG=$d timeout 200 python tools/bench_sgm_sched.py C4 --sched fam --reps 2 --masks 0x1c | cut -c1-250; done
