import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import bench
from oracle import capi
print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "usable", capi.usable_cores())
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/proc/loadavg"):
    try: print(p, open(p).read().strip())
    except OSError as e: print(p, "absent")
L, R = bench.synthetic_pair(64, 2048, 0, 128)
for th in (1, 4, 8, 16, 32, 64):
    capi.lib(); 
    t0 = time.perf_counter()
    b, _ = bench.cpu_baseline(L, R, 0, 128, 5, 8.0, 32.0, 64, threads=th)
    print(th, b["value"], "Mdisp/s", round(time.perf_counter() - t0, 2), "s", flush=True)
