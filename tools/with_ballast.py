"""EXPERIMENT: hold N GB of device memory (plain hipMalloc, never touched) while a script runs in this process, so that the script's own
allocations come from further into the device's memory.  tools/ubench/phys_map.hip shows the first ~60 GB a process is given (and the
last ~30) fill / read 7 % slower than the ~150 GB between them.
    python tools/with_ballast.py <GB> <script.py> [args ...]"""
import ctypes
import runpy
import sys

gb = float(sys.argv[1])
script = sys.argv[2]
sys.argv = [script] + sys.argv[3:]
hip = ctypes.CDLL("libamdhip64.so")
held = []
left = int(gb * (1 << 30))
while left > 0:  # blocks of 4 GB
    n = min(left, 4 << 30)
    p = ctypes.c_void_p()
    rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(n))
    if rc != 0:
        print(f"with_ballast: hipMalloc failed ({rc}) with {left >> 30} GB to go", file=sys.stderr)
        break
    held.append(p)
    left -= n
runpy.run_path(script, run_name="__main__")
