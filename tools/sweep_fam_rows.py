#!/usr/bin/env python3
"""The integer path on short images: the eight path volumes (SGM8_FAM=0) against the direction families (SGM8_FAM=1) and the library's
choice - what the 480-row bound of `pmx_launch_sgm8`'s rule is checked against.  ms per census + SGM + WTA + vfit step, a fresh context
per figure.  Usage: python tools/sweep_fam_rows.py H W D [H W D ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench  # noqa: E402
from sweep_hpair import measure  # noqa: E402

if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    for H, W, D in [tuple(a[i:i + 3]) for i in range(0, len(a), 3)]:
        L, R = bench.synthetic_pair(H, W, 0, D - 1)
        measure(L, R, D, {})  # (a shape's first context runs slow: not counted)
        out = [(measure(L, R, D, o), n) for n, o in (("default", {}), ("eight volumes", {"SGM8_FAM": "0"}), ("families", {"SGM8_FAM": "1"}))]
        print(f"{H} x {W} x {D}: " + "  ".join(f"[{n}] {ms:.2f}" for ms, n in out), flush=True)
