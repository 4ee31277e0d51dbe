#!/usr/bin/env python3
"""HBM bytes per step of the SGM kernels of the integer path from the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_int.sh (or
tools/profile_round.sh), corrected as MI355X_MICROARCH.md (section HBM) prescribes: bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
(gfx950's FETCH_SIZE reports half of a wide coalesced read; WRITE_SIZE is uncalibrated).
Usage: pmc_traffic.py <pmc_hbm.csv> H W D [<pmc_hbm.csv> H W D ...] -> JSON list that bench.py reads for `roofline.traffic` (it
cannot run rocprofv3 --pmc on itself).  Every entry is stamped with the commit and with a hash of the kernel sources it was taken
on (bench.sgm_source_hash): bench.py drops `traffic` when the built sources differ."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

SGM_KERNELS = ("sgm_u8_packed_kernel", "sgm_fam8_kernel", "sgm_u8_hpair_kernel")

out = []
args = sys.argv[1:]
try:
    commit = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip() or None
except OSError:
    commit = None
for i in range(0, len(args), 4):
    path, H, W, D = args[i], int(args[i + 1]), int(args[i + 2]), int(args[i + 3])
    per = {}
    with open(path) as f:
        for row in csv.reader(line for line in f if not line.startswith("#")):
            if len(row) >= 4 and any(k in row[0] for k in SGM_KERNELS):
                per.setdefault(row[0], {})[row[1]] = float(row[3])
    kernels = {k: v for k, v in per.items() if "FETCH_SIZE" in v and "WRITE_SIZE" in v}
    if kernels:
        each = {k: int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024) for k, v in kernels.items()}
        out.append({"workload": {"H": H, "W": W, "D": D, "kernels": sorted(kernels)},
                    "per_kernel": {k: {"FETCH_SIZE_KiB_per_dispatch": v["FETCH_SIZE"], "WRITE_SIZE_KiB_per_dispatch": v["WRITE_SIZE"],
                                       "hbm_bytes_per_launch": each[k]} for k, v in kernels.items()},
                    "hbm_bytes_per_launch": sum(each.values()),
                    "source": path.split("/")[-1], "commit": commit, "kernel_source_sha16": bench.sgm_source_hash()})
print(json.dumps(out, indent=1))
