#!/usr/bin/env python3
"""HBM bytes per launch of the headline kernel from the FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh, corrected as
MI355X_MICROARCH.md (section HBM) prescribes: bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950's FETCH_SIZE reports half of a wide
coalesced read; WRITE_SIZE is uncalibrated).  Usage: pmc_traffic.py <pmc_hbm.csv> H W D [<pmc_hbm.csv> H W D ...] -> JSON list that
bench.py reads for `roofline.traffic` (it cannot run rocprofv3 --pmc on itself)."""
import csv
import json
import sys

out = []
args = sys.argv[1:]
for i in range(0, len(args), 4):
    path, H, W, D = args[i], int(args[i + 1]), int(args[i + 2]), int(args[i + 3])
    vals = {}
    with open(path) as f:
        for row in csv.reader(line for line in f if not line.startswith("#")):
            if len(row) >= 4 and "sgm_u8_packed_kernel" in row[0]:
                vals[row[1]] = (row[0], float(row[3]))
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        out.append({"workload": {"H": H, "W": W, "D": D, "kernel": vals["FETCH_SIZE"][0]},
                    "FETCH_SIZE_KiB_per_dispatch": vals["FETCH_SIZE"][1], "WRITE_SIZE_KiB_per_dispatch": vals["WRITE_SIZE"][1],
                    "hbm_bytes_per_launch": int((2 * vals["FETCH_SIZE"][1] + vals["WRITE_SIZE"][1]) * 1024),
                    "source": path.split("/")[-1]})
print(json.dumps(out, indent=1))
