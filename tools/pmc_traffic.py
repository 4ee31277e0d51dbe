#!/usr/bin/env python3
"""Counted HBM bytes per STEP of the benchmarked pipelines, from the FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh,
corrected as MI355X_MICROARCH.md (section HBM) prescribes and as profiles/r03_d_fetch_calib.txt confirmed on this code's access
patterns: bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950's FETCH_SIZE reports half of the fetched bytes, WRITE_SIZE is exact).

Usage: pmc_traffic.py <label> <pmc_hbm.csv> H W D [<label> <pmc_hbm.csv> H W D ...]
-> JSON list that bench.py reads for `roofline.traffic` / `frac_counted` / `traffic_amplification` (it cannot run rocprofv3 --pmc
on itself).  Per workload: every kernel's bytes per dispatch and dispatches per step (a step = one dispatch of the refinement
kernel, which every pipeline ends with), `sgm_hbm_bytes_per_step` (the SGM kernels: what the roofline block prices) and
`step_hbm_bytes` (every kernel of the step).  Every entry is stamped with the commit and with a hash of the kernel sources it was
taken on (bench.kernel_source_hash): bench.py drops the figures when the built sources differ."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

SGM_KERNELS = ("sgm_u8_packed_kernel", "sgm_fam8_kernel", "sgm_u8_hpair", "sgm_u8_hrow", "sgm_family_kernel", "sgm_h_checkpoint_kernel",
               "sgm_h_backward_kernel", "sgm_path_kernel", "sgm_sum_paths_kernel", "sgm_census_fused_kernel")
STEP_MARKERS = ("sum8_refine_kernel", "near_refine_kernel", "refine_kernel")
NOT_A_STEP = ("placement_probe_kernel", "stream_fill_kernel", "stream_copy_kernel", "stream_read_kernel", "__amd_rocclr")  # allocation-time probes, pmx_measure_hbm's streams, the runtime's fills and copies: listed, not summed


def short(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0]


def main(args):
    try:
        commit = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip() or None
    except OSError:
        commit = None
    out = []
    for i in range(0, len(args), 5):
        label, path, H, W, D = args[i], args[i + 1], int(args[i + 2]), int(args[i + 3]), int(args[i + 4])
        per = {}
        with open(path) as f:
            for row in csv.reader(line for line in f if not line.startswith("#")):
                if len(row) >= 4 and row[1] in ("FETCH_SIZE", "WRITE_SIZE"):
                    k = per.setdefault(short(row[0]), {})
                    k[row[1]] = float(row[3])
                    k["dispatches"] = max(k.get("dispatches", 0), int(float(row[2])))
        kernels = {k: v for k, v in per.items() if "FETCH_SIZE" in v and "WRITE_SIZE" in v}
        steps = max([v["dispatches"] for k, v in kernels.items() if k.startswith(STEP_MARKERS)] or [0])
        if not kernels or not steps:
            continue
        table, sgm, total = {}, 0.0, 0.0
        for k, v in sorted(kernels.items()):
            b = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
            per_step = b * v["dispatches"] / steps
            table[k] = {"FETCH_SIZE_KiB_per_dispatch": v["FETCH_SIZE"], "WRITE_SIZE_KiB_per_dispatch": v["WRITE_SIZE"],
                        "hbm_bytes_per_dispatch": int(b), "dispatches_per_step": round(v["dispatches"] / steps, 3),
                        "hbm_bytes_per_step": int(per_step)}
            if k.startswith(NOT_A_STEP):
                table[k]["not_part_of_a_step"] = True
                continue
            total += per_step
            if k.startswith(SGM_KERNELS):
                sgm += per_step
        out.append({"workload": {"label": label, "H": H, "W": W, "D": D}, "steps_profiled": steps, "per_kernel": table,
                    "sgm_hbm_bytes_per_step": int(sgm), "step_hbm_bytes": int(total), "hbm_bytes_per_launch": int(sgm),
                    "source": os.path.basename(path), "commit": commit, "kernel_source_sha16": bench.kernel_source_hash()})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1:])
