#!/usr/bin/env python3
"""The GPU box has no .git: a traffic file made there (tools/pmc_traffic.py) carries the hash of the kernel sources it was counted on
but no commit.  Run here, after copying it into profiles/: stamps the current commit into every entry whose source hash is that of the
working tree's kernels (and refuses the others).  Usage: python tools/stamp_traffic.py profiles/r05_pmc_traffic.json"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

path = sys.argv[1]
sha = bench.kernel_source_hash()
commit = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
dirty = subprocess.run(["git", "status", "--porcelain", "pandora_amd/csrc"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
with open(path) as f:
    d = json.load(f)
for w in d if isinstance(d, list) else [d]:
    if w.get("kernel_source_sha16") != sha:
        sys.exit(f"{path}: counted on sources {w.get('kernel_source_sha16')}, the working tree's kernels are {sha}")
    w["commit"] = commit + ("+uncommitted csrc" if dirty else "")
with open(path, "w") as f:
    json.dump(d, f, indent=1)
print(f"{path}: stamped commit {commit}, kernel sources {sha}")
