#!/usr/bin/env python3
"""Footprint of a long series of PandoraMachine runs (two-sided, filters, cross-checking, maps sometimes read, sometimes dropped
unread, changing shapes): device memory as the HIP runtime sees it, the recycled pinned host blocks and the process's resident
set must level off - the snapshots of unread maps, the winner caches and the staging buffers are all per run or per context.
Usage: python tools/leak_probe_machine.py [runs]"""
import ctypes
import json
import os
import resource
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pandora_amd  # noqa: E402
from pandora_amd import engine, runtime  # noqa: E402
from pandora_amd.dataset import make_image  # noqa: E402
from pandora_amd.state_machine import PandoraMachine  # noqa: E402

_hip = ctypes.CDLL("libamdhip64.so")


def device_mb():
    free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
    _hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total))
    return (total.value - free.value) / 2**20


PIPE = {"matching_cost": {"matching_cost_method": "census", "window_size": 5, "subpix": 1},
        "optimization": {"optimization_method": "sgm", "penalty": {"P1": 8, "P2": 32}},
        "disparity": {"disparity_method": "wta", "invalid_disparity": "NaN"},
        "refinement": {"refinement_method": "vfit"},
        "filter": {"filter_method": "median", "filter_size": 3},
        "validation": {"validation_method": "cross_checking_accurate", "cross_checking_threshold": 1},
        "filter.after": {"filter_method": "median", "filter_size": 3}}
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(0)
shapes = [(600, 800), (512, 512), (700, 900)]
log = []
for it in range(runs):
    H, W = shapes[it % len(shapes)]
    L = rng.integers(0, 255, (H, W)).astype(np.float32)
    R = np.roll(L, 4, 1)
    left, right = make_image(L, disparity=[-40, 0]), make_image(R, disparity=[0, 40])
    machine = PandoraMachine()
    cfg = {"pipeline": json.loads(json.dumps(PIPE))}
    cfg["pipeline"] = machine.check_conf(cfg, left, right)["pipeline"]
    out = pandora_amd.run(machine, left, right, cfg)
    if it % 3 == 0:  # the caller reads its maps ...
        for side in out:
            for k in ("disparity_map", "validity_mask"):
                side[k].data
    # ... or drops them unread
    if it % 25 == 24:
        runtime.get_engine().sync()
        log.append((it + 1, round(device_mb()), round(engine._PinnedBlock._pool_bytes[0] / 2**20),
                    round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024)))
        print("run %4d  device %6d MB  pinned pool %5d MB  max RSS %6d MB" % log[-1], flush=True)
half = len(log) // 2
dev = [x[1] for x in log[half:]]
rss = [x[3] for x in log[half:]]
print("second half: device memory %d..%d MB, max RSS %d..%d MB" % (min(dev), max(dev), min(rss), max(rss)))
print("LEVEL" if max(dev) - min(dev) < 64 and max(rss) - min(rss) < 64 else "GROWING")
