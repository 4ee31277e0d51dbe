#!/usr/bin/env python3
"""ONE stereo pair over the GPUs of a node with the cost volume sharded over D (BASELINE configs[3]'s layout for the
steps that shard exactly: ZNCC 11x11 + WTA + vfit; the SGM step of that configuration does not shard over D, bench.py --gpus N
runs SGM pipelines over row tiles).  Every exchange is a RCCL collective inside libpandora_amd.so on device buffers:
ncclAllReduce(min, uint64) of the packed keys, ncclAllReduce(sum) of the owner-refined maps.
Launch:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/bench_multi.py
(only the launcher's environment variables are used).  bench.py --gpus N runs the same steps as its `d_sharded_exact` leg.
Rank 0 prints one JSON line: wall time per pair (max over ranks), Mdisp/s of the whole pair, ms spent in the collectives."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pandora_amd import dist as pdist  # noqa: E402
from pandora_amd.comm import Comm, env_world  # noqa: E402
from pandora_amd.engine import Engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=4096)
    ap.add_argument("--width", type=int, default=4096)
    ap.add_argument("--dmin", type=int, default=0)
    ap.add_argument("--dmax", type=int, default=256)
    ap.add_argument("--win", type=int, default=11)
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    rank, world, local_rank, _, _ = env_world()
    eng = Engine(local_rank)
    comm = Comm(eng, always=True)
    H, W, dmin, dmax = args.height, args.width, args.dmin, args.dmax
    L, R = bench.synthetic_pair(H, W, dmin, dmax)
    eng.set_images(L, R, 1)
    (olo, ohi), (wlo, whi) = pdist.disparity_shard(dmin, dmax, 1, world, rank, halo=1)
    cv = eng.alloc_cv(whi - wlo + 1, wlo)

    def step():
        eng.zncc(cv, args.win)
        eng.set_validity(None)
        pdist.sharded_wta(eng, comm, cv, True, wlo - dmin, dmin, 1, -9999.0)
        eng.shard_refine_pack(cv, "vfit", True, olo, ohi, rank == world - 1)
        comm.allreduce_xbuf("refine_pack", "sum")
        comm.allreduce_xbuf("refine_flags", "sum")
        eng.shard_refine_unpack()

    step()
    eng.set_profiling(True)
    eng.reset_stage_times()
    comm.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    eng.sync()
    dt = float(comm.host_allreduce(np.array([time.perf_counter() - t0]), "max")[0]) / args.steps
    coll = eng.stage_time("collective")[0] / args.steps
    if rank == 0:
        cells = H * W * (dmax - dmin + 1)
        print(json.dumps({"mode": "D-sharded ZNCC + WTA + vfit", "n_gpus": world, "shape": [H, W, dmax - dmin + 1], "ms_per_pair": round(dt * 1e3, 3),
                          "Mdisp/s": round(cells / dt / 1e6, 1), "collective_ms_per_pair": round(coll, 3),
                          "collective_bytes_per_pair": H * W * (8 + 16 + 8)}))
    comm.barrier()
    comm.close()
    eng.close()


if __name__ == "__main__":
    main()
