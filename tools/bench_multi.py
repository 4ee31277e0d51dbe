#!/usr/bin/env python3
"""ONE stereo pair over the GPUs of a node (strong scaling), the two layouts of SURVEY 8e:
  --mode tiled   row tiles with a 40-row margin, any pipeline (BASELINE configs[4]'s layout; default census + SGM)
  --mode dshard  cost volume sharded over D, one all_reduce(MIN) of packed keys (BASELINE configs[3]'s layout without the SGM
                 step: ZNCC 11x11 + WTA + vfit)
Launch:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/bench_multi.py --mode tiled
(one rank per GPU, backend nccl = RCCL; PANDORA_BENCH_BACKEND=gloo PANDORA_AMD_DEVICE=0 runs several ranks on one GPU for tests).
Rank 0 prints one JSON line: wall time per pair (max over ranks) and Mdisp/s of the whole pair."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=["tiled", "dshard"], default="tiled")
    ap.add_argument("--height", type=int, default=4096)
    ap.add_argument("--width", type=int, default=4096)
    ap.add_argument("--dmin", type=int, default=-128)
    ap.add_argument("--dmax", type=int, default=0)
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    import torch
    import torch.distributed as dist

    backend = os.environ.get("PANDORA_BENCH_BACKEND", "nccl")
    local = int(os.environ.get("PANDORA_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(local)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    import bench
    from pandora_amd import dist as pdist
    from pandora_amd.dataset import make_image

    H, W, dmin, dmax = args.height, args.width, args.dmin, args.dmax
    R, L = bench.synthetic_pair(H, W, 0, dmax - dmin)  # (swapped: Pandora's convention wants negative disparities here)
    left, right = make_image(L, disparity=[dmin, dmax]), make_image(R, disparity=[-dmax, -dmin])
    if args.mode == "tiled":
        cfg = {"pipeline": {"matching_cost": {"matching_cost_method": "census", "window_size": 5},
                            "optimization": {"optimization_method": "sgm", "penalty": {"P1": 8, "P2": 32}},
                            "disparity": {"disparity_method": "wta", "invalid_disparity": "NaN"},
                            "refinement": {"refinement_method": "vfit"}}}
        run = lambda: pdist.run_row_tiled(left, right, cfg, margin=40)[0]
    else:
        cfg = {"pipeline": {"matching_cost": {"matching_cost_method": "zncc", "window_size": 11},
                            "disparity": {"disparity_method": "wta", "invalid_disparity": "NaN"},
                            "refinement": {"refinement_method": "vfit"}}}
        run = lambda: pdist.run_d_sharded(left, right, cfg)
    out = run()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = run()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if dist.get_rank() == 0:
        ms = float(t.item()) / args.steps * 1e3
        cells = H * W * (dmax - dmin + 1)
        print(json.dumps({"mode": args.mode, "n_gpus": dist.get_world_size(), "shape": [H, W, dmax - dmin + 1], "ms_per_pair": round(ms, 2),
                          "value": round(cells / ms / 1e3, 1), "unit": "Mdisp/s", "scaling": "strong",
                          "note": "functional flow through the Python plugin API: the time is dominated by host-side numpy and by gathering the 2-D maps on every rank, not by the kernels (bench.py measures those)",
                          "finite_fraction": float(np.isfinite(out["disparity_map"]).mean())}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
