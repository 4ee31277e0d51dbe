#!/usr/bin/env python3
"""bench.py - headline benchmark of the stereo hot path on MI355X.

Metric (BASELINE.json): Mdisparities/s = H*W*D / seconds / 1e6 for Census 5x5 -> 8-path SGM (P1=8, P2=32) -> WTA -> vfit on one
synthetic stereo pair, inputs already resident in HBM.

One "step" = one pass of that pipeline over ONE pair, at every N.  Workload = the shape BASELINE.json's north_star quotes its
target on: 4096x4096, d=[0,256] (D=257; configs[3]'s size with the metric's Census+SGM pipeline; it fits one GPU).  N>1 = the same
pair over N ranks (strong scaling): row tiles with the 40-pixel margin the reference's SGM step asks of ROI runs
(optimization/optimization.py:43, marge.py:86-101), every rank runs the pipeline on its rows + margin, and ONE RCCL exchange per step
(a group of ncclSend / ncclRecv inside libpandora_amd.so, on the engine's stream: every rank sends the owned rows of its three result
maps straight to rank 0 over its own xGMI link) leaves the full maps on GPU 0.  One
process per GPU, launched by `python -m torch.distributed.run` (only its environment variables are used: no PyTorch in this file) -
or by bench.py itself: `python bench.py --gpus N` without a launcher's WORLD_SIZE spawns its N ranks (LOCAL_RANK = device, a free
MASTER_PORT), and refuses to run when the box has fewer than N devices or when --gpus disagrees with WORLD_SIZE: a run that was
meant to span N GPUs cannot print a 1-GPU line.  The line carries RCCL's own rank count (ncclCommCount) as `rccl_ranks`.
At N>1 a second leg, `d_sharded_exact`, times north_star's exact multi-GPU form on BASELINE configs[3]'s steps that shard over D
(ZNCC 11x11 + WTA + vfit at 4096x4096x257: disparity slices per rank, ONE ncclAllReduce(min, uint64) of the packed per-pixel keys,
one ncclAllReduce(sum) of the owner-refined maps) and reports the maps' identity with one GPU doing the whole volume; a third,
`pair_per_rank`, is the weak-scaling figure beside the strong-scaling `value`: one whole pair per rank and step, no exchange; a
fourth, `c5_row_tiled`, is BASELINE configs[4] as worded ("10000x10000 ... Census+CBCA+SGM ... row-tiled 8 GPUs"): the strip's fine
scale over the ranks' row tiles, with the whole strip on one GPU timed beside it; a fifth, `c4_row_tiled`, is configs[3]'s pipeline
WITH its SGM step (ZNCC 11x11 + SGM + WTA + vfit, 4096x4096x257) over the same row tiles.  A self-launched run fails fast: the first rank
that exits non-zero stops the others, a watchdog bounds the whole run, every rank's last stderr lines are printed.

Prints ONE JSON line (rank 0) carrying `roofline` (dominant kernel = the 8-path SGM kernel, HIP-event timed on the engine's stream
inside the timed region), `cpu_baseline` (the C oracle, kind "port", 1 thread, on a bounded row strip of the same pair),
`cpu_baseline_all_cores` (same strip, OpenMP), `cpu_baseline_reference_compiled` (the reference's OWN census C++, oracle/_ref,
census stage only) and, at N=1, `c3_shape` (BASELINE configs[2], 2048x2048x129, the round-1 headline, same protocol),
`c2_cones` (configs[1]: census + CBCA + SGM on the reference's cones pair), `c4_as_stated` (configs[3] as BASELINE words it: ZNCC 11x11 + SGM + WTA + vfit, 4096x4096x257, float32 kernels) and `c5_as_stated`
(configs[4]'s fine scale on one GPU: census + CBCA + SGM + WTA + vfit, 10000x10000x129, float32 kernels), each with its own roofline
block (its dominant kernel family), and `plain_hipmalloc` (the headline step with the library's buffer placement switched off,
pmx_set_placement_trials(ctx, 1): what every caller got until round 5.  `value` itself is measured on a context as pmx_create makes
it - since round 6 the library probes six candidates for every new volume-sized buffer by default, so `value` IS what a plugin user
gets).  `disparity_linf_vs_cpu` compares the GPU with the CPU oracle on the strip the CPU baseline ran, through the TIMED step's kernel
instantiation (`disparity_linf_vs_cpu_kernels` names it).  `roofline.peak_measured` is what plain
streaming kernels reach on the box in the same run (pmx_measure_hbm), beside `peak` = the data sheet's 8000 GB/s.
The headline part of the line is complete when the timed region ends; everything after it is a rider: one that raises is dropped
from the line with its error (`leg_errors`), and if the riders are not done `--extras-budget` seconds later (a rank stuck in a
collective of an extra leg) the line is printed as far as it got and every rank leaves with status 0 (HeadlineGuard).
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
SGM_ALGO_BYTES_PER_CELL = 20.0  # two-sweep minimum for 8 paths
PIPELINE_ALGO_BYTES_PER_CELL = 28.0  # census 4 + sgm 20 + wta 4
SGM_MARGIN = 40  # rows of context an SGM tile carries on each side (reference: UniformMargins(40))
STAGES = ("census_transform", "census_cost", "sgm_path", "sgm_family", "sgm_fused", "sgm_span", "wta", "refine", "collective")


def synthetic_pair(H, W, dmin, dmax, seed=20260928):
    """SURVEY 8(d) generator: low-passed integer texture, piecewise-constant ground-truth disparity,
    +-2 integer noise on the right image.  float32, values in [0,255]."""
    rng = np.random.default_rng(seed)
    pad = max(abs(dmin), abs(dmax)) + 8
    T = rng.integers(0, 256, (H, W + 2 * pad)).astype(np.int64)
    T = (T + np.roll(T, 1, 0) + np.roll(T, -1, 0) + np.roll(T, 1, 1) + np.roll(T, -1, 1)
         + np.roll(np.roll(T, 1, 0), 1, 1) + np.roll(np.roll(T, 1, 0), -1, 1)
         + np.roll(np.roll(T, -1, 0), 1, 1) + np.roll(np.roll(T, -1, 0), -1, 1)) // 9
    g = np.random.default_rng(7).integers(dmin, dmax + 1, ((H + 63) // 64, (W + 63) // 64))
    gt = np.kron(g, np.ones((64, 64), np.int64))[:H, :W]
    R = T[:, pad:pad + W]
    cols = np.arange(W)[None, :] + pad + gt
    L = np.take_along_axis(T, cols, axis=1)
    R = np.clip(R + np.random.default_rng(11).integers(-2, 3, (H, W)), 0, 255)
    return L.astype(np.float32), R.astype(np.float32)


def run_pipeline(eng, cv, win, P1, P2):
    eng.census(cv, win)
    eng.sgm(cv, P1, P2, False, float(win * win + 1), False)
    eng.set_validity(None)
    eng.wta(cv, False, -9999.0)
    eng.refine(cv, "vfit", False)


def cpu_baseline(L, R, dmin, dmax, win, P1, P2, rows, threads=1):
    """The C oracle (kind 'port') on a row strip: threads=1 is the reference's serial execution, threads=0 the same loops
    with OpenMP over rows / columns on every host core (the fair CPU ceiling); same results either way."""
    from oracle import capi

    capi.lib()
    cores = capi.set_threads(threads)
    Ls, Rs = np.ascontiguousarray(L[:rows]), np.ascontiguousarray(R[:rows])
    D = dmax - dmin + 1
    t0 = time.perf_counter()
    cv = capi.census_cost(Ls, Rs, D, dmin, 1, win)
    cv = capi.sgm(cv, P1, P2, False, float(win * win + 1), False)
    disp, val = capi.wta(cv, dmin, 1, False, -9999.0)
    capi.refine(cv, disp, val, dmin, dmax, 1, False, "vfit")
    dt = time.perf_counter() - t0
    cells = rows * L.shape[1] * D
    capi.set_threads(1)
    return {"value": round(cells / dt / 1e6, 3), "unit": "Mdisp/s", "cores": cores, "kind": "port",
            "sample": f"first {rows} rows of the {L.shape[0]}x{L.shape[1]} pair, D={D}, census{win}+sgm8+wta+vfit, "
                      f"{dt:.1f} s of oracle/liboracle.so (gcc -O2 -fopenmp), {cores} thread{'s' if cores > 1 else ''}"}, (disp, val)


def cpu_reference_compiled(L, R, dmin, dmax, win, rows):
    """The reference's OWN C++ (matching_cost/cpp/src/census.cpp:97-180, compiled by oracle/Makefile into oracle/_ref) on the same
    strip: the census stage only - it is the only stage of this pipeline the reference holds as compiled code (SGM is an
    un-vendored plugin, WTA is numpy, refinement calls back into Python per pixel).  None when oracle/_ref is not built."""
    from oracle import ref

    mc = ref.load("matching_cost_cpp")
    if mc is None:
        return None
    Ls, Rs = np.ascontiguousarray(L[:rows]), np.ascontiguousarray(R[:rows])
    D = dmax - dmin + 1
    cv = np.full((rows, L.shape[1], D), np.nan, np.float32)
    t0 = time.perf_counter()
    mc.compute_matching_costs(Ls, [Rs], cv, np.arange(D, dtype=np.float32) + dmin, win, win)
    dt = time.perf_counter() - t0
    return {"value": round(rows * L.shape[1] * D / dt / 1e6, 3), "unit": "Mdisp/s", "cores": 1, "kind": "reference-compiled",
            "sample": f"census stage only (matching_cost_cpp.compute_matching_costs, g++ -O2), first {rows} rows of the "
                      f"{L.shape[0]}x{L.shape[1]} pair, D={D}, {dt:.1f} s"}


def kernel_source_hash():
    """sha256 (16 hex digits) of the native sources (csrc/*.hip, *.h, *.cpp): a committed traffic figure (rocprofv3 --pmc cannot
    run inside bench.py) is only quoted while the kernels it was counted on are the ones that were built."""
    import hashlib

    h = hashlib.sha256()
    d = os.path.join(ROOT, "pandora_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".cpp")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def roofline_block(stage, steps, cells):
    """The dominant kernel of the step.  Integer path, family form: sgm_fam8_kernel (the six vertical / diagonal paths as two
    direction families in ONE launch) runs beside sgm_u8_hpair_kernel (the horizontal pair) on the context's two streams; the
    launch duration of the marching kernel IS the span of the SGM step (fork -> join on the context's stream, HIP events), and all
    8 paths are priced at SURVEY 8(d)'s 20 B/cell.  Integer path, eight volumes: all 8 paths in one launch.  Otherwise the float32
    schedule's path launches of one step together."""
    if stage.get("sgm_span", (0, 0))[1] > 0 and stage["sgm_fused"][1] > 0:  # (the float32 family schedule has a span stage too: the pair + the downward family)
        name = ("sgm_fam8_kernel (two direction families = 6 paths, packed u16 marching kernel, one launch) beside sgm_u8_hpair_kernel "
                "(horizontal pair) on a second stream: the SGM step, fork -> join")
        ms, n = stage["sgm_span"]
        avg = ms / max(n, 1)
    elif stage["sgm_fused"][1] > 0:
        name = "sgm_u8_packed_kernel (all 8 SGM paths in one launch, packed u16 arithmetic on 5-bit / byte costs)"
        ms, n = stage["sgm_fused"]
        avg = ms / max(n, 1)
    else:
        name = "float32 SGM schedule (sgm_h_* / sgm_path_kernel + sgm_family_kernel launches of one step)"
        ms = stage["sgm_path"][0] + stage["sgm_family"][0]
        n = stage["sgm_path"][1] + stage["sgm_family"][1]
        avg = ms / max(steps, 1)
    algo = SGM_ALGO_BYTES_PER_CELL * cells
    achieved = algo / (avg * 1e-3) / 1e9 if n else 0.0
    return {"bound": "hbm", "kernel": name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "avg_launch_ms": round(avg, 4), "launches": n,
            "algorithmic_bytes_per_launch": algo}


def own_format(out, w, H, W, D, ms_per_step):
    """The integer route prices badly against SURVEY 8(d)'s float32 bytes (a `frac` above 1 says that the step moves fewer bytes
    than 20 B/cell, not that it beats the memory).  What it moves against its OWN format: the bytes its volumes need if every
    one were written once and read once - the packed costs (Dc bytes per pixel) and the three byte volumes (Dp bytes per pixel
    each: horizontal pair, downward family, upward family) - and how many times that the counters saw over the whole step."""
    kpl = next(k for k in (4, 8, 12, 16, 20) if 16 * k >= D)
    nact = -(-D // kpl)
    dp, dc = nact * kpl, nact * 4 * -(-kpl // 6)  # (five-bit costs, six per dword: census windows up to 5x5)
    own = (2 * dc + 6 * dp) / D
    out["own_format_bytes_per_cell"] = round(own, 3)
    if w is not None:
        cells = H * W * D
        out["counted_bytes_per_cell"] = round(w["step_hbm_bytes"] / cells, 3)
        out["traffic_amplification"] = round(w["step_hbm_bytes"] / (own * cells), 3)
        out["pipeline_hbm_frac_counted"] = round(w["step_hbm_bytes"] / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)


def counted_traffic(label, H, W, D):
    """The committed counter passes of this workload (profiles/*_pmc_traffic.json, tools/pmc_traffic.py: (2 FETCH_SIZE + WRITE_SIZE)
    * 1024 per kernel and step) - only while the native sources are the ones the passes ran on; None otherwise."""
    prof = os.path.join(ROOT, "profiles")
    sha = kernel_source_hash()
    for name in sorted(os.listdir(prof), reverse=True):
        if not name.endswith("_pmc_traffic.json"):
            continue
        try:
            with open(os.path.join(prof, name)) as f:
                pmc = json.load(f)
            for w in pmc if isinstance(pmc, list) else [pmc]:
                wl = w["workload"]
                if (wl.get("label"), wl["H"], wl["W"], wl["D"]) == (label, H, W, D) and w.get("kernel_source_sha16") == sha:
                    return dict(w, file=f"profiles/{name}")
        except (OSError, KeyError, ValueError, TypeError):
            continue
    return None


def add_traffic(roof, label, H, W, D):
    """`traffic` = counted HBM bytes of the SGM kernels per step; `frac_counted` prices the same launch time with those bytes instead
    of the algorithmic ones.  Returns the whole entry (bench.py also quotes the step's total)."""
    w = counted_traffic(label, H, W, D)
    if w is None:
        return None
    roof["traffic"] = w["sgm_hbm_bytes_per_step"]
    roof["traffic_source"] = f"{w['file']} (commit {w.get('commit')})"
    span_ms = roof.get("span_ms_per_step", roof["avg_launch_ms"])
    if span_ms > 0:
        roof["frac_counted"] = round(w["sgm_hbm_bytes_per_step"] / (span_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    return w


def pcie_inclusive_ms(eng, cv, L, R, win, P1, P2, reps=3):
    """Host images in (pmx_set_images: staged through pinned memory, transfer queued), the pipeline, the three 2-D result maps
    out into pinned host arrays (pmx_get_disparity): one step as a caller streaming pairs pays it; best of `reps`."""
    host_out = eng.get_disparity(want_itp=True)  # (also faults the host pages in once, as a streaming caller would)
    best = None
    for _ in range(reps):
        eng.sync()
        t1 = time.perf_counter()
        eng.set_images(L, R, 1)
        run_pipeline(eng, cv, win, P1, P2)
        eng.get_disparity(want_itp=True, out=host_out)
        dt = (time.perf_counter() - t1) * 1e3
        best = dt if best is None or dt < best else best
    return best


def measure_shape(eng, H, W, dmin, dmax, steps, warmup, seed, pcie=True):
    """One pair on one GPU through the whole protocol; returns (ms per step, stage times, (L, R))."""
    L, R = synthetic_pair(H, W, dmin, dmax, seed=seed)
    eng.set_images(L, R, 1)
    cv = eng.alloc_cv(dmax - dmin + 1, dmin)
    for _ in range(warmup):
        run_pipeline(eng, cv, 5, 8.0, 32.0)
    eng.sync()
    eng.set_profiling(True)
    eng.reset_stage_times()
    t0 = time.perf_counter()
    for _ in range(steps):
        run_pipeline(eng, cv, 5, 8.0, 32.0)
    eng.sync()
    ms = (time.perf_counter() - t0) / steps * 1e3
    stage = {name: eng.stage_time(name) for name in STAGES}
    eng.set_profiling(False)
    if pcie:
        stage["(pcie inclusive)"] = (pcie_inclusive_ms(eng, cv, L, R, 5, 8.0, 32.0), 1)
    cv.free()
    return ms, stage, (L, R)


def config_leg(eng, L, R, dmin, dmax, cost, cbca, steps, label, tag):
    """One BASELINE configuration AS STATED on one GPU through the general float32 kernels (the cost volume is float32 between the
    steps): cost = ("zncc", 11) or ("census", 5), optional CBCA (intensity 30, distance 5), SGM 8-path (P1 = 8, P2 = 32), WTA, vfit.
    Same protocol as the headline: inputs resident, one warm-up, `steps` timed steps between stream syncs; per-stage HIP events of
    the same steps.  Its roofline block prices the float32 SGM launches of a step at SURVEY 8(d)'s 20 B/cell."""
    from pandora_amd import _lib

    H, W = L.shape
    D = dmax - dmin + 1
    is_max = cost[0] == "zncc"
    eng.set_images(L, R, 1)
    cv = eng.alloc_cv(D, dmin)

    def step():
        if cost[0] == "census":
            eng.census(cv, cost[1])
        else:
            eng.zncc(cv, cost[1])
        if cbca:
            eng.cbca(cv, cost[1] // 2, 30.0, 5)
        eng.sgm(cv, 8.0, 32.0, is_max, float(cost[1] ** 2 + 1) if cost[0] == "census" else 2.0, False)
        eng.set_validity(None)
        eng.wta(cv, is_max, -9999.0)
        eng.refine(cv, "vfit", is_max)

    step()
    eng.sync()
    eng.set_profiling(True)
    eng.reset_stage_times()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    eng.sync()
    ms = (time.perf_counter() - t0) / steps * 1e3
    stage = {k: eng.stage_time(k) for k in _lib.STAGES}
    eng.set_profiling(False)
    cv.free()
    cells = H * W * D
    roof = roofline_block({k: stage[k] for k in STAGES}, steps, cells)
    per_cell = 36.0 if cbca else 28.0
    w = add_traffic(roof, tag, H, W, D)
    extra = {}
    if w is not None:
        extra = {"counted_bytes_per_cell": round(w["step_hbm_bytes"] / cells, 3),
                 "pipeline_hbm_frac_counted": round(w["step_hbm_bytes"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    return {**extra, "workload": label, "steps": steps, "ms_per_step": round(ms, 3), "value": round(cells / ms / 1e3, 1), "unit": "Mdisp/s",
            "dtype": "f32", "roofline": roof,
            "stage_ms_per_step": {k: round(v[0] / steps, 4) for k, v in stage.items() if v[1]},
            "pipeline_hbm_frac": round(per_cell * cells / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "pipeline_algorithmic_bytes_per_cell": per_cell}


def d_sharded_leg(eng, comm, L, R, dmin, dmax, win, steps, check_device):
    """north_star's exact multi-GPU form on the steps of BASELINE configs[3] that shard over D: every rank builds the
    ZNCC costs of its disparity slice (+1 disparity of halo for the refinement), ONE ncclAllReduce(min, uint64) of a packed
    (orderable cost, global index) key per pixel is the winner-takes-all over the whole range (np.argmax's first extremum), the
    rank owning a pixel's winner refines it and one ncclAllReduce(sum) of value-or-zero maps merges.  Every rank takes part;
    rank 0 returns the block, with the maps' identity against ONE GPU running the whole volume (outside the timed region)."""
    from pandora_amd import dist as pdist
    from pandora_amd.engine import Engine

    rank, world = comm.rank, comm.world
    H, W = L.shape
    eng.set_images(L, R, 1)
    (olo, ohi), (wlo, whi) = pdist.disparity_shard(dmin, dmax, 1, world, rank, halo=1)
    cv = eng.alloc_cv(whi - wlo + 1, wlo)

    def step():
        eng.zncc(cv, win)
        eng.set_validity(None)
        pdist.sharded_wta(eng, comm, cv, True, wlo - dmin, dmin, 1, -9999.0)
        eng.shard_refine_pack(cv, "vfit", True, olo, ohi, rank == world - 1)
        comm.allreduce_xbuf("refine_pack", "sum")
        comm.allreduce_xbuf("refine_flags", "sum")
        eng.shard_refine_unpack()

    step()
    eng.sync()
    comm.barrier()
    eng.set_profiling(True)
    eng.reset_stage_times()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    eng.sync()
    dt = float(comm.host_allreduce(np.array([time.perf_counter() - t0]), "max")[0]) / steps
    coll_ms = eng.stage_time("collective")[0] / steps
    eng.set_profiling(False)
    comm.barrier()
    out = None
    if rank == 0:
        gd, gv, gi = eng.get_disparity(want_itp=True)
        one = Engine(check_device)
        one.set_images(L, R, 1)
        cv1 = one.alloc_cv(dmax - dmin + 1, dmin)
        one.zncc(cv1, win)
        one.set_validity(None)
        one.wta(cv1, True, -9999.0)
        one.refine(cv1, "vfit", True)
        od, ov, oi = one.get_disparity(want_itp=True)
        cv1.free()
        one.close()
        same = [float(np.mean((a == b) | (np.isnan(a) & np.isnan(b)))) for a, b in ((gd, od), (gi, oi))] + [float(np.mean(gv == ov))]
        cells = H * W * (dmax - dmin + 1)
        out = {"workload": f"{H}x{W} pair, d=[{dmin},{dmax}], ZNCC {win}x{win} + WTA + vfit (BASELINE configs[3] without its SGM step, which "
                           f"does not shard over D); costs sharded over D across {world} ranks",
               "collectives": "ncclAllReduce(min, uint64) of 8 B/pixel packed keys + ncclAllReduce(sum) of the owner-refined maps (20 B/pixel)",
               "ms_per_step": round(dt * 1e3, 3), "value": round(cells / dt / 1e6, 1), "unit": "Mdisp/s", "steps": steps,
               "collective_ms_per_step": round(coll_ms, 4),
               "maps_identical_to_one_gpu": round(min(same), 6)}
    cv.free()
    comm.barrier()
    return out


def row_tiled_leg(eng, comm, which, H5, W5, steps, check_device, c4_dmax=256):
    """N > 1: a BASELINE configuration AS STATED over all ranks by row tiles, the reference's ROI convention (marge.py:86-101;
    optimization/optimization.py:43: 40 rows for SGM, plus the cost window's radius):
      which = "c5": configs[4] as BASELINE words it - "10000x10000 ... Census+CBCA+SGM ... row-tiled 8 GPUs" - the fine scale of the
                    strip, d = [-64, 64], census 5x5 + CBCA + SGM 8-path + WTA + vfit;
      which = "c4": configs[3]'s pipeline - 4096x4096, d = [0, 256], ZNCC 11x11 + SGM 8-path + WTA + vfit - WITH its SGM step (the
                    exact D-sharded leg beside it has to leave SGM out: SURVEY 8e).
    float32 volumes between the steps.  Every rank runs the pipeline on its H / N rows + margin, places its owned rows in the
    full-size maps and ONE group of ncclSend / ncclRecv brings them to rank 0 (on the communication stream, under the next step's
    kernels).  Same barrier / max-over-ranks clock as the headline; after the timed region rank 0 runs the whole pair on its one GPU:
    `one_gpu_ms_per_step` and the identity of the gathered maps."""
    from pandora_amd import _lib
    from pandora_amd.dist import row_tile
    from pandora_amd.engine import Engine

    rank, world = comm.rank, comm.world
    if which == "c5":
        dmin, dmax, win, cost, cbca, is_max, inv = -64, 64, 5, "census", True, False, 26.0
    else:
        dmin, dmax, win, cost, cbca, is_max, inv = 0, c4_dmax, 11, "zncc", False, True, 2.0
    D = dmax - dmin + 1
    margin = SGM_MARGIN + win // 2
    L5, R5 = synthetic_pair(H5, W5, dmin, dmax)
    (own_lo, own_hi), (tile_lo, tile_hi) = row_tile(H5, world, rank, margin)
    eng.set_placement_trials(1)
    eng.set_images(np.ascontiguousarray(L5[tile_lo:tile_hi]), np.ascontiguousarray(R5[tile_lo:tile_hi]), 1)
    cv = eng.alloc_cv(D, dmin)

    def pipeline(e, c):
        if cost == "census":
            e.census(c, win)
        else:
            e.zncc(c, win)
        if cbca:
            e.cbca(c, win // 2, 30.0, 5)
        e.sgm(c, 8.0, 32.0, is_max, inv, False)
        e.set_validity(None)
        e.wta(c, is_max, -9999.0)
        e.refine(c, "vfit", is_max)

    def step():
        pipeline(eng, cv)
        eng.tile_place(H5, own_lo, own_hi, tile_lo, True)
        comm.gather_rows(H5, True, root=0)

    step()
    eng.sync()
    comm.barrier()
    eng.set_profiling(True)
    eng.reset_stage_times()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    eng.sync()
    dt = float(comm.host_allreduce(np.array([time.perf_counter() - t0]), "max")[0]) / steps
    stage = {k: eng.stage_time(k) for k in _lib.STAGES}
    eng.set_profiling(False)
    comm.barrier()
    out = None
    gathered = eng.get_full_maps(H5, want_itp=True) if rank == 0 else None
    cv.free()
    if rank == 0:
        one = Engine(check_device)
        one.set_images(L5, R5, 1)
        cv1 = one.alloc_cv(D, dmin)
        pipeline(one, cv1)
        one.sync()
        t1 = time.perf_counter()
        pipeline(one, cv1)
        one.sync()
        one_ms = (time.perf_counter() - t1) * 1e3
        od, ov, oi = one.get_disparity(want_itp=True)
        cv1.free()
        one.close()
        gd, gv, gi = gathered
        cells = H5 * W5 * D
        what = ("BASELINE configs[4], fine scale, row-tiled" if which == "c5" else "BASELINE configs[3]'s pipeline WITH its SGM step, row-tiled")
        steps_txt = ("Census 5x5 + CBCA + SGM 8-path + WTA + vfit" if which == "c5" else "ZNCC 11x11 + SGM 8-path + WTA + vfit")
        out = {"workload": f"{what}: {H5}x{W5} synthetic pair, d=[{dmin},{dmax}] (D={D}), {steps_txt}, float32 volumes between the steps; "
                           f"ONE pair per step over {world} ranks",
               "parallelism": f"row tiles of {H5 // world} rows + {margin}-row margin (40 SGM + {win // 2} window: the reference's ROI convention), "
                              f"one ncclSend / ncclRecv gather of the owned rows of 3 maps to GPU 0 per step; strong scaling",
               "steps": steps, "ms_per_step": round(dt * 1e3, 3), "value": round(cells / dt / 1e6, 1), "unit": "Mdisp/s", "dtype": "f32",
               "one_gpu_ms_per_step": round(one_ms, 3), "speedup_vs_one_gpu": round(one_ms / (dt * 1e3), 3),
               "stage_ms_per_step_rank0": {k: round(v[0] / steps, 4) for k, v in stage.items() if v[1]},
               "gathered_maps_vs_one_gpu": {
                   "disparity_identical": round(float(np.mean((gd == od) | (np.isnan(gd) & np.isnan(od)))), 6),
                   "validity_identical": round(float(np.mean(gv == ov)), 6),
                   "coefficient_identical": round(float(np.mean((gi == oi) | (np.isnan(gi) & np.isnan(oi)))), 6)}}
        if which == "c4":
            # ZNCC costs are not integers: the rows a tile does not see change every sum in its last bits, so the REFINED disparity is
            # rarely the same float; what tiling preserves is the winner and the disparity to a hundredth of a pixel
            with np.errstate(invalid="ignore"):
                both_nan = np.isnan(gd) & np.isnan(od)
                out["gathered_maps_vs_one_gpu"]["winner_identical"] = round(float(np.mean((np.rint(gd) == np.rint(od)) | both_nan)), 6)
                out["gathered_maps_vs_one_gpu"]["disparity_within_0.01"] = round(float(np.mean((np.abs(gd - od) <= 0.01) | both_nan)), 6)
    comm.barrier()
    return out


def pair_per_rank_leg(comm, device, H, W, dmin, dmax, win, P1, P2, steps, rank, world):
    """N > 1, weak scaling beside the strong-scaling headline: every rank runs the WHOLE pipeline on a pair of its own (the way the
    reference is deployed on many tiles or many pairs: no exchange at all), same barrier / max-over-ranks protocol; the aggregate is
    world x cells / time.  Outside the headline's timed region."""
    from pandora_amd.engine import Engine

    one = Engine(device)
    L, R = synthetic_pair(H, W, dmin, dmax, seed=20260928 + rank)
    one.set_images(L, R, 1)
    cv = one.alloc_cv(dmax - dmin + 1, dmin)
    run_pipeline(one, cv, win, P1, P2)
    one.sync()
    comm.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        run_pipeline(one, cv, win, P1, P2)
    one.sync()
    elapsed = float(comm.host_allreduce(np.array([time.perf_counter() - t0]), "max")[0])
    comm.barrier()
    cv.free()
    one.close()
    cells = H * W * (dmax - dmin + 1)
    return {"workload": f"one {H}x{W}x{dmax - dmin + 1} pair PER RANK and step (Census5x5 + SGM + WTA + vfit), no exchange",
            "scaling": "weak", "steps": steps, "ms_per_step": round(elapsed / steps * 1e3, 3),
            "value": round(world * cells / (elapsed / steps) / 1e6, 1), "unit": "Mdisp/s"}


def _free_port_pair():
    """a MASTER_PORT whose successor is free too: the library's rendezvous listens one above the launcher's port (comm.py)"""
    import socket

    for _ in range(64):
        with socket.socket() as a:
            a.bind(("127.0.0.1", 0))
            port = a.getsockname()[1]
            with socket.socket() as b:
                try:
                    b.bind(("127.0.0.1", port + 1))
                except OSError:
                    continue
        return port
    raise OSError("bench.py: no pair of free rendezvous ports at 127.0.0.1")


def spawn_ranks(n, watchdog_s):
    """`python bench.py --gpus N` without a launcher: N ranks of this very command, one per device (LOCAL_RANK = device), a free
    rendezvous port pair at 127.0.0.1; rank 0's stdout (the JSON line) is passed through.  FAILS FAST: all children are polled, the
    first non-zero exit (a rank that died in ncclCommInitRank, say) terminates the others - which would otherwise sit in a
    collective until the driver's timeout - and the launcher exits non-zero within seconds, after printing every rank's last
    stderr lines; so does an overall watchdog.  RCCL's own warnings are switched on for the children (NCCL_DEBUG=WARN)."""
    import subprocess
    import tempfile

    port = _free_port_pair()
    logs, procs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                   NCCL_DEBUG=os.environ.get("NCCL_DEBUG", "WARN"))
        logs.append(tempfile.TemporaryFile(mode="w+", prefix=f"bench_rank{r}_"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL, stderr=logs[r]))

    def tails(lines=25):
        for r, f in enumerate(logs):
            f.flush()
            f.seek(0)
            text = f.read().splitlines()[-lines:]
            if text:
                sys.stderr.write(f"---- rank {r}: last {len(text)} stderr lines ----\n" + "\n".join(text) + "\n")
        sys.stderr.flush()

    def stop_all():
        for p in procs:
            if p.poll() is None:
                p.terminate()
        deadline = time.time() + 5.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()

    t0 = time.time()
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            # a rank that dies takes its peers' connections with it: they may fail within the same tenth of a second - every one
            # that has is named (the exit code is the first named rank's)
            time.sleep(0.3)
            bad = [(r, p.poll()) for r, p in enumerate(procs) if p.poll() not in (None, 0)]
            sys.stderr.write("bench.py: " + "; ".join(f"rank {r} exited with code {c}" for r, c in bad) + "; stopping the other ranks\n")
            stop_all()
            tails()
            rc = bad[0][1] if bad[0][1] > 0 else 1
            break
        if all(c == 0 for c in codes):
            f = logs[0]  # a clean run: rank 0's stderr (RCCL's banner, warnings) is passed on
            f.flush()
            f.seek(0)
            sys.stderr.write(f.read())
            break
        if time.time() - t0 > watchdog_s:
            sys.stderr.write(f"bench.py: the ranks did not finish within {watchdog_s:.0f} s (watchdog); stopping them\n")
            stop_all()
            tails()
            rc = 124
            break
        time.sleep(0.1)
    for f in logs:
        f.close()
    return rc


class HeadlineGuard:
    """The line's headline must survive its riders: once the timed region is over, every rank arms this timer; if the extra legs
    (the BASELINE configurations at N = 1, the row-tiled / D-sharded / pair-per-rank legs at N > 1) have not finished within
    `budget_s`, rank 0 prints the line as far as it got - with the reason under `leg_errors` - and every rank leaves with status 0
    (a rank stuck inside a collective cannot be unstuck from Python; the launcher's watchdog would otherwise take the line with it)."""

    def __init__(self, rank, budget_s):
        self.rank, self.budget_s = rank, budget_s
        self.out, self.leg_errors = None, None
        self.lock = threading.Lock()
        self.done = False
        self.timer = threading.Timer(budget_s, self._bail)
        self.timer.daemon = True

    def arm(self, out, leg_errors):
        self.out, self.leg_errors = out, leg_errors
        if self.budget_s > 0:
            self.timer.start()

    def _bail(self):
        with self.lock:
            if self.done:
                return
            self.done = True
            if self.rank == 0 and self.out is not None:
                line = dict(self.out)
                errs = dict(self.leg_errors or {})
                errs["(extra legs)"] = f"not finished within {self.budget_s:.0f} s of the timed region: dropped, the headline stands"
                line["leg_errors"] = errs
                try:
                    text = json.dumps(line)
                except (TypeError, ValueError):  # a half-built rider
                    text = json.dumps({k: v for k, v in line.items() if k in HEADLINE_KEYS or k == "leg_errors"})
                print(text, flush=True)
            sys.stderr.write(f"bench.py: rank {self.rank}: extra legs over their budget of {self.budget_s:.0f} s; leaving\n")
            sys.stderr.flush()
            os._exit(0)

    def finish(self):
        """True when the caller may print (the timer has not fired and never will)."""
        with self.lock:
            if self.done:
                return False
            self.done = True
            self.timer.cancel()
            return True


HEADLINE_KEYS = ("metric", "value", "unit", "n_gpus", "rccl_ranks", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline", "stage_ms_per_step", "pipeline_hbm_frac")


# the routes the 4096-row headline takes by the library's size rules, forced on the short CPU-parity strip (bench parity leg)
PARITY_ROUTES = (("SGM8_FAM", "1"), ("SGM8_HPAIR", "1"), ("SGM8_CODES", "0"), ("SGM8_FAMCODES", "0"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--height", type=int, default=4096)
    ap.add_argument("--width", type=int, default=4096)
    ap.add_argument("--dmin", type=int, default=0)
    ap.add_argument("--dmax", type=int, default=256)
    ap.add_argument("--cpu-rows", type=int, default=512, help="rows of the CPU-baseline strip (0 = skip)")
    ap.add_argument("--placement-trials", type=int, default=None,
                    help="candidates probed for every new volume-sized buffer (pmx_set_placement_trials; 1 = plain hipMalloc).  Not given: "
                         "the LIBRARY's default (6 since round 6) - `value` is what any caller of pmx_create gets; the line also carries "
                         "`plain_hipmalloc`: the same step with the placement switched off")
    ap.add_argument("--tuned-trials", type=int, default=6, help="candidates of the `placement_tuned` extra leg when --placement-trials is 1")
    ap.add_argument("--no-c3", action="store_true", help="skip the 2048x2048x129 leg (BASELINE configs[2])")
    ap.add_argument("--no-configs", action="store_true", help="skip the c4_as_stated / c5_as_stated / default_allocation legs (N=1)")
    ap.add_argument("--no-dshard", action="store_true", help="skip the exact D-sharded leg (N>1)")
    ap.add_argument("--no-weak", action="store_true", help="skip the pair-per-rank (weak scaling) leg (N>1)")
    ap.add_argument("--no-c5tiled", action="store_true", help="skip the row-tiled BASELINE configs[4] leg (N>1)")
    ap.add_argument("--no-c4tiled", action="store_true", help="skip the row-tiled BASELINE configs[3] (ZNCC + SGM) leg (N>1)")
    ap.add_argument("--c4-height", type=int, default=4096, help="rows of the row-tiled configs[3] leg (N>1)")
    ap.add_argument("--c4-width", type=int, default=4096, help="columns of the row-tiled configs[3] leg (N>1)")
    ap.add_argument("--c4-dmax", type=int, default=256, help="last disparity of the row-tiled configs[3] leg (N>1; the tests shrink it)")
    ap.add_argument("--c5-height", type=int, default=10000, help="rows of the row-tiled configs[4] leg (N>1)")
    ap.add_argument("--c5-width", type=int, default=10000, help="columns of the row-tiled configs[4] leg (N>1)")
    ap.add_argument("--extras-budget", type=float, default=480.0,
                    help="seconds the extra legs / configurations may take after the timed region before the line is printed without them (0: no limit)")
    ap.add_argument("--watchdog", type=float, default=1200.0, help="seconds after which a self-launched multi-rank run is stopped")
    ap.add_argument("--test-fail-rider", default=None, help="TEST HOOK: this rider of the line (c3_shape) raises")
    ap.add_argument("--test-die-rank", type=int, default=None, help="TEST HOOK: this rank exits with code 3 after the warm-up")
    ap.add_argument("--test-comm", default=None, metavar="MODULE:CLASS",
                    help="TEST HOOK: a pandora_amd.comm.Comm subclass from tests/ (e.g. tests.transports:TcpComm) that carries the "
                         "exchange steps through the host, so that several ranks can share the one GPU of a test box")
    ap.add_argument("--test-device", type=int, default=None, help="TEST HOOK: every rank uses this device (with --test-comm)")
    args = ap.parse_args()

    from pandora_amd import _lib
    from pandora_amd.comm import Comm, env_world
    from pandora_amd.dist import row_tile
    from pandora_amd.engine import Engine

    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    ndev = _lib.lib().pmx_device_count()
    if args.test_device is None and ndev < args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} needs {args.gpus} devices, this box shows {ndev}")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, args.watchdog))  # no launcher: bench.py starts its own ranks and relays rank 0's line
    rank, world, local_rank, _, _ = env_world()
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher's WORLD_SIZE is {world}: refusing to print a line for the wrong N")
    if args.test_device is not None:
        local_rank = args.test_device
    H, W, dmin, dmax, win, P1, P2 = args.height, args.width, args.dmin, args.dmax, 5, 8.0, 32.0
    D = dmax - dmin + 1
    eng = Engine(local_rank)
    comm = None
    if world > 1:
        if args.test_comm:
            import importlib

            mod, cls = args.test_comm.split(":")
            comm = getattr(importlib.import_module(mod), cls)(eng)
        else:
            comm = Comm(eng)
        if comm.nranks != world:
            sys.exit(f"bench.py: the communicator reports {comm.nranks} ranks, WORLD_SIZE is {world}")
    if args.placement_trials is not None:  # (not given: whatever pmx_create set up - the default a plugin user runs with)
        eng.set_placement_trials(args.placement_trials)
    placed = args.placement_trials is None or args.placement_trials > 1

    # the SAME pair on every rank (strong scaling); a rank keeps its rows + margin resident
    L, R = synthetic_pair(H, W, dmin, dmax)
    (own_lo, own_hi), (tile_lo, tile_hi) = row_tile(H, world, rank, SGM_MARGIN if world > 1 else 0)
    eng.set_images(np.ascontiguousarray(L[tile_lo:tile_hi]), np.ascontiguousarray(R[tile_lo:tile_hi]), 1)
    cv = eng.alloc_cv(D, dmin)

    def step():
        run_pipeline(eng, cv, win, P1, P2)
        if comm is not None:
            eng.tile_place(H, own_lo, own_hi, tile_lo, True)
            comm.gather_rows(H, True, root=0)

    for _ in range(args.warmup):
        step()
    eng.sync()
    if args.test_die_rank is not None and args.test_die_rank == rank:
        os._exit(3)  # TEST HOOK: a rank that dies mid-run; the launcher must notice and stop the others
    eng.set_profiling(True)
    eng.reset_stage_times()

    def barrier():
        eng.sync()
        if comm is not None:
            comm.barrier()
        eng.sync()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    eng.sync()
    elapsed = time.perf_counter() - t0
    if comm is not None:
        elapsed = float(comm.host_allreduce(np.array([elapsed]), "max")[0])
    barrier()

    cells = H * W * D
    stage = {name: eng.stage_time(name) for name in STAGES}
    eng.set_profiling(False)
    gathered = eng.get_full_maps(H, want_itp=True) if comm is not None and rank == 0 else None  # the last step's maps, before anything else runs
    # The extra legs of an N > 1 run: a leg that fails (on any rank) is dropped from the line with its error, it does not take the
    # headline with it.  (Every rank agrees on the outcome through one small reduction; a rank that hangs inside a collective
    # is left behind by the HeadlineGuard's timer below, which prints the line as far as it got.)
    leg_errors = {}
    out = None

    leg_seconds = {}
    trace = os.environ.get("BENCH_TRACE")  # debugging aid: every rank says on stderr when it enters / leaves a leg

    def leg(name, fn):
        err = None
        res = None
        t_leg = time.perf_counter()
        if trace:
            print(f"[bench rank {rank}] {time.strftime('%H:%M:%S')} -> {name}", file=sys.stderr, flush=True)
        try:
            res = fn()
        except Exception as e:  # noqa: BLE001 - reported in the line
            err = f"{type(e).__name__}: {e}"[:300]
        leg_seconds[name] = round(time.perf_counter() - t_leg, 1)
        if trace:
            print(f"[bench rank {rank}] {time.strftime('%H:%M:%S')} <- {name} {leg_seconds[name]} s {err or ''}", file=sys.stderr, flush=True)
        failed = float(comm.host_allreduce(np.array([1.0 if err else 0.0]), "max")[0]) > 0
        if failed:
            leg_errors[name] = err or "failed on another rank"
            return None
        return res

    class rider:  # a rider of the line that fails is dropped from it with its error; the headline stands
        def __init__(self, name):
            self.name = name

        def __enter__(self):
            return self

        def __exit__(self, et, ev, tb):
            if et is not None and issubclass(et, Exception):
                leg_errors[self.name] = f"{et.__name__}: {ev}"[:300]
                return True
            return False

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = cells / (elapsed / args.steps) / 1e6
        tile_cells = (tile_hi - tile_lo) * W * D  # what this rank's kernels worked on
        roof = roofline_block(stage, args.steps, tile_cells)
        wtraffic = add_traffic(roof, "headline", H, W, D) if stage["sgm_fused"][1] > 0 and world == 1 else None
        if world == 1:
            parallelism = "1 GPU, no collective"
        else:
            parallelism = (f"one pair over {world} GPUs: row tiles of {H // world} rows + {SGM_MARGIN}-row margin (the reference's ROI "
                           f"convention for SGM, paths cut at the margin), one RCCL gather (ncclSend / ncclRecv group) of the owned rows of "
                           f"disparity / validity / coefficient maps to GPU 0 per step, on a stream of its own (it runs under the next step's kernels; "
                           f"the last step's is inside the timed region); strong scaling")
        out = {
            "metric": "Mdisparities/s (HxWxD/s) Census5x5+SGM",
            "value": round(value, 1),
            "unit": "Mdisp/s",
            "n_gpus": world,
            "rccl_ranks": 1,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": ("u8 storage / packed u16 arithmetic (exact; float32-identical)" if stage["sgm_fused"][1] > 0 else "f32"),
            "data": "synthetic",
            "config": {"workload": f"{H}x{W} synthetic pair, d=[{dmin},{dmax}] (D={D}), Census5x5 + SGM 8-path (P1=8,P2=32) + WTA + vfit; "
                                   f"ONE pair per step at every N", "parallelism": parallelism},
            "roofline": roof,
            "stage_ms_per_step": {k: round(v[0] / args.steps, 4) for k, v in stage.items()},
            "pipeline_hbm_frac": round(PIPELINE_ALGO_BYTES_PER_CELL * cells / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS / world, 4),
        }
        if world == 1 and stage["sgm_span"][1] > 0:
            own_format(out, wtraffic, H, W, D, ms_per_step)
        if world == 1:
            # "measure achievable with a device memcpy/triad on the box and report both" - plain 16-byte-per-lane
            # streams over 4 GB in this run, on this box (pmx_measure_hbm), beside the 8000 GB/s of the data sheet
            hbm = eng.measure_hbm(4 << 30)
            peak_m = max(hbm.values())
            roof["peak_measured"] = round(peak_m, 1)
            roof["peak_measured_streams"] = {k: round(v, 1) for k, v in hbm.items()}
            roof["peak_measured_note"] = "GB/s of a 4 GB fill / read / copy (copy counts both directions) in this run; peak_measured = the best"
            if peak_m > 0:
                roof["frac_of_measured"] = round(roof["achieved"] / peak_m, 4)
                if "frac_counted" in roof:
                    roof["frac_counted_of_measured"] = round(roof["frac_counted"] * HBM_PEAK_GBS / peak_m, 4)
                    if hbm["copy"] > 0:  # (the SGM step reads and writes: the copy stream is its like-for-like yardstick)
                        roof["frac_counted_of_measured_copy"] = round(roof["frac_counted"] * HBM_PEAK_GBS / hbm["copy"], 4)
                if "pipeline_hbm_frac_counted" in out:
                    out["pipeline_hbm_frac_counted_of_measured"] = round(out["pipeline_hbm_frac_counted"] * HBM_PEAK_GBS / peak_m, 4)
        if world > 1:
            # RCCL's own rank count (ncclCommCount) - or, under the --test-comm hook, what carried the exchange instead
            if getattr(comm, "nranks_note", None):
                del out["rccl_ranks"]
                out["transport"] = f"{comm.nranks_note} ({type(comm).__name__}, {comm.world} ranks): NOT RCCL"
            else:
                out["rccl_ranks"] = comm.nranks
            out["collective"] = {"kind": "ncclSend / ncclRecv group: owned rows of 3 maps to rank 0 (10 B/pixel, validity as uint16)",
                                 "ms_per_step": round(stage["collective"][0] / args.steps, 4), "bytes_per_step": H * W * 10}
    # from here on nothing may take the line: a timer on every rank prints what there is and leaves (HeadlineGuard)
    guard = HeadlineGuard(rank, args.extras_budget)
    guard.arm(out, leg_errors)
    dshard = weak = c5tiled = c4tiled = None
    if comm is not None:
        cv.free()  # (the extra legs bring their own volumes)
    if comm is not None and not args.no_c5tiled:
        c5tiled = leg("c5_row_tiled", lambda: row_tiled_leg(eng, comm, "c5", args.c5_height, args.c5_width, max(2, args.steps // 4), local_rank))
    if comm is not None and not args.no_c4tiled:
        c4tiled = leg("c4_row_tiled", lambda: row_tiled_leg(eng, comm, "c4", args.c4_height, args.c4_width, max(2, args.steps // 4), local_rank,
                                                              c4_dmax=args.c4_dmax))
    if comm is not None and not args.no_dshard and D >= 2 * world:
        dshard = leg("d_sharded_exact", lambda: d_sharded_leg(eng, comm, L, R, dmin, dmax, 11, max(2, args.steps // 2), local_rank))
    if comm is not None and not args.no_weak:
        weak = leg("pair_per_rank", lambda: pair_per_rank_leg(comm, local_rank, H, W, dmin, dmax, win, P1, P2, max(2, args.steps // 2), rank, world))

    if rank == 0:
        if world > 1:
            # what arrived (outside the timed region): the gathered maps of the last step against ONE GPU doing the whole pair.
            # Tiles cut the SGM paths at their 40-row margin, like the reference's ROI tiling: a fraction of a percent of the pixels
            # near the seams may differ, everything else must be identical - anything else means the exchange is broken.
            with rider("gathered_maps_vs_one_gpu"):
                gd, gv, gi = gathered
                one = Engine(local_rank)
                one.set_images(L, R, 1)
                cv1 = one.alloc_cv(D, dmin)
                run_pipeline(one, cv1, win, P1, P2)
                od, ov, oi = one.get_disparity(want_itp=True)
                cv1.free()
                one.close()
                out["gathered_maps_vs_one_gpu"] = {
                    "disparity_identical": round(float(np.mean((gd == od) | (np.isnan(gd) & np.isnan(od)))), 6),
                    "validity_identical": round(float(np.mean(gv == ov)), 6),
                    "coefficient_identical": round(float(np.mean((gi == oi) | (np.isnan(gi) & np.isnan(oi)))), 6)}
            if dshard is not None:
                out["d_sharded_exact"] = dshard
            if weak is not None:
                out["pair_per_rank"] = weak
            if c5tiled is not None:
                out["c5_row_tiled"] = c5tiled
            if c4tiled is not None:
                out["c4_row_tiled"] = c4tiled
            if leg_errors:
                out["leg_errors"] = leg_errors
            out["leg_seconds"] = leg_seconds
        else:
            # PCIe-inclusive rate (never `value`): host images in, the three 2-D result maps out, one step, after the barrier
            pcie_s = pcie_inclusive_ms(eng, cv, L, R, win, P1, P2) * 1e-3
            out["pcie_inclusive"] = {"ms_per_step": round(pcie_s * 1e3, 3), "value": round(cells / pcie_s / 1e6, 1), "unit": "Mdisp/s",
                                     "note": "pmx_set_images (2 float32 images up) + pipeline + pmx_get_disparity (disp, validity "
                                             "int64, itp down); informational only"}
            cv.free()
            with rider("c3_shape"):
                if args.test_fail_rider == "c3_shape":
                    raise RuntimeError("test hook: --test-fail-rider")
                if not args.no_c3 and (H, W, D) != (2048, 2048, 129):
                    ms3, st3, _ = measure_shape(eng, 2048, 2048, 0, 128, args.steps, args.warmup, 20260928)
                    pcie3 = st3.pop("(pcie inclusive)")[0]
                    c3 = 2048 * 2048 * 129
                    r3 = roofline_block(st3, args.steps, c3)
                    if st3["sgm_fused"][1] > 0:
                        add_traffic(r3, "c3", 2048, 2048, 129)
                    out["c3_shape"] = {"workload": "2048x2048 synthetic pair, d=[0,128] (D=129): BASELINE configs[2], the round-1 headline; same "
                                                   "pipeline and protocol", "steps": args.steps, "ms_per_step": round(ms3, 3),
                                       "value": round(c3 / ms3 / 1e3, 1), "unit": "Mdisp/s", "roofline": r3,
                                       "stage_ms_per_step": {k: round(v[0] / args.steps, 4) for k, v in st3.items()},
                                       "pipeline_hbm_frac": round(PIPELINE_ALGO_BYTES_PER_CELL * c3 / (ms3 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                       "pcie_inclusive_ms": round(pcie3, 3)}
            if not args.no_configs and (H, W, D) == (4096, 4096, 257):
                # the headline step with the library's placement switched off (plain hipMalloc: the round-5 default), or - when `value`
                # itself was measured that way - on a context that probes `--tuned-trials` candidates per volume:
                with rider("plain_hipmalloc"):
                    if placed:
                        plain = Engine(local_rank)
                        plain.set_placement_trials(1)
                        msd, std, _ = measure_shape(plain, H, W, dmin, dmax, args.steps, args.warmup, 20260928, pcie=False)
                        out["plain_hipmalloc"] = {"placement_trials": 1, "ms_per_step": round(msd, 3), "value": round(cells / msd / 1e3, 1),
                                                  "unit": "Mdisp/s", "note": "same workload and protocol on a fresh context with "
                                                  "pmx_set_placement_trials(ctx, 1): plain hipMalloc, what every caller got until round 5"}
                        plain.close()
                    elif args.tuned_trials > 1:
                        tuned = Engine(local_rank)
                        tuned.set_placement_trials(args.tuned_trials)
                        mst, _, _ = measure_shape(tuned, H, W, dmin, dmax, args.steps, args.warmup, 20260928, pcie=False)
                        out["placement_tuned"] = {"placement_trials": args.tuned_trials, "ms_per_step": round(mst, 3),
                                                  "value": round(cells / mst / 1e3, 1), "unit": "Mdisp/s",
                                                  "note": "same workload and protocol on a fresh context with pmx_set_placement_trials (opt-in)"}
                        tuned.close()
                # BASELINE configs[3] and configs[4] as stated, float32 between the steps, one GPU
                eng.set_placement_trials(1)  # (six candidates of a 51.6 GB volume would not fit the device)
                with rider("c4_as_stated"):
                    out["c4_as_stated"] = config_leg(eng, L, R, dmin, dmax, ("zncc", 11), False, 3,
                                                     "BASELINE configs[3] as stated: 4096x4096 synthetic pair, d=[0,256] (D=257), ZNCC 11x11 + SGM 8-path "
                                                     "+ WTA + vfit, float32 cost volume between the steps; one GPU", "c4")
                # BASELINE configs[1] on the reference's own cones pair (tests/golden/cones: data, not code): census + CBCA + SGM + WTA +
                # vfit, d = [-60, 0] as data_samples/json_conf_files/a_semi_global_matching.json has it for this pair
                with rider("c2_cones"):
                    cones_dir = os.path.join(ROOT, "tests", "golden", "cones")
                    if os.path.exists(os.path.join(cones_dir, "left.png")):
                        from PIL import Image

                        Lc = np.array(Image.open(os.path.join(cones_dir, "left.png"))).astype(np.float32)
                        Rc = np.array(Image.open(os.path.join(cones_dir, "right.png"))).astype(np.float32)
                        out["c2_cones"] = config_leg(eng, Lc, Rc, -60, 0, ("census", 5), True, 20,
                                                     "BASELINE configs[1]: the cones pair (375x450), d=[-60,0] (D=61), Census 5x5 + CBCA + SGM 8-path "
                                                     "+ WTA + vfit; one GPU (a launch-latency-sized problem: 10.3 M cells)", "c2")
                with rider("c5_as_stated"):
                    L5, R5 = synthetic_pair(10000, 10000, -64, 64)
                    out["c5_as_stated"] = config_leg(eng, L5, R5, -64, 64, ("census", 5), True, 2,
                                                     "BASELINE configs[4], fine scale, as stated: 10000x10000 synthetic pair, d=[-64,64] (D=129), "
                                                     "Census 5x5 + CBCA + SGM 8-path + WTA + vfit, float32 cost volume between the steps; the whole "
                                                     "strip on one GPU", "c5")
                    del L5, R5
            with rider("cpu_baseline"):
                if args.cpu_rows > 0:  # the CPU legs belong to the N=1 line only
                    rows = min(args.cpu_rows, H)
                    base, (cdisp, cval) = cpu_baseline(L, R, dmin, dmax, win, P1, P2, rows)
                    out["cpu_baseline"] = base
                    # all host cores: probe on a short strip first, so that a box whose cores are not really available (container
                    # quota, oversubscription) costs seconds, not minutes; the full strip only when the threads pay off
                    probe_rows = min(rows, 64)
                    probe, _ = cpu_baseline(L, R, dmin, dmax, win, P1, P2, probe_rows, threads=0)
                    if probe["value"] > 1.5 * base["value"]:
                        out["cpu_baseline_all_cores"], (mdisp, mval) = cpu_baseline(L, R, dmin, dmax, win, P1, P2, rows, threads=0)
                        assert np.array_equal(mdisp, cdisp, equal_nan=True) and np.array_equal(mval, cval)  # thread count changes nothing
                    else:
                        out["cpu_baseline_all_cores"] = probe
                    refc = cpu_reference_compiled(L, R, dmin, dmax, win, min(rows, 256))
                    if refc is not None:
                        out["cpu_baseline_reference_compiled"] = refc
                    # parity in the same run: the same strip through the GPU path (vertical paths see only the strip, so the GPU is
                    # re-run on the strip alone)
                    # ... with the TIMED step's kernels: at this height the library's own size rules would take the row walk and the
                    # code-word forms (csrc/k_sgm8.hip pmx_launch_sgm8), so the headline's routes are forced on the strip: the cost
                    # volume, the one-sided horizontal pair, the direction families (roofline.kernel names them)
                    eng2 = Engine(local_rank)
                    for name, val in PARITY_ROUTES:
                        eng2.set_option(name, val)
                    eng2.set_images(L[:rows], R[:rows], 1)
                    cv2 = eng2.alloc_cv(D, dmin)
                    eng2.set_profiling(True)
                    eng2.reset_stage_times()
                    eng2.census(cv2, win)
                    eng2.sgm(cv2, P1, P2, False, float(win * win + 1), False)
                    eng2.set_validity(None)
                    eng2.wta(cv2, False, -9999.0)
                    gdisp, gval = eng2.get_disparity()
                    ran = {k: eng2.stage_time(k)[1] for k in ("census_cost", "sgm_fused", "sgm_family", "wta")}
                    eng2.set_profiling(False)
                    out["disparity_linf_vs_cpu"] = float(np.max(np.abs(gdisp - cdisp)))
                    out["disparity_linf_vs_cpu_kernels"] = (
                        "the strip through the timed step's instantiation (" + ", ".join(f"{n}={v}" for n, v in PARITY_ROUTES) + "): "
                        "census_cost_u8_kernel, sgm_u8_hpair_kernel beside sgm_fam8_kernel, sum3_wta_kernel; launches counted on the strip: "
                        + json.dumps(ran))
                    assert ran["census_cost"] and ran["sgm_fused"] and ran["sgm_family"], "the parity strip did not run the headline's kernels"
                    eng2.close()
        if world == 1 and leg_errors:
            out["leg_errors"] = leg_errors
        if guard.finish():
            print(json.dumps(out), flush=True)
    if comm is not None:
        comm.barrier()
        comm.close()
    guard.finish()
    eng.close()


if __name__ == "__main__":
    main()
