"""GPU: bench.py honours the driver's contract (one JSON line with the agreed keys, roofline and cpu_baseline objects,
a parity figure from the same run) on a small workload."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_contract_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                          "--height", "192", "--width", "256", "--dmax", "40", "--cpu-rows", "64"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["scaling"] == "strong" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] == 1 and c["value"] > 0 and c["sample"]
    assert d["disparity_linf_vs_cpu"] == 0.0  # the GPU strip equals the CPU oracle strip, same run
    assert d["value"] > c["value"]
    assert d["cpu_baseline_reference_compiled"]["kind"] == "reference-compiled"  # the reference's own census C++, same strip


@pytest.mark.parametrize("world,port,height,width", [(2, 29547, 300, 256), (8, 29561, 300, 256), (2, 29583, 1000, 2600)])
def test_bench_runs_one_pair_over_the_ranks(world, port, height, width):
    """--gpus N = ONE pair over N ranks (row tiles + 40-row margin, gather of the owned rows on rank 0), launched with the launcher's
    environment variables; the ranks share the box's one GPU, so the exchange goes through the TcpComm stand-in of tests/transports.py
    (bench.py's explicit --test-comm hook; the product Comm is RCCL only).  Eight ranks over
    300 rows: tiles of 37 / 38 owned rows whose margins reach over several neighbours.  1000 x 2600 over two ranks: tiles of 540 rows,
    wide enough for the integer path's direction-family form (marching kernel + two-sided horizontal walk) inside every tile."""
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
                                       "--height", str(height), "--width", str(width), "--dmax", "40", "--placement-trials", "1",
                                       "--c5-height", "640", "--c5-width", "700", "--test-comm", "tests.transports:TcpComm", "--test-device", "0"],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT))
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), "".join(o[1][-1500:] for o in outs)
    lines = [ln for ln in outs[0][0].splitlines() if ln.strip()]
    assert len(lines) == 1 and not any(o[0].strip() for o in outs[1:]), outs
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["scaling"] == "strong" and "row tiles" in d["config"]["parallelism"]
    # the line says what carried the exchange: RCCL's own rank count only when RCCL ran it, the stand-in's name otherwise
    assert "rccl_ranks" not in d and "NOT RCCL" in d["transport"] and f"{world} ranks" in d["transport"], d.get("transport")
    # the exact multi-GPU form rides along: costs sharded over D, one all-reduce(min) of packed keys - identical to one GPU
    assert d["d_sharded_exact"]["maps_identical_to_one_gpu"] == 1.0, d["d_sharded_exact"]
    assert d["collective"]["bytes_per_step"] == height * width * 10
    # ... and the weak-scaling figure: one whole pair per rank and step, no exchange
    assert d["pair_per_rank"]["scaling"] == "weak" and d["pair_per_rank"]["value"] > 0
    if width >= 2560:
        assert d["stage_ms_per_step"]["sgm_span"] > 0  # the family form ran in the tiles
    # BASELINE configs[4] as worded (census + CBCA + SGM, row-tiled), here on a 640 x 700 strip: the gathered maps against one GPU
    t = d["c5_row_tiled"]
    assert "row-tiled" in t["workload"] and "CBCA" in t["workload"] and t["value"] > 0 and t["one_gpu_ms_per_step"] > 0
    assert f"row tiles of {640 // world} rows + 42-row margin" in t["parallelism"], t["parallelism"]
    assert t["gathered_maps_vs_one_gpu"]["disparity_identical"] > (0.97 if world == 2 else 0.85), t
    assert t["gathered_maps_vs_one_gpu"]["validity_identical"] > 0.97, t
    # what arrived on rank 0 is the pair's result: identical to one GPU doing the whole pair except near the tile seams (SGM paths
    # are cut at the 40-row margin, as in the reference's ROI tiling; with 8 ranks over 300 rows there is a seam every 37 rows)
    g = d["gathered_maps_vs_one_gpu"]
    assert g["disparity_identical"] > (0.97 if world == 2 else 0.85) and g["validity_identical"] > 0.97, g
    if height >= 1000:
        assert g["disparity_identical"] > 0.99, g


def test_bench_refuses_to_mislabel_a_run():
    """`--gpus 2` on a box with one device must fail loudly (no 1-GPU line labelled as 2), and so must a --gpus that disagrees with
    the launcher's WORLD_SIZE."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    from pandora_amd import _lib
    if _lib.lib().pmx_device_count() < 2:
        assert out.returncode != 0 and "needs 2 devices" in out.stderr and not out.stdout.strip(), (out.stdout, out.stderr[-500:])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert out.returncode != 0 and "WORLD_SIZE" in out.stderr and not out.stdout.strip()


def test_launcher_fails_fast_when_a_rank_dies():
    """`bench.py --gpus 2` launching its own ranks: rank 1 exits after the warm-up (--test-die-rank), rank 0 is left waiting in the
    exchange of its next step; the launcher must stop it and exit non-zero within seconds, not after the transport's timeout."""
    import time

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--height", "300",
                          "--width", "256", "--dmax", "40", "--placement-trials", "1", "--no-dshard", "--no-weak", "--no-c5tiled",
                          "--no-c4tiled", "--test-comm", "tests.transports:TcpComm", "--test-device", "0", "--test-die-rank", "1"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    took = time.time() - t0
    # (rank 0 may itself fail on the closed connection before the launcher looks: the code is the first named rank's)
    assert out.returncode != 0, (out.returncode, out.stderr[-1500:])
    assert not out.stdout.strip(), out.stdout
    assert "rank 1 exited with code 3" in out.stderr and "stopping the other ranks" in out.stderr, out.stderr[-1500:]
    assert took < 30.0, took


def test_launcher_runs_its_own_ranks_to_the_end():
    """the same launcher without the fault: one JSON line from rank 0, exit code 0"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--height", "300",
                          "--width", "256", "--dmax", "40", "--placement-trials", "1", "--no-dshard", "--no-weak", "--no-c5tiled",
                          "--c4-height", "400", "--c4-width", "320", "--c4-dmax", "40", "--test-comm", "tests.transports:TcpComm", "--test-device", "0"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-1500:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2, out.stdout
    # BASELINE configs[3]'s pipeline WITH its SGM step over the ranks' row tiles (ZNCC 11x11 + SGM + WTA + vfit; d = [0, 40] here): the leg is
    # in the line, and what the two tiles gathered is the one-GPU result except near the seam (SGM paths cut at the 40-row margin)
    d = json.loads(lines[0])
    assert "leg_errors" not in d, d.get("leg_errors")
    c4 = d["c4_row_tiled"]
    assert "ZNCC 11x11 + SGM" in c4["workload"] and c4["ms_per_step"] > 0 and c4["stage_ms_per_step_rank0"]["zncc"] > 0
    # (float costs: a tile's sums differ from the whole image's in their last bits, the refined disparity with them; the winner and the
    # disparity to 0.01 px are what the tiling keeps)
    g = c4["gathered_maps_vs_one_gpu"]
    assert g["winner_identical"] > 0.95 and g["disparity_within_0.01"] > 0.7 and g["validity_identical"] > 0.97, g  # (tiles of 200 rows: half the image is within a margin of the seam)


def test_bench_prices_the_integer_route_against_its_own_format():
    """A pair wide enough for the direction-family form (three byte volumes): the line says what the route's own volumes need
    (`own_format_bytes_per_cell`) beside the float32-priced `frac`; counted figures appear only when committed counter passes of
    this very workload and these very sources exist (they do not for this shape: `traffic` stays null)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--height", "520",
                          "--width", "2600", "--dmax", "40", "--cpu-rows", "0", "--no-c3", "--no-configs", "--placement-trials", "1"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.strip()][0])
    assert d["stage_ms_per_step"]["sgm_span"] > 0  # the family form ran
    # D = 41: KPL 4, 11 lanes: Dp = 44 bytes, Dc = 11 dwords = 44 bytes per pixel -> (2 * 44 + 6 * 44) / 41
    assert abs(d["own_format_bytes_per_cell"] - (2 * 44 + 6 * 44) / 41) < 1e-3
    assert d["roofline"]["traffic"] is None and "traffic_amplification" not in d
    # what plain streams reach on this box, measured in the same run, beside the data sheet's peak
    r = d["roofline"]
    assert r["peak"] == 8000.0 and 2000.0 < r["peak_measured"] < 8000.0 and set(r["peak_measured_streams"]) == {"read", "write", "copy"}
    assert abs(r["frac_of_measured"] - r["achieved"] / r["peak_measured"]) < 1e-3


def test_the_headline_survives_extra_legs_that_do_not_finish():
    """bench.py --extras-budget: when the riders of the line (the BASELINE configurations at N = 1, the row-tiled legs at N > 1)
    are not done within the budget after the timed region, rank 0 prints the line as far as it got - the headline whole, the reason
    under leg_errors - and every rank leaves with status 0.  A budget of a few milliseconds makes that happen on purpose."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--height", "192",
                          "--width", "256", "--dmax", "40", "--cpu-rows", "64", "--extras-budget", "0.005"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    two = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--height", "300",
                          "--width", "256", "--dmax", "40", "--placement-trials", "1", "--no-dshard", "--no-weak", "--c5-height", "640",
                          "--c5-width", "700", "--c4-height", "400", "--c4-width", "320", "--c4-dmax", "40", "--extras-budget", "0.005",
                          "--test-comm", "tests.transports:TcpComm", "--test-device", "0"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    for out, n in ((one, 1), (two, 2)):
        assert out.returncode == 0, out.stderr[-1500:]
        lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, out.stdout
        d = json.loads(lines[0])
        assert d["n_gpus"] == n and d["value"] > 0 and d["ms_per_step"] > 0 and d["roofline"]["achieved"] > 0
        assert "(extra legs)" in d["leg_errors"], d["leg_errors"]
        assert "over their budget" in out.stderr


def test_a_failing_rider_leaves_the_line():
    """a configuration leg that raises is dropped from the line with its error; the headline and the other riders stay"""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--height", "192",
                          "--width", "256", "--dmax", "40", "--cpu-rows", "64", "--test-fail-rider", "c3_shape"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-1500:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.strip()][-1])
    assert "c3_shape" not in d and "test hook" in d["leg_errors"]["c3_shape"] and d["cpu_baseline"]["value"] > 0, d.get("leg_errors")
