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


@pytest.mark.parametrize("world,port", [(2, 29547), (8, 29561)])
def test_bench_runs_one_pair_over_the_ranks(world, port):
    """--gpus N = ONE pair over N ranks (row tiles + 40-row margin, gather of the owned rows on rank 0), launched with the launcher's
    environment variables; the ranks share the box's one GPU, so the exchange goes through the tcp test transport.  Eight ranks over
    300 rows: tiles of 37 / 38 owned rows whose margins reach over several neighbours."""
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   PANDORA_COMM_BACKEND="tcp", PANDORA_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
                                       "--height", "300", "--width", "256", "--dmax", "40", "--placement-trials", "1"],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT))
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), "".join(o[1][-1500:] for o in outs)
    lines = [ln for ln in outs[0][0].splitlines() if ln.strip()]
    assert len(lines) == 1 and not any(o[0].strip() for o in outs[1:]), outs
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["scaling"] == "strong" and "row tiles" in d["config"]["parallelism"]
    assert d["collective"]["bytes_per_step"] == 300 * 256 * 10
    # what arrived on rank 0 is the pair's result: identical to one GPU doing the whole pair except near the tile seams (SGM paths
    # are cut at the 40-row margin, as in the reference's ROI tiling; with 8 ranks over 300 rows there is a seam every 37 rows)
    g = d["gathered_maps_vs_one_gpu"]
    assert g["disparity_identical"] > (0.97 if world == 2 else 0.85) and g["validity_identical"] > 0.97, g
