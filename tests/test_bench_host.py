"""CPU: the host-side pieces of bench.py and tools/pmc_traffic.py that need no GPU - the own-format figure, the counted-traffic
lookup (stamped with a hash of the native sources), the rendezvous port pair, the parsing of a rocprofv3 counter summary."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_own_format_bytes_of_the_headline_lane_map():
    out = {}
    bench.own_format(out, None, 4096, 4096, 257, 13.7)
    # D = 257: 13 lanes x 20 disparities: Dp = 260 bytes, five-bit costs 4 dwords per lane: Dc = 208 bytes per pixel
    assert abs(out["own_format_bytes_per_cell"] - (2 * 208 + 6 * 260) / 257) < 1e-3
    assert "traffic_amplification" not in out  # no committed counters given: nothing is made up
    w = {"step_hbm_bytes": 59_192_298_086}
    bench.own_format(out, w, 4096, 4096, 257, 13.7)
    cells = 4096 * 4096 * 257
    assert abs(out["counted_bytes_per_cell"] - w["step_hbm_bytes"] / cells) < 1e-3
    assert abs(out["traffic_amplification"] - w["step_hbm_bytes"] / cells / out["own_format_bytes_per_cell"]) < 2e-3
    assert 0 < out["pipeline_hbm_frac_counted"] < 1


def test_counted_traffic_is_quoted_only_for_the_sources_it_was_counted_on(tmp_path, monkeypatch):
    sha = bench.kernel_source_hash()
    assert len(sha) == 16 and sha == bench.kernel_source_hash()
    prof = tmp_path / "profiles"
    prof.mkdir()
    entry = {"workload": {"label": "headline", "H": 8, "W": 8, "D": 5}, "sgm_hbm_bytes_per_step": 1000, "step_hbm_bytes": 1500,
             "commit": "abc", "kernel_source_sha16": sha}
    stale = dict(entry, workload={"label": "c4", "H": 8, "W": 8, "D": 5}, kernel_source_sha16="0" * 16)
    (prof / "r99_pmc_traffic.json").write_text(json.dumps([entry, stale]))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: sha)
    roof = {"avg_launch_ms": 1.0, "traffic": None}
    w = bench.add_traffic(roof, "headline", 8, 8, 5)
    assert w is not None and roof["traffic"] == 1000 and roof["frac_counted"] == round(1000 / 1e-3 / 1e9 / bench.HBM_PEAK_GBS, 4)
    roof2 = {"avg_launch_ms": 1.0, "traffic": None}
    assert bench.add_traffic(roof2, "c4", 8, 8, 5) is None and roof2["traffic"] is None        # other sources
    assert bench.add_traffic(dict(roof2), "headline", 8, 8, 6) is None                          # other workload


def test_the_committed_traffic_file_matches_the_built_sources():
    """profiles/r04_pmc_traffic.json must have been taken on the sources in the tree (else bench.py quotes no counted bytes)"""
    w = bench.counted_traffic("headline", 4096, 4096, 257)
    if w is None:
        import pytest

        pytest.skip("the native sources changed after the last counter passes: re-run tools/profile_round.sh")
    assert w["step_hbm_bytes"] > w["sgm_hbm_bytes_per_step"] > 0
    assert not any(k.startswith("placement_probe") and not v.get("not_part_of_a_step") for k, v in w["per_kernel"].items())


def test_rendezvous_port_pair_is_free():
    port = bench._free_port_pair()
    for p in (port, port + 1):
        with socket.socket() as s:
            s.bind(("127.0.0.1", p))


def test_pmc_traffic_reads_a_counter_summary(tmp_path):
    csv = tmp_path / "x_pmc_hbm.csv"
    csv.write_text("# comment\nkernel,counter,dispatches,avg_value,units_note\n"
                   '"void sgm_u8_hpair_kernel<20, 5>(sgm8_args)",FETCH_SIZE,4,1000.0,KiB\n'
                   '"void sgm_u8_hpair_kernel<20, 5>(sgm8_args)",WRITE_SIZE,4,500.0,KiB\n'
                   '"sum8_refine_kernel(sum8_args)",FETCH_SIZE,4,10.0,KiB\n"sum8_refine_kernel(sum8_args)",WRITE_SIZE,4,5.0,KiB\n'
                   '"placement_probe_kernel(x)",FETCH_SIZE,6,99999.0,KiB\n"placement_probe_kernel(x)",WRITE_SIZE,6,0.0,KiB\n')
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), "t", str(csv), "4", "4", "3"], capture_output=True,
                         text=True, cwd=ROOT)
    assert out.returncode == 0, out.stderr
    (w,) = json.loads(out.stdout)
    assert w["steps_profiled"] == 4 and w["sgm_hbm_bytes_per_step"] == (2 * 1000 + 500) * 1024
    assert w["step_hbm_bytes"] == (2 * 1000 + 500 + 2 * 10 + 5) * 1024  # the allocation-time probe is listed, not summed
    assert w["per_kernel"]["placement_probe_kernel"]["not_part_of_a_step"] is True
