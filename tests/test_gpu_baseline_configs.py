"""BASELINE.json's configs[3] and configs[4] AS STATED, at full size, on the GPU:

  C4  4096x4096, d=[0,256]: ZNCC 11x11 -> SGM -> WTA -> vfit, through the float32 kernels (17.3 GB volumes);
  C5  10000x10000, d=[-64,64]: census -> CBCA -> SGM -> WTA -> vfit (51.6 GB volumes), the 2-scale run on a strip, row tiles.

The oracle cannot run these sizes, so parity is established through what the domain offers:
  * every path direction against the oracle, EXACTLY, on strips where the oracle's answer does not depend on the rest of the
    image: the two horizontal paths on any rows (a row's horizontal paths see that row alone), the three downward paths on the top
    rows, the three upward paths on the bottom rows (pmx_debug_sgm_directions selects the paths; the oracle gets the GPU's own
    matching costs of the strip, so float32 SGM arithmetic is compared bit for bit);
  * the two float32 schedules that run at these sizes (one launch per path / fused families) agree bit for bit on the maps;
  * a vertically periodic pair gives vertically periodic maps away from the borders."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P1, P2 = 8.0, 32.0
STRIP = 40


@pytest.fixture(scope="module")
def eng():
    from pandora_amd.engine import Engine

    e = Engine(0)
    e.set_lazy(False)
    yield e
    e.close()


def periodic_pair(H, W, shift, seed):
    """period 16 in the rows, so one small tile defines the pair (cheap to build at 10000^2)"""
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 256, (16, W + 24)).astype(np.float32)
    base = np.floor((base + np.roll(base, 1, 1) + np.roll(base, 1, 0)) / 3.0)
    L = base[:, 12:12 + W]
    R = base[:, 12 - shift:12 - shift + W] + rng.integers(-2, 3, (16, W))
    reps = -(-H // 16)
    return (np.ascontiguousarray(np.tile(L, (reps, 1))[:H], np.float32), np.ascontiguousarray(np.tile(R, (reps, 1))[:H], np.float32))


def costs(eng, cv, kind):
    if kind == "zncc11":
        eng.zncc(cv, 11)
    else:
        eng.census(cv, 5)
        eng.cbca(cv, 2, 30.0, 5)


def maps(eng, cv, is_max):
    eng.set_validity(None)
    eng.wta(cv, is_max, -9999.0)
    eng.refine(cv, "vfit", is_max)
    return eng.get_disparity(want_itp=True)


def check_config(eng, oracle, hooks, H, W, dmin, dmax, kind, shift):
    D = dmax - dmin + 1
    is_max = kind == "zncc11"
    inv = 2.0 if is_max else 26.0
    L, R = periodic_pair(H, W, shift, H + D)
    eng.set_images(L, R, 1)
    cv = eng.alloc_cv(D, dmin)
    # ---- every path family against the oracle on the strip where the oracle can know the answer
    hooks.setenv("PMX_SGM_SCHED", "fam")
    for mask, rows in ((0x03, (H // 2, H // 2 + STRIP)), (0x03, (H - STRIP, H)), (0x1C, (0, STRIP)), (0xE0, (H - STRIP, H))):
        costs(eng, cv, kind)
        strip_costs = cv.rows_to_host(*rows)
        eng.sgm(cv, P1, P2, is_max, inv, False, dir_mask=mask)
        got = cv.rows_to_host(*rows)
        exp = oracle.sgm(strip_costs, P1, P2, is_max, inv, False, dir_mask=mask)
        np.testing.assert_array_equal(got, exp, err_msg=f"paths {mask:#x} rows {rows}")
    # ---- the whole pipeline as BASELINE states it, both float32 schedules
    out = {}
    for sched in ("fam", "seq"):
        hooks.setenv("PMX_SGM_SCHED", sched)
        costs(eng, cv, kind)
        eng.sgm(cv, P1, P2, is_max, inv, False)
        out[sched] = maps(eng, cv, is_max)
    # lazy mode: the upward family runs fused with the WTA, the optimised volume is never written
    hooks.setenv("PMX_SGM_SCHED", "fam")
    eng.set_lazy(True)
    costs(eng, cv, kind)
    eng.sgm(cv, P1, P2, is_max, inv, False)
    out["fused"] = maps(eng, cv, is_max)
    eng.set_lazy(False)
    cv.free()
    for a, b in zip(out["fam"], out["seq"]):
        np.testing.assert_array_equal(a, b)
    for a, b in zip(out["fam"], out["fused"]):
        np.testing.assert_array_equal(a, b)
    disp, val, itp = out["fam"]
    o = 5 if is_max else 2
    inner = disp[o:-o, o:-o]
    assert np.isfinite(inner).all() and inner.min() >= dmin and inner.max() <= dmax
    lo = 16 * (H // 64)  # far from the top and bottom borders: the maps repeat with the pair's period (all eight paths included)
    a, b = disp[lo:lo + 160], disp[lo + 160:lo + 320]
    if is_max:  # non-integer costs: float32 path costs carry their history's rounding, so the repetition is not bit for bit
        assert (a == b).mean() > 0.999
        np.testing.assert_allclose(itp[lo:lo + 160][a == b], itp[lo + 160:lo + 320][a == b], rtol=2e-3, atol=1e-3)
    else:       # integer-valued costs: every float32 operation of the recurrence is exact
        np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(itp[lo:lo + 160], itp[lo + 160:lo + 320])
    # the pair is a pure shift (I_L(x) = I_R(x + shift)) + noise: the pipeline finds it on most pixels
    assert (np.abs(inner - shift) <= 1).mean() > 0.9, (np.abs(inner - shift) <= 1).mean()


def test_c4_zncc11_sgm_wta_vfit_at_4096(eng, oracle, hooks):
    check_config(eng, oracle, hooks, 4096, 4096, 0, 256, "zncc11", 9)


def test_c5_census_cbca_sgm_wta_vfit_at_10000(eng, oracle, hooks):
    try:
        check_config(eng, oracle, hooks, 10000, 10000, -64, 64, "census+cbca", 7)
    except RuntimeError as err:
        if "memory" in str(err).lower():
            pytest.skip(f"two 51.6 GB volumes do not fit this device: {err}")
        raise


MULTISCALE = {"pipeline": {"matching_cost": {"matching_cost_method": "census", "window_size": 5},
                           "aggregation": {"aggregation_method": "cbca"},
                           "optimization": {"optimization_method": "sgm", "penalty": {"P1": 8, "P2": 32}},
                           "disparity": {"disparity_method": "wta", "invalid_disparity": "NaN"},
                           "refinement": {"refinement_method": "vfit"},
                           "multiscale": {"multiscale_method": "fixed_zoom_pyramid", "num_scales": 2, "scale_factor": 2, "marge": 3}}}


def test_c5_two_scale_run_on_a_strip():
    """configs[4]'s "multiscale x2": the coarse-to-fine loop of pandora.run (state_machine.py:521-556) with census + CBCA + SGM on
    a 2048 x 10000 strip: the fine scale searches +-marge around the coarse result and must land on the pair's shift."""
    import pandora_amd
    from pandora_amd.dataset import make_image
    from pandora_amd.state_machine import PandoraMachine

    H, W, shift = 2048, 10000, 7
    L, R = periodic_pair(H, W, shift, 99)
    left, right = make_image(L, disparity=[-64, 64]), make_image(R, disparity=[-64, 64])
    m = PandoraMachine()
    cfg = {"pipeline": m.check_conf({"pipeline": MULTISCALE["pipeline"]}, left, right)["pipeline"]}
    out, _ = pandora_amd.run(m, left, right, cfg)
    disp = np.asarray(out["disparity_map"].data)
    assert disp.shape == (H, W)
    inner = disp[8:-8, 80:-80]
    assert np.isfinite(inner).mean() > 0.98
    assert (np.abs(inner - shift) <= 1)[np.isfinite(inner)].mean() > 0.9


TILED = r'''
import os, sys
import numpy as np
sys.path.insert(0, %(root)r)
import pandora_amd
from pandora_amd import dist as pdist, runtime
from tests.transports import TcpComm
from pandora_amd.dataset import make_image
from pandora_amd.state_machine import PandoraMachine
sys.path.insert(0, os.path.join(%(root)r, "tests"))
from test_gpu_baseline_configs import periodic_pair
comm = TcpComm(runtime.get_engine())
H, W, shift = 2400, 3000, 7
L, R = periodic_pair(H, W, shift, 5)
rng = np.random.default_rng(3)
L += rng.integers(0, 3, L.shape).astype(np.float32)   # break the vertical period: every row is its own problem
CFG = {"pipeline": {"matching_cost": {"matching_cost_method": "census", "window_size": 5},
                    "aggregation": {"aggregation_method": "cbca"},
                    "optimization": {"optimization_method": "sgm", "penalty": {"P1": 8, "P2": 32}},
                    "disparity": {"disparity_method": "wta", "invalid_disparity": "NaN"},
                    "refinement": {"refinement_method": "vfit"}}}
left, right = make_image(L, disparity=[-64, 64]), make_image(R, disparity=[-64, 64])
tl, _ = pdist.run_row_tiled(left, right, CFG, comm=comm)   # margin = what the steps ask for: 40 (SGM) + 2 (census)
if comm.rank == 0:
    m = PandoraMachine()
    cfg = {"pipeline": m.check_conf({"pipeline": CFG["pipeline"]}, left, right)["pipeline"]}
    full, _ = pandora_amd.run(m, left, right, cfg)
    fd = np.asarray(full["disparity_map"].data)
    assert tl["disparity_map"].shape == (H, W)
    same = np.isclose(tl["disparity_map"], fd, equal_nan=True)
    print("TILED_SAME", same.mean(), same[:1100].mean(), same[1300:].mean())
    assert same.mean() > 0.995            # SGM paths are cut at the margin: a few pixels near the seam may move
    far = np.r_[0:1100, 1300:H]           # rows further than 100 from the seam at row 1200
    assert same[far].mean() > 0.9995
comm.barrier()
comm.close()
if comm.rank == 0:
    print("BIG_TILED_OK")
'''


def test_c5_row_tiles_of_1200_rows_over_two_ranks(tmp_path):
    """configs[4]'s "row-tiled": census + CBCA + SGM over two ranks, tiles of 1200 rows + the margin the steps ask for, against the
    untiled run (the reference's ROI convention: paths are cut at the margin, so a seam may differ on a few pixels)."""
    script = tmp_path / "bigtiles.py"
    script.write_text(TILED % {"root": ROOT})
    import bench

    port = str(bench._free_port_pair())  # (a fixed port collides with whatever else runs on the box)
    procs = []
    for rank in range(2):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PANDORA_AMD_DEVICE="0", RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT))
    outs = [p.communicate(timeout=900) for p in procs]
    assert "BIG_TILED_OK" in outs[0][0], "".join(o[0][-2000:] + o[1][-4000:] for o in outs)
