"""GPU: pandora_amd.dist.run_row_tiled / run_d_sharded with TWO ranks, both on the box's one GPU (RCCL refuses two ranks on one
device, so the exchange buffers travel through tests/transports.py's TcpComm stand-in; no PyTorch involved) against the untiled
run: a local pipeline must be identical everywhere; a census+SGM pipeline is cut at the 40-row margin like the reference's
own ROI tiling, so it may differ on a few pixels but must stay within the cones gate."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import os, sys
import numpy as np
sys.path.insert(0, %(root)r)
from PIL import Image
import pandora_amd
from pandora_amd import dist as pdist, runtime
from tests.transports import TcpComm
from pandora_amd.dataset import make_image
from pandora_amd.state_machine import PandoraMachine
comm = TcpComm(runtime.get_engine())
rank = comm.rank
cones = os.path.join(%(root)r, "tests", "golden", "cones")
L = np.array(Image.open(os.path.join(cones, "left.png"))).astype(np.float32)
R = np.array(Image.open(os.path.join(cones, "right.png"))).astype(np.float32)
gt = np.array(Image.open(os.path.join(cones, "disp_left.tif"))).astype(np.float32)
LOCAL = {"pipeline": {"matching_cost": {"matching_cost_method": "zncc", "window_size": 5, "subpix": 2},
                      "disparity": {"disparity_method": "wta", "invalid_disparity": "NaN"},
                      "refinement": {"refinement_method": "vfit"}}}
SGM = {"pipeline": {"matching_cost": {"matching_cost_method": "census", "window_size": 5},
                    "optimization": {"optimization_method": "sgm", "penalty": {"P1": 8, "P2": 32}},
                    "disparity": {"disparity_method": "wta", "invalid_disparity": "NaN"},
                    "refinement": {"refinement_method": "vfit"},
                    "validation": {"validation_method": "cross_checking_accurate"}}}
for name, cfg in (("local", LOCAL), ("sgm", SGM)):
    left, right = make_image(L, disparity=[-60, 0]), make_image(R, disparity=[0, 60])
    tl, tr = pdist.run_row_tiled(left, right, cfg, margin=40, comm=comm)
    if rank == 0:
        m = PandoraMachine()
        full_cfg = {"pipeline": m.check_conf({"pipeline": cfg["pipeline"]}, left, right)["pipeline"]}
        fl, fr = pandora_amd.run(m, left, right, full_cfg)
        assert tl["disparity_map"].shape == L.shape
        same = np.isclose(tl["disparity_map"], fl["disparity_map"].data, equal_nan=True)
        if name == "local":
            assert same.all() and np.array_equal(tl["validity_mask"], fl["validity_mask"].data) and tr is None
            assert np.array_equal(tl["interpolated_coeff"], fl["interpolated_coeff"].data, equal_nan=True)
        else:
            bad = (np.abs(np.nan_to_num(tl["disparity_map"], nan=1e4) + gt) > 1) & (gt != 0)
            assert bad.sum() / gt.size <= 0.20 and tr is not None and tr["disparity_map"].shape == L.shape
            print("SGM_TILED_SAME_FRACTION", same.mean())
            assert same.mean() > (0.97 if comm.world == 2 else 0.90)  # (every seam costs a few pixels)
comm.barrier()
comm.close()
if rank == 0:
    print("TILED_OK")
'''


def ranks(script, port, world=2):
    """All ranks on the box's one GPU, launched as plain processes with the launcher's environment variables."""
    procs = []
    for rank in range(world):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PANDORA_AMD_DEVICE="0", RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT))
    outs = [p.communicate(timeout=900) for p in procs]
    return outs[0][0], "".join(o[0][-2000:] + o[1][-4000:] for o in outs)


@pytest.mark.parametrize("world,port", [(2, 29541), (5, 29571)])
def test_row_tiled_pair_over_the_ranks(tmp_path, world, port):
    """(five ranks over cones' 375 rows: 75 owned rows each, the 40-row margins of a tile reach into both neighbours)"""
    script = tmp_path / "tiled.py"
    script.write_text(SCRIPT % {"root": ROOT})
    out, log = ranks(script, port, world)
    assert "TILED_OK" in out, log


DSHARD = r'''
import os, sys
import numpy as np
sys.path.insert(0, %(root)r)
from PIL import Image
import pandora_amd
from pandora_amd import dist as pdist, runtime
from tests.transports import TcpComm
from pandora_amd.dataset import make_image
from pandora_amd.state_machine import PandoraMachine
comm = TcpComm(runtime.get_engine())
rank = comm.rank
cones = os.path.join(%(root)r, "tests", "golden", "cones")
L = np.array(Image.open(os.path.join(cones, "left.png"))).astype(np.float32)[40:200, 30:330]
R = np.array(Image.open(os.path.join(cones, "right.png"))).astype(np.float32)[40:200, 30:330]
msk = np.zeros(L.shape, np.int16); msk[50:60, 100:130] = 2; msk[5, 7] = 1
CASES = [
    ({"matching_cost": {"matching_cost_method": "zncc", "window_size": 5, "subpix": 2},
      "disparity": {"disparity_method": "wta", "invalid_disparity": "NaN"}, "refinement": {"refinement_method": "vfit"}}, None),
    ({"matching_cost": {"matching_cost_method": "sad", "window_size": 3}, "aggregation": {"aggregation_method": "cbca"},
      "disparity": {"disparity_method": "wta", "invalid_disparity": -9999}, "refinement": {"refinement_method": "quadratic"}}, msk),
    ({"matching_cost": {"matching_cost_method": "census", "window_size": 5}, "disparity": {"disparity_method": "wta", "invalid_disparity": -9999}}, None),
]
for pipe, m in CASES:
    left, right = make_image(L, disparity=[-37, 3], msk=m), make_image(R, msk=m)
    got = pdist.run_d_sharded(left, right, {"pipeline": pipe}, comm)
    if rank == 0:
        left, right = make_image(L, disparity=[-37, 3], msk=m), make_image(R, msk=m)
        mach = PandoraMachine()
        cfg = {"pipeline": mach.check_conf({"pipeline": pipe}, left, right)["pipeline"]}
        full, _ = pandora_amd.run(mach, left, right, cfg)
        for key in got:
            np.testing.assert_array_equal(got[key], np.asarray(full[key].data), err_msg=key + " " + pipe["matching_cost"]["matching_cost_method"])
comm.barrier()
comm.close()
if rank == 0:
    print("DSHARD_OK")
'''


@pytest.mark.parametrize("world,port", [(2, 29543), (5, 29575)])
def test_d_sharded_pair_over_the_ranks(tmp_path, world, port):
    """pandora_amd.dist.run_d_sharded (costs sharded over D, one all_reduce(MIN) of packed keys, owner-rank refinement) with two
    and with five ranks (41 disparities: shards of 8 and 9) == the unsharded machine run, bit for bit: ZNCC sub-pixel + vfit,
    SAD + CBCA + quadratic with masks, census."""
    script = tmp_path / "dshard.py"
    script.write_text(DSHARD % {"root": ROOT})
    out, log = ranks(script, port, world)
    assert "DSHARD_OK" in out, log
