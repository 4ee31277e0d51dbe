"""GPU: the D-sharded WTA through pandora_amd.dist.sharded_wta with a real RCCL process group
(1 rank - the GPU box has one GPU; the merge itself is covered at world_size 2 by
tests/test_dist_cpu.py).  Runs in a subprocess because torch (which bundles its own HIP runtime)
must be imported BEFORE libpandora_amd.so in a process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import os, sys
import numpy as np
import torch, torch.distributed as dist          # torch first
sys.path.insert(0, %(root)r)
from pandora_amd.engine import Engine
from pandora_amd import dist as pdist
from oracle import capi
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
rng = np.random.default_rng(12)
H, W, dmin, dmax = 17, 29, -9, 6
D = dmax - dmin + 1
cv = rng.integers(0, 4, (H, W, D)).astype(np.float32) - 1.0
cv[rng.random(cv.shape) < 0.2] = np.nan
cv[3, 4] = np.nan
eng = Engine(0)
z = np.zeros((H, W), np.float32)
eng.set_images(z, z, 1)
for is_max in (False, True):
    merged = None
    for rank in range(2):                         # two disparity shards, reduced on this GPU
        (lo, hi), _ = pdist.disparity_shard(dmin, dmax, 1, 2, rank)
        shard = eng.alloc_cv(hi - lo + 1, lo)
        shard.from_host(np.ascontiguousarray(cv[:, :, lo - dmin:hi - dmin + 1]))
        keys = torch.empty(H * W, dtype=torch.int64, device="cuda:0")
        torch.cuda.synchronize()
        eng.wta_minkey(shard, is_max, lo - dmin, keys.data_ptr())
        eng.sync()
        merged = keys if merged is None else torch.minimum(merged, keys)
    pdist.allreduce_min_keys(merged)              # RCCL all_reduce(MIN), world 1
    torch.cuda.synchronize()
    eng.set_validity(None)
    eng.wta_from_keys(merged.data_ptr(), dmin, 1, -9999.0)
    disp, val = eng.get_disparity()
    edisp, eval_ = capi.wta(cv, dmin, 1, is_max, -9999.0)
    np.testing.assert_array_equal(disp, edisp)
    np.testing.assert_array_equal(val, eval_)
    # and the one-call helper on the full volume
    full = eng.alloc_cv(D, dmin)
    full.from_host(cv)
    eng.set_validity(None)
    pdist.sharded_wta(eng, full, is_max, 0, dmin, 1, -9999.0)
    disp2, val2 = eng.get_disparity()
    np.testing.assert_array_equal(disp2, edisp)
dist.destroy_process_group()
print("DIST_OK")
'''


def test_sharded_wta_with_rccl():
    pytest.importorskip("torch")
    out = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], capture_output=True, text=True, timeout=600)
    assert "DIST_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
