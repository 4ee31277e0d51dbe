"""GPU: the exchange steps of csrc/pmx_comm.hip through a real RCCL communicator.  The GPU box has one GPU, so the communicator
has one rank (the collectives are identities but every call goes through librccl.so on the engine's stream); merges over two
ranks are covered by tests/test_gpu_tiled.py (two ranks on the one GPU, test transport) and tests/test_dist_cpu.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def world1():
    from pandora_amd.comm import Comm
    from pandora_amd.engine import Engine

    eng = Engine(0)
    comm = Comm(eng, rank=0, world=1, always=True)
    yield eng, comm
    comm.close()
    eng.close()


@pytest.mark.parametrize("is_max", [False, True])
def test_sharded_wta_and_refinement_through_rccl(world1, oracle, is_max):
    from pandora_amd import dist as pdist

    eng, comm = world1
    rng = np.random.default_rng(12)
    H, W, dmin, dmax = 17, 29, -9, 6
    D = dmax - dmin + 1
    cv = rng.integers(0, 4, (H, W, D)).astype(np.float32) - 1.0
    cv[rng.random(cv.shape) < 0.2] = np.nan
    cv[3, 4] = np.nan
    z = np.zeros((H, W), np.float32)
    eng.set_images(z, z, 1)
    full = eng.alloc_cv(D, dmin)
    full.from_host(cv)
    # all-NaN flags: shard -> exchange buffer -> ncclAllReduce(min) -> host
    eng.shard_nan_pixels(full)
    comm.allreduce_xbuf("nanpix", "min")
    np.testing.assert_array_equal(eng.xbuf_download("nanpix").reshape(H, W).astype(bool), np.isnan(cv).all(axis=2))
    # keys -> ncclAllReduce(min, uint64) -> decode
    eng.set_validity(None)
    pdist.sharded_wta(eng, comm, full, is_max, 0, dmin, 1, -9999.0)
    disp, val = eng.get_disparity()
    edisp, eval_ = oracle.wta(cv, dmin, 1, is_max, -9999.0)
    np.testing.assert_array_equal(disp, edisp)
    np.testing.assert_array_equal(val, eval_)
    # owner refinement: pack -> ncclAllReduce(sum) x2 -> unpack; one rank owns every winner
    eng.shard_refine_pack(full, "vfit", is_max, dmin, dmax, True)
    comm.allreduce_xbuf("refine_pack", "sum")
    comm.allreduce_xbuf("refine_flags", "sum")
    eng.shard_refine_unpack()
    rdisp, rval, ritp = eng.get_disparity(want_itp=True)
    oitp, odisp, oval = oracle.refine(cv, edisp, eval_, dmin, dmax, 1, is_max, "vfit")
    np.testing.assert_array_equal(rdisp, odisp)
    np.testing.assert_array_equal(rval, oval)
    np.testing.assert_array_equal(ritp, oitp)
    full.free()


@pytest.mark.parametrize("H", [64, 61])
def test_row_tile_gather_through_rccl(world1, H):
    """pmx_tile_place + ncclAllGather (in place when the rows divide evenly, staged otherwise) + pmx_get_full_maps."""
    eng, comm = world1
    rng = np.random.default_rng(H)
    W, margin = 40, 7
    tile = rng.random((H + margin, W)).astype(np.float32)
    eng.set_images(tile, tile, 1)
    val = rng.integers(0, 4096, (H + margin, W)).astype(np.int64)
    eng.set_disparity(tile * 3, val)
    cv = eng.alloc_cv(3, 0)
    cv.from_host(rng.random((H + margin, W, 3)).astype(np.float32))
    eng.refine(cv, "vfit", False)  # fills the coefficient map
    d, v, t = eng.get_disparity(want_itp=True)
    eng.tile_place(H, 0, H, 0, True)   # the tile starts at image row 0 and owns rows [0, H)
    comm.allgather_rows(H, True)
    fd, fv, ft = eng.get_full_maps(H, want_itp=True)
    np.testing.assert_array_equal(fd, d[:H])
    np.testing.assert_array_equal(fv, v[:H])
    np.testing.assert_array_equal(ft, t[:H])
    # the same through the gather to one rank (ncclSend / ncclRecv group; with one rank: the validity narrowing / widening only)
    eng.tile_place(H, 0, H, 0, True)
    comm.gather_rows(H, True, root=0)
    fd, fv, ft = eng.get_full_maps(H, want_itp=True)
    np.testing.assert_array_equal(fd, d[:H])
    np.testing.assert_array_equal(fv, v[:H])
    # a tile that starts above its owned rows
    eng.tile_place(H + margin - 5, 5, H + margin - 5, 0, False)
    fd, fv = eng.get_full_maps(H + margin - 5)
    np.testing.assert_array_equal(fd[5:], d[5:H + margin - 5])
    cv.free()


def test_host_scalars_through_rccl(world1):
    _, comm = world1
    out = comm.engine.comm_allreduce_scalars(np.arange(8, dtype=np.float64), "max")
    np.testing.assert_array_equal(out, np.arange(8.0))


def test_gathers_on_their_own_stream_keep_their_order(world1):
    """pmx_comm_gather_rows queues its work on the communication stream, between two events: a loop of steps as bench.py runs them
    (new maps, pmx_tile_place, gather - nothing synchronised in between) ends with the LAST step's maps in the full-size buffers,
    and a collective on the context's own stream afterwards still works (it joins the communication stream first)."""
    eng, comm = world1
    rng = np.random.default_rng(3)
    H, W = 48, 40
    img = rng.random((H, W)).astype(np.float32)
    eng.set_images(img, img, 1)
    last = None
    for step in range(6):
        d = rng.random((H, W)).astype(np.float32) + step
        v = rng.integers(0, 4096, (H, W)).astype(np.int64)
        eng.set_disparity(d, v)
        eng.tile_place(H, 0, H, 0, False)
        comm.gather_rows(H, False, root=0)
        last = (d, v)
    out = comm.engine.comm_allreduce_scalars(np.arange(8, dtype=np.float64), "sum")
    np.testing.assert_array_equal(out, np.arange(8.0))
    fd, fv = eng.get_full_maps(H)
    np.testing.assert_array_equal(fd, last[0])
    np.testing.assert_array_equal(fv, last[1])
    eng.sync()
