"""world_size-2 tests (CPU) of the multi-GPU exchange logic.  The product exchanges are RCCL collectives inside the library; here
the same merge arithmetic runs through the two host stand-ins of tests/transports.py: GlooComm (torch.distributed on CPU) and TcpComm
(the launcher-independent socket rendezvous that also hands out the RCCL id).  D-sharded WTA: two processes each reduce their
disparity slice to packed keys (numpy restatement of the kernel's packing, test-only), one all-reduce(MIN) merges them, and the
decode equals np.argmin over the full volume (first minimum on ties, NaN = +inf, all-NaN -> invalid)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pack_keys_numpy(cv, is_max, index_offset):
    """mirror of minkey_kernel (pandora_amd/csrc/k_disparity.hip)"""
    v = np.where(np.isnan(cv), -np.inf if is_max else np.inf, cv).astype(np.float32)
    idx = (np.argmax(v, axis=2) if is_max else np.argmin(v, axis=2))
    best = np.take_along_axis(v, idx[..., None], axis=2)[..., 0]
    f = (-best if is_max else best).astype(np.float32)
    f = np.where(f == 0, np.float32(0), f)
    u = f.view(np.uint32).astype(np.uint64)
    order = np.where(u & 0x80000000, (~u) & 0xFFFFFFFF, u | 0x80000000).astype(np.uint64)
    keys = ((order << np.uint64(31)) | (idx + index_offset).astype(np.uint64)).astype(np.int64)
    keys[np.all(np.isnan(cv), axis=2)] = np.int64(0x7FFFFFFFFFFFFFFF)
    return keys


def _worker(rank, world, port, is_max, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from pandora_amd import dist as pdist
    from tests.transports import GlooComm

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = GlooComm()
    rng = np.random.default_rng(0)  # same volume on both ranks
    H, W, dmin, dmax = 9, 13, -7, 5
    D = dmax - dmin + 1
    cv = rng.integers(0, 5, (H, W, D)).astype(np.float32)
    cv[rng.random(cv.shape) < 0.2] = np.nan
    cv[0, 0] = np.nan
    (lo, hi), _ = pdist.disparity_shard(dmin, dmax, 1, world, rank)
    shard = cv[:, :, lo - dmin:hi - dmin + 1]
    keys = comm.host_allreduce(pack_keys_numpy(shard, is_max, lo - dmin).ravel().copy(), "min")
    disp, none = pdist.decode_keys_numpy(keys.reshape(H, W), dmin, 1, -9999.0)
    if rank == 0:
        q.put((disp, none, cv))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("is_max", [False, True])
def test_d_sharded_wta_merge_world2(oracle, is_max):
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, is_max, q)) for r in range(2)]
    for p in procs:
        p.start()
    disp, none, cv = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    edisp, _ = oracle.wta(cv, -7, 1, is_max, -9999.0)
    np.testing.assert_array_equal(disp, edisp)
    assert none[0, 0] and none.sum() == np.all(np.isnan(cv), axis=2).sum()


def test_shard_ranges_cover_everything():
    from pandora_amd import dist as pdist

    for n in (1, 7, 129, 257):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                s, e = pdist.shard_range(n, world, r)
                cover += list(range(s, e))
            assert cover == list(range(n))
    own, halo = pdist.disparity_shard(-60, 0, 1, 8, 3, halo=1)
    assert halo[0] == own[0] - 1 and halo[1] == own[1] + 1


def test_row_tiles_cover_the_image_with_margins():
    """Row tiling (BASELINE configs[4]): owned rows partition the image, read windows add the margin and stay
    inside the image; cropping every rank's window and stitching gives back the full map."""
    from pandora_amd import dist as pd

    H, W = 1003, 7
    full = np.arange(H * W, dtype=np.float32).reshape(H, W)
    for world in (1, 2, 8):
        owned, parts = [], []
        for rank in range(world):
            (lo, hi), (rlo, rhi) = pd.row_tile(H, world, rank, margin=40)
            assert 0 <= rlo <= lo < hi <= rhi <= H
            assert lo - rlo == min(40, lo) and rhi - hi == min(40, H - hi)
            owned.append((lo, hi))
            parts.append(pd.crop_tile(full[rlo:rhi], H, world, rank, margin=40))
        assert owned[0][0] == 0 and owned[-1][1] == H and all(a[1] == b[0] for a, b in zip(owned, owned[1:]))
        np.testing.assert_array_equal(pd.stitch_tiles(parts), full)


def _tcp_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from pandora_amd import dist as pdist
    from tests.transports import TcpComm

    comm = TcpComm(None, rank=rank, world=world, addr="127.0.0.1", port=port)
    uid = comm.rdv.broadcast(bytes(range(128)) if rank == 0 else None)  # how the RCCL id travels
    rng = np.random.default_rng(5)
    H, W, dmin, dmax = 11, 17, -20, 12
    D = dmax - dmin + 1
    cv = rng.integers(0, 6, (H, W, D)).astype(np.float32)
    cv[rng.random(cv.shape) < 0.3] = np.nan
    (lo, hi), _ = pdist.disparity_shard(dmin, dmax, 1, world, rank)
    keys = comm.host_allreduce(pack_keys_numpy(cv[:, :, lo - dmin:hi - dmin + 1], False, lo - dmin), "min")
    disp, _ = pdist.decode_keys_numpy(keys, dmin, 1, -9999.0)
    tmax = comm.host_allreduce(np.array([float(rank + 1)]), "max")
    rows = comm.rdv.allgather(bytes([rank]) * (rank + 2))
    comm.barrier()
    if rank == world - 1:
        q.put((uid, disp, cv, float(tmax[0]), rows))
    comm.close()


def test_tcp_rendezvous_world3(oracle):
    """Three processes, no torch: the socket star broadcasts rank 0's bytes, reduces and gathers host arrays."""
    import multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_tcp_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    uid, disp, cv, tmax, rows = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert uid == bytes(range(128)) and tmax == 3.0 and rows == [b"\x00" * 2, b"\x01" * 3, b"\x02" * 4]
    edisp, _ = oracle.wta(cv, -20, 1, False, -9999.0)
    np.testing.assert_array_equal(disp, edisp)


def _rdv_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from pandora_amd.comm import Rendezvous

    rdv = Rendezvous(rank, world, "127.0.0.1", port, timeout=60.0)
    got = rdv.broadcast(b"the id" if rank == 0 else None)
    rdv.barrier()
    q.put((rank, got))
    rdv.close()


def test_rendezvous_steps_past_a_port_somebody_else_holds():
    """The port one above the launcher's may belong to another service: rank 0 listens at the next free candidate, the other
    ranks recognise this run's rank 0 by its greeting (a foreign listener that accepts and stays silent - or talks - is skipped)."""
    import multiprocessing as mp
    import threading

    foreign = socket.socket()
    foreign.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    foreign.bind(("127.0.0.1", 0))
    port = foreign.getsockname()[1]
    foreign.listen(8)
    stop = threading.Event()

    def babble():  # accepts, says something that is not the greeting, hangs up
        foreign.settimeout(0.2)
        while not stop.is_set():
            try:
                c, _ = foreign.accept()
                c.sendall(b"HTTP/1.1 400 Bad Request\r\n\r\n")
                c.close()
            except OSError:
                pass

    t = threading.Thread(target=babble, daemon=True)
    t.start()
    try:
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_rdv_worker, args=(r, 3, port, q)) for r in range(3)]
        for p in procs:
            p.start()
        got = sorted(q.get(timeout=120) for _ in range(3))
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        assert got == [(0, b"the id"), (1, b"the id"), (2, b"the id")]
    finally:
        stop.set()
        t.join()
        foreign.close()


def test_row_tiles_keep_the_layers_a_geometric_prior_names():
    """dist.tile_dataset slices segm / edges like the image and classif along its rows, band names kept: an SGM geometric_prior
    (plugin_libsgm.rst:49-78) on a row-tiled run finds its source layer in every tile."""
    from pandora_amd import dist as pd
    from pandora_amd.dataset import DataArray, make_image

    H, W = 23, 11
    rng = np.random.default_rng(1)
    ds = make_image(rng.random((H, W)).astype(np.float32), disparity=[-3, 3], msk=rng.integers(0, 2, (H, W)).astype(np.int16))
    ds["segm"] = (("row", "col"), rng.integers(0, 4, (H, W)).astype(np.int16))
    ds["edges"] = (("row", "col"), rng.integers(0, 2, (H, W)).astype(np.int16))
    ds.coords["band_classif"] = np.array(["water", "forest"], dtype=object)
    ds["classif"] = DataArray(rng.integers(0, 2, (2, H, W)).astype(np.int16), ("band_classif", "row", "col"))
    for world in (2, 3):
        for rank in range(world):
            (lo, hi), (rlo, rhi) = pd.row_tile(H, world, rank, margin=4)
            t = pd.tile_dataset(ds, rlo, rhi)
            for name in ("im", "msk", "segm", "edges"):
                np.testing.assert_array_equal(np.asarray(t[name].data), np.asarray(ds[name].data)[rlo:rhi])
            np.testing.assert_array_equal(np.asarray(t["classif"].data), np.asarray(ds["classif"].data)[:, rlo:rhi])
            np.testing.assert_array_equal(np.asarray(t["disparity"].data), np.asarray(ds["disparity"].data)[:, rlo:rhi])
            assert list(t.coords["band_classif"]) == ["water", "forest"] and t["classif"].dims == ("band_classif", "row", "col")
            assert t.sizes["row"] == rhi - rlo
