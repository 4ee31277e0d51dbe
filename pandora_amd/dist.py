"""Multi-GPU (one process per GPU, torch.distributed: backend "nccl" = RCCL over xGMI; "gloo" in CPU
tests).  SURVEY 8e:

* pair / row-tile sharding needs no collective (bench.py --gpus N);
* for pipelines WITHOUT SGM the cost volume shards over D exactly: every rank builds the costs of
  its disparity slice, reduces them to one packed (cost, global index) key per pixel
  (pmx_wta_minkey), a single all_reduce(MIN) of 8 B x H*W merges the shards and pmx_wta_from_keys
  decodes the winner - identical to np.argmin over the full volume, ties to the lowest index.
  SGM cannot shard over D: its recurrence needs min_k over all k at every pixel.
"""
import numpy as np

KEY_NONE = np.int64(0x7FFFFFFFFFFFFFFF)


def shard_range(n, world, rank):
    """Contiguous split of n items: [start, stop) of `rank` (first n % world ranks get one more)."""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def disparity_shard(dmin, dmax, subpix, world, rank, halo=0):
    """Disparity slice [d_lo, d_hi] (integers) owned by `rank`, optionally widened by `halo` integer
    disparities on each side (sub-pixel refinement reads the winner's neighbours)."""
    n = dmax - dmin + 1
    s, e = shard_range(n, world, rank)
    own = (dmin + s, dmin + e - 1)
    return own, (max(dmin, own[0] - halo), min(dmax, own[1] + halo))


def allreduce_min_keys(keys, group=None):
    """In-place MIN all-reduce of an int64 key tensor (any device)."""
    import torch.distributed as dist

    dist.all_reduce(keys, op=dist.ReduceOp.MIN, group=group)
    return keys


def decode_keys_numpy(keys, d0_global, subpix, invalid_disparity):
    """Host decode (tests / CPU tooling): int64 keys -> float32 disparity map + all-invalid mask."""
    keys = np.asarray(keys, np.int64)
    none = keys == KEY_NONE
    idx = (keys & 0x7FFFFFFF).astype(np.float64)
    disp = (d0_global + idx / subpix).astype(np.float32)
    disp[none] = invalid_disparity
    return disp, none


def sharded_wta(engine, cv_shard, is_max, index_offset, d0_global, subpix, invalid_disparity, group=None):
    """D-sharded winner-takes-all on the GPU.  `cv_shard` holds this rank's disparity slice whose
    first sample has GLOBAL index `index_offset`.  The merged disparity / validity end up in the
    engine exactly as after Engine.wta on the full volume."""
    import torch

    npix = engine.H * engine.W
    keys = torch.empty(npix, dtype=torch.int64, device=torch.device("cuda", engine.device))
    torch.cuda.synchronize(engine.device)
    engine.wta_minkey(cv_shard, is_max, index_offset, keys.data_ptr())
    engine.sync()  # the engine has its own HIP stream
    allreduce_min_keys(keys, group)
    torch.cuda.synchronize(engine.device)
    engine.wta_from_keys(keys.data_ptr(), d0_global, subpix, invalid_disparity)
    engine.sync()
    return keys


# ---- row tiles (the reference's own scaling convention: ROI tiles with a margin, img_tools.get_window /
# marge.py:86-101; the SGM plugin asks for 40 px, optimization/optimization.py:43) ---------------------------------
def row_tile(H, world, rank, margin=40):
    """Rows [own_lo, own_hi) owned by `rank` and the rows [read_lo, read_hi) it has to process so that every owned row
    sees `margin` rows of context on both sides (clipped at the image).  No data-path collective: every rank computes its
    tile from the images and keeps the owned rows."""
    lo, hi = shard_range(H, world, rank)
    return (lo, hi), (max(0, lo - margin), min(H, hi + margin))


def crop_tile(arr, H, world, rank, margin=40):
    """The owned rows of a per-pixel result computed on the rank's read window."""
    (lo, hi), (rlo, _) = row_tile(H, world, rank, margin)
    return arr[lo - rlo:hi - rlo]


def stitch_tiles(parts):
    """Concatenate the owned rows of all ranks (rank order) into the full map."""
    return np.concatenate(parts, axis=0)
