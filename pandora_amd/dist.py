"""One stereo pair over several GPUs (one process per GPU; the exchanges are RCCL collectives inside libpandora_amd.so,
csrc/pmx_comm.hip, bootstrapped by pandora_amd.comm.Comm - no PyTorch anywhere).  SURVEY 8e:

* row tiles with the steps' margin - the reference's own convention (optimization/optimization.py:43, marge.py:86-101) - for
  any pipeline, SGM included: no data-path collective, one all-gather of the owned rows of the 2-D results;
* for pipelines WITHOUT SGM the cost volume shards over D exactly: every rank builds the costs of its disparity slice, reduces
  them to one packed (cost, global index) key per pixel, a single all-reduce(MIN) of 8 B x H*W merges the shards and the decode
  is identical to np.argmin over the full volume, ties to the lowest index; the rank owning a pixel's winner refines it.
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


def decode_keys_numpy(keys, d0_global, subpix, invalid_disparity):
    """Host decode (tests / CPU tooling): int64 keys -> float32 disparity map + all-invalid mask."""
    keys = np.asarray(keys, np.int64)
    none = keys == KEY_NONE
    idx = (keys & 0x7FFFFFFF).astype(np.float64)
    disp = (d0_global + idx / subpix).astype(np.float32)
    disp[none] = invalid_disparity
    return disp, none


def sharded_wta(engine, comm, cv_shard, is_max, index_offset, d0_global, subpix, invalid_disparity):
    """D-sharded winner-takes-all.  `cv_shard` holds this rank's disparity slice whose first sample has GLOBAL index
    `index_offset`.  Keys, reduction and decode stay on the device (pmx_shard_minkey -> ncclAllReduce(min) ->
    pmx_shard_from_keys); the merged disparity / validity end up in the engine exactly as after Engine.wta on the full volume."""
    engine.shard_minkey(cv_shard, is_max, index_offset)
    comm.allreduce_xbuf("keys", "min")
    engine.shard_from_keys(d0_global, subpix, invalid_disparity)


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


def tile_dataset(ds, rlo, rhi):
    """Rows [rlo, rhi) of an image dataset (im, msk, disparity grids, segm / edges / classif layers), attrs shared."""
    from .dataset import DataArray, Dataset

    im = np.asarray(ds["im"].data)
    rows = slice(rlo, rhi)
    out = Dataset({"im": (ds["im"].dims, np.ascontiguousarray(im[..., rows, :]))},
                  coords={k: (np.asarray(v)[rows] if k == "row" else v) for k, v in ds.coords.items()}, attrs=dict(ds.attrs))
    if "msk" in ds.data_vars:
        out["msk"] = (("row", "col"), np.ascontiguousarray(np.asarray(ds["msk"].data)[rows]))
    if "disparity" in ds.data_vars:
        out["disparity"] = DataArray(np.ascontiguousarray(np.asarray(ds["disparity"].data)[:, rows]), ("band_disp", "row", "col"),
                                     {"band_disp": ["min", "max"]})
    for name in ("segm", "edges"):  # the layers an SGM geometric_prior may name (img_tools.add_layers): 2-D, sliced like the image
        if name in ds.data_vars:
            out[name] = (("row", "col"), np.ascontiguousarray(np.asarray(ds[name].data)[rows]))
    if "classif" in ds.data_vars:   # (band_classif, row, col)
        out["classif"] = DataArray(np.ascontiguousarray(np.asarray(ds["classif"].data)[:, rows]), ds["classif"].dims,
                                   {k: v for k, v in getattr(ds["classif"], "coords", {}).items()})
    return out


def _global_margin(img_left, img_right, cfg):
    """what the configured steps ask for (PandoraMachine.margins; reference: margins/margins.py:71-143)"""
    from .state_machine import PandoraMachine

    probe = PandoraMachine()
    probe.check_conf({"pipeline": cfg["pipeline"]}, img_left, img_right)
    g = probe.margins.global_margins
    return max(g.up, g.down)


def run_row_tiled(img_left, img_right, cfg, margin=None, comm=None):
    """One stereo pair over all ranks, the reference's way (ROI tiles with a margin, marge.py:86-101; 40 px is what the SGM plugin
    asks for, optimization/optimization.py:43): every rank runs the whole pipeline of ``cfg`` on its rows plus the margin (default:
    the global margins of the configured steps) on its own GPU and keeps the rows it owns; the owned rows of the 2-D results are
    all-gathered (RCCL, the only exchange) so that every rank holds the full maps.  Local pipelines are exact with a margin of at
    least the window radius; SGM paths are cut at the tile margin, as in the reference.  Returns (left, right) dicts of full-size
    arrays: disparity_map, validity_mask and, when present, interpolated_coeff.  comm=None: one rank."""
    from . import run as run_pipeline
    from . import runtime
    from .state_machine import PandoraMachine

    world, rank = (comm.world, comm.rank) if comm is not None else (1, 0)
    H = img_left.sizes["row"]
    if margin is None:
        margin = _global_margin(img_left, img_right, cfg)
    (lo, hi), (rlo, rhi) = row_tile(H, world, rank, margin)
    tile_l, tile_r = tile_dataset(img_left, rlo, rhi), tile_dataset(img_right, rlo, rhi)
    machine = PandoraMachine()
    tcfg = {"pipeline": machine.check_conf({"pipeline": cfg["pipeline"]}, tile_l, tile_r)["pipeline"]}
    out_l, out_r = run_pipeline(machine, tile_l, tile_r, tcfg)
    eng = comm.engine if comm is not None and comm.engine is not None else runtime.get_engine()

    def gathered(ds):
        if len(ds.sizes) == 0:
            return None
        keys = [k for k in ("disparity_map", "validity_mask", "interpolated_coeff") if k in ds.data_vars]
        own = {k: np.ascontiguousarray(np.asarray(ds[k].data)[lo - rlo:hi - rlo]) for k in keys}
        if world == 1:
            return own
        # the machine's final maps live on the host (filters, validation): the owned rows go up into the engine's full-size
        # maps, one all-gather over xGMI, and the full maps come down
        itp = own.get("interpolated_coeff")
        eng.set_full_rows(H, lo, hi, own["disparity_map"], own["validity_mask"], itp)
        comm.allgather_rows(H, itp is not None)
        full = eng.get_full_maps(H, want_itp=itp is not None)
        return dict(zip(keys, full))

    return gathered(out_l), gathered(out_r)


def run_d_sharded(img_left, img_right, cfg, comm):
    """One stereo pair, the cost volume sharded over D across the ranks (pipelines WITHOUT optimization): every rank
    builds and aggregates the costs of its disparity slice (one integer disparity of halo on each side), the slices meet in ONE
    all-reduce(MIN) of a packed (cost, global index) key per pixel (ncclAllReduce over xGMI), and the rank that owns a pixel's
    winner refines it (one all-reduce(SUM) of value-or-zero maps: exact).  Keys, winner maps and refinement packs never leave the
    device; the one host hop is the validity mask, which the host computes as in the single-GPU flow (criteria.py) from the
    reduced all-NaN flags.  The maps equal the unsharded run bit for bit.  Supported steps: matching_cost, aggregation,
    disparity, refinement; uniform disparity ranges.  Returns {"disparity_map", "validity_mask"[, "interpolated_coeff"]}."""
    from . import criteria, matching_cost, runtime
    from .dataset import make_image
    from .state_machine import PandoraMachine

    pipe = cfg["pipeline"]
    extra = [k for k in pipe if k.split(".")[0] not in ("matching_cost", "aggregation", "disparity", "refinement")]
    if extra or "disparity" not in pipe:
        raise NotImplementedError(f"run_d_sharded handles matching_cost / aggregation / disparity / refinement only (got {extra})")
    world, rank = comm.world, comm.rank
    grids = np.asarray(img_left["disparity"].data)
    dmin, dmax = int(grids[0].min()), int(grids[1].max())
    if grids[0].max() != dmin or grids[1].min() != dmax:
        raise NotImplementedError("run_d_sharded needs one disparity range for the whole image")
    if dmax - dmin + 1 < world:
        raise ValueError("fewer disparities than ranks")
    (olo, ohi), (wlo, whi) = disparity_shard(dmin, dmax, 1, world, rank, halo=1)
    left_w = make_image(np.asarray(img_left["im"].data), disparity=[wlo, whi], msk=img_left["msk"].data if "msk" in img_left.data_vars else None,
                        valid_pixels=img_left.attrs.get("valid_pixels", 0), no_data_mask=img_left.attrs.get("no_data_mask", 1),
                        band_names=list(img_left.coords["band_im"]) if "band_im" in img_left.coords else None)
    machine = PandoraMachine()
    head = {"pipeline": {k: pipe[k] for k in pipe if k.split(".")[0] in ("matching_cost", "aggregation")}}
    head["pipeline"]["disparity"] = pipe["disparity"]  # (checked for the sequencing, not run here)
    checked = machine.check_conf(head, left_w, img_right)["pipeline"]
    machine.run_prepare({"pipeline": checked}, left_w, img_right)
    for step in checked:
        if step.split(".")[0] != "disparity":
            machine.run(step, {"pipeline": checked})
    cv = machine.left_cv
    dcv = cv["cost_volume"].device_cv
    eng = dcv.engine
    if comm.engine is not eng:
        raise ValueError("the communicator belongs to another engine than the one the machine runs on")
    subpix = int(cv.attrs["subpixel"])
    is_max = cv.attrs["type_measure"] == "max"
    # ---- the validity mask of the WHOLE range (criteria.py:66-158, :291-353), with the all-NaN pixels of the whole volume
    mc = matching_cost.AbstractMatchingCost(**{k: v for k, v in checked["matching_cost"].items()})
    grid = mc.allocate_cost_volume(img_left, (img_left["disparity"].sel(band_disp="min"), img_left["disparity"].sel(band_disp="max")))
    grid = criteria.validity_mask(img_left, img_right, grid)
    eng.shard_nan_pixels(dcv)
    comm.allreduce_xbuf("nanpix", "min")  # NaN for every disparity of every shard
    missing = eng.xbuf_download("nanpix").reshape(eng.H, eng.W).astype(bool)
    criteria.mask_invalid_variable_disparity_range(grid, missing)
    if grid.attrs["offset_row_col"] > 0:
        criteria.mask_border(grid)
    eng.set_validity(np.asarray(grid["validity_mask"].data, np.int64))
    # ---- winner-takes-all over the shards: one all-reduce of 8 bytes per pixel
    invalid = pipe["disparity"].get("invalid_disparity", -9999)
    invalid = float("nan") if isinstance(invalid, str) else float(invalid)
    sharded_wta(eng, comm, dcv, is_max, (wlo - dmin) * subpix, dmin, subpix, invalid)
    if "refinement" in pipe:
        eng.shard_refine_pack(dcv, pipe["refinement"]["refinement_method"], is_max, olo, ohi, rank == world - 1)
        comm.allreduce_xbuf("refine_pack", "sum")   # exactly one owner per valid pixel: value + zeros is exact
        comm.allreduce_xbuf("refine_flags", "sum")
        eng.shard_refine_unpack()
        disp, val, itp = eng.get_disparity(want_itp=True)
        out = {"disparity_map": disp, "validity_mask": val, "interpolated_coeff": itp}
    else:
        disp, val = eng.get_disparity()
        out = {"disparity_map": disp, "validity_mask": val}
    runtime.invalidate(eng.device)
    return out
