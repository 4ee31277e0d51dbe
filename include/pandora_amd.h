/* pandora_amd.h - C ABI of the MI355X-native stereo cost-volume engine (libpandora_amd.so).
 *
 * This is the drop-in boundary for ONE hot path of CNES/Pandora:
 *     matching_cost -> aggregation -> optimization -> disparity -> refinement
 * Every entry point names the reference interface it replaces (paths relative to
 * /root/reference/src/pandora).  Plain pointers and sizes only; no exceptions cross the ABI:
 * functions return 0 on success and a negative code on error (text via pmx_last_error()).
 *
 * Ownership: the caller owns every host buffer; the library owns all device memory behind the
 * opaque handles.  Host buffers are C-contiguous.  Images are float32 [H][W]; a cost volume is
 * float32 [H][W][D] (disparity index innermost, exactly the reference's xarray layout,
 * matching_cost/matching_cost.py:394-397); disparity index k means disparity d0 + k/subpix.
 * Threading: a context is single-threaded (one HIP stream); use one context per thread/GPU.
 */
#ifndef PANDORA_AMD_H
#define PANDORA_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pmx_ctx pmx_ctx; /* one GPU, one stream, the resident stereo pair + scratch */
typedef struct pmx_cv pmx_cv;   /* a device-resident cost volume */

enum { PMX_OK = 0, PMX_ERR_ARG = -1, PMX_ERR_HIP = -2, PMX_ERR_STATE = -3, PMX_ERR_UNSUPPORTED = -4 };
enum { PMX_REFINE_VFIT = 0, PMX_REFINE_QUADRATIC = 1 };

/* stage ids for pmx_stage_time() */
enum {
    PMX_STAGE_CENSUS_TRANSFORM = 0, PMX_STAGE_CENSUS_COST = 1, PMX_STAGE_SAD_SSD = 2, PMX_STAGE_ZNCC = 3,
    PMX_STAGE_MASK = 4, PMX_STAGE_CBCA_ARMS = 5, PMX_STAGE_CBCA_H = 6, PMX_STAGE_CBCA_V = 7,
    PMX_STAGE_SGM_PATH = 8, PMX_STAGE_SGM_FINAL = 9, PMX_STAGE_WTA = 10, PMX_STAGE_REFINE = 11,
    PMX_STAGE_REVERSE = 12, PMX_STAGE_MINKEY = 13, PMX_STAGE_SGM_FUSED = 14, PMX_STAGE_SGM_FAMILY = 15, PMX_STAGE_COLLECTIVE = 16,
    PMX_STAGE_SGM_SPAN = 17, /* the whole SGM step when its kernels run side by side on two streams (fork -> join on the context's stream) */
    PMX_STAGE_COUNT = 24
};

const char* pmx_last_error(void);
int pmx_device_count(void);

/* ---- context / data residency --------------------------------------------------------------- */
pmx_ctx* pmx_create(int device);
void pmx_destroy(pmx_ctx* ctx);
int pmx_sync(pmx_ctx* ctx);

/* Upload the stereo pair.  Also builds the subpix-1 shifted right images on the device
 * (img_tools.py:713-752 shift_right_img, linear interpolation in double).  Replaces the numpy
 * image hand-off of every compute_cost_volume (e.g. matching_cost/census.py:113-133). */
int pmx_set_images(pmx_ctx* ctx, const float* left, const float* right, int H, int W, int subpix);
/* The resident pair with left and right exchanged (masks too; the sub-pixel right images are rebuilt by linear interpolation):
 * what compute_cost_volume(img_right, img_left, ...) of the validation step's right-side volume needs (state_machine.py:311-331),
 * without sending the same two images again. */
int pmx_swap_images(pmx_ctx* ctx);
/* pmx_set_images that also returns pmx_host_fingerprint of the two images, taken in the pass that copies them to the
 * staging buffer (a caller that keeps track of which pair is resident reads every image once instead of twice). */
int pmx_set_images_fingerprinted(pmx_ctx* ctx, const float* left, const float* right, int H, int W, int subpix,
                                 uint64_t* fp_left, uint64_t* fp_right);

/* Replace the k-th shifted right image (k = 1 .. subpix-1, float32 [H][W-1]) that pmx_set_images built with linear interpolation:
 * matching_cost's "spline_order" 2..5 resamples with scipy.ndimage.zoom(order=...) on the host (img_tools.py:713-752; 2-D work) and
 * hands the result over. */
int pmx_set_shifted_right(pmx_ctx* ctx, int k, const float* shifted);

/* Optional int16 masks (NULL = all valid), convention of img attrs valid_pixels / no_data_mask.
 * Feeds the masks_dilatation predicate of cv_masked (matching_cost/matching_cost.py:484-602) and
 * the CBCA pre-masking (aggregation/cbca.py:217-262). */
int pmx_set_masks(pmx_ctx* ctx, const int16_t* msk_left, const int16_t* msk_right, int valid_value, int nodata_value);

/* Optional per-pixel disparity grids, double [H][W] (NULL,NULL = none):
 * matching_cost/matching_cost.py:845-860. */
int pmx_set_disparity_grids(pmx_ctx* ctx, const double* disp_min, const double* disp_max);

/* Placement-aware allocation (no reference counterpart: the reference works in host memory).  For every NEW buffer of 256 MB
 * or more the context allocates up to `trials` candidates (never holding more than half of the free memory), times one streaming fill
 * and read of each and keeps the fastest: on MI355X the bandwidth of a hipMalloc'd buffer depends on where the driver placed it
 * (DESIGN.md 4).  One-time cost of a few hundred ms per buffer size; cached buffers are reused as they are.  The default is 6
 * (round 6: what every caller of the library gets - until round 5 it was an opt-in that only bench.py used); trials = 1 switches
 * it off (plain hipMalloc). */
int pmx_set_placement_trials(pmx_ctx* ctx, int trials);
/* What plain streaming kernels reach on this device, in GB/s (no reference counterpart; SURVEY 8d: "measure achievable with a device
 * memcpy/triad on the box and report both"): a 16-byte-per-lane fill, read and copy of `bytes` (two buffers of that size are allocated
 * and freed), best of three passes each; copy counts read + written bytes.  bench.py's roofline.peak_measured. */
int pmx_measure_hbm(pmx_ctx* ctx, size_t bytes, double* read_gbs, double* write_gbs, double* copy_gbs);
/* Gives the device memory a context keeps between calls back to the driver (no reference counterpart: the reference's buffers are
 * numpy arrays that die with their last reference): the cache of freed volumes, the SGM accumulator volume and the hand-off buffer
 * of the marching kernels - after a 16384 x 16384 x 65 float32 run that is ~100 GB.  Everything comes back on demand; cost volume
 * handles, the resident pair and result maps are untouched.  free_bytes / total_bytes (either may be NULL): the device's memory
 * after the release (hipMemGetInfo). */
int pmx_release_caches(pmx_ctx* ctx, size_t* free_bytes, size_t* total_bytes);
/* Kernel-route and tuning options of a context (no reference counterpart; the list with meanings: DESIGN.md 7b).  By default the
 * library chooses every kernel from the call's arguments alone.  An option forces a choice - what the parity tests use to drive every
 * route on small inputs, and the A/B scripts under tools/.  `name` is one of the known names ("SGM8_FAM", "SGM_SCHED", "CBCA_FAST",
 * ...), `value` a short string ("0", "1", "fam", "16x20"); value = NULL clears the option.  pmx_create seeds the options ONCE
 * from the environment variables PMX_<name>; after that the environment is never read again.  Unknown names: PMX_ERR_ARG.
 * pmx_get_option returns the value or NULL (the pointer is valid until the option changes). */
int pmx_set_option(pmx_ctx* ctx, const char* name, const char* value);
const char* pmx_get_option(const pmx_ctx* ctx, const char* name);
const char* pmx_option_name(int index); /* the known names, 0, 1, ...; NULL past the last */
/* Lazy evaluation (default ON).  A cost volume handle may hold the volume in a cheaper exact form
 * than float32 [H][W][D] - "all NaN", "census codes, costs not yet written", "eight uint8 SGM path
 * volumes" - and only materialises float32 when a step or the caller needs it.  With census costs
 * and integer penalties this lets pmx_census -> pmx_sgm -> pmx_wta -> pmx_refine run without ever
 * writing a float volume; results are bit-identical.  pmx_set_lazy(ctx, 0) forces the float path. */
int pmx_set_lazy(pmx_ctx* ctx, int enabled);

/* ---- cost volume handles -------------------------------------------------------------------- */
/* allocate_cost_volume (matching_cost/matching_cost.py:377-407): NaN-filled [H][W][D] on device */
pmx_cv* pmx_cv_alloc(pmx_ctx* ctx, int D, int d0);
void pmx_cv_free(pmx_ctx* ctx, pmx_cv* cv);
int pmx_cv_fill_nan(pmx_ctx* ctx, pmx_cv* cv);
int pmx_cv_upload(pmx_ctx* ctx, pmx_cv* cv, const float* host);   /* cv["cost_volume"].data = ... */
int pmx_cv_download(pmx_ctx* ctx, pmx_cv* cv, float* host);       /* ... = cv["cost_volume"].data */
/* rows [row_lo, row_hi) of the volume only (cv["cost_volume"].data[row_lo:row_hi]): a 17 GB volume need not cross PCIe whole */
int pmx_cv_download_rows(pmx_ctx* ctx, pmx_cv* cv, int row_lo, int row_hi, float* host);
int pmx_cv_dims(const pmx_cv* cv, int* H, int* W, int* D, int* d0, int* subpix);

/* ---- matching cost -------------------------------------------------------------------------- */
/* matching_cost_cpp.compute_matching_costs (matching_cost/cpp/src/census.cpp:97-180) driven by
 * Census.compute_cost_volume (matching_cost/census.py:74-153).  win in {3,5,7,9,11,13}. */
int pmx_census(pmx_ctx* ctx, pmx_cv* cv, int win);
/* SadSsd.compute_cost_volume (matching_cost/sad_ssd.py:75-207); squared=1 -> ssd */
int pmx_sad_ssd(pmx_ctx* ctx, pmx_cv* cv, int win, int squared);
/* Zncc.compute_cost_volume (matching_cost/zncc.py:114-277) */
int pmx_zncc(pmx_ctx* ctx, pmx_cv* cv, int win);
/* AbstractMatchingCost.cv_masked NaN injection (matching_cost/matching_cost.py:770-872) using the
 * masks / grids set on the context. */
int pmx_cv_masked(pmx_ctx* ctx, pmx_cv* cv, int win);
/* The "use_confidence" option of the SGM step (docs/source/userguide/plugins/plugin_libsgm.rst:38-47; the arithmetic is in the
 * un-vendored pandora_plugin_libsgm): E(D) = sum_p C(p, D_p) * Confidence(p) + penalties, i.e. every cost of pixel p is multiplied
 * by the pixel's confidence before the optimisation.  weights: float32 [H][W] host (NaN = no confidence for that pixel = 1);
 * the volume is brought to float32 and scaled in place, NaN costs stay NaN. */
int pmx_cv_scale_pixels(pmx_ctx* ctx, pmx_cv* cv, const float* weights);
/* Pixels whose cost is NaN for every disparity (np.min(np.isnan(cv), axis=2)), uint8 [H][W] on the
 * host: input of criteria.mask_invalid_variable_disparity_range (criteria.py:291-322). */
int pmx_nan_pixels(pmx_ctx* ctx, const pmx_cv* cv, uint8_t* host_out);
/* The same flags kept ON THE DEVICE with the volume: a snapshot of "NaN for every disparity" per pixel as the volume is when
 * this is called (the machine calls it where the reference calls criteria.mask_invalid_variable_disparity_range, right after
 * cv_masked: matching_cost/matching_cost.py:866-872); pmx_cv_get_missing copies the snapshot to the host, uint8 [H][W];
 * pmx_compose_validity consumes it where it is. */
int pmx_cv_mark_missing(pmx_ctx* ctx, pmx_cv* cv);
int pmx_cv_get_missing(pmx_ctx* ctx, const pmx_cv* cv, uint8_t* host_out);
/* matching_cost_cpp.reverse_cost_volume (matching_cost/cpp/src/matching_cost.cpp:26-56) */
pmx_cv* pmx_reverse_cost_volume(pmx_ctx* ctx, const pmx_cv* left_cv, int min_disp);

/* ---- aggregation ---------------------------------------------------------------------------- */
/* CrossBasedCostAggregation.cost_volume_aggregation (aggregation/cbca.py:90-182) =
 * median 3x3 (filter/median.py:134-179) + aggregation_cpp.cross_support
 * (aggregation/cpp/src/aggregation.cpp:224-321) + aggregation_cpp.cbca (:28-221,323-356) per d. */
int pmx_cbca(pmx_ctx* ctx, pmx_cv* cv, int offset, float intensity, int distance);
/* the arms alone (int16 [H-2o][W-2o][4]) for tests: side 0 = left, 1.. = right shifted image k-1 */
int pmx_cross_support(pmx_ctx* ctx, int side, int offset, float intensity, int distance, int16_t* host_out);

/* The reference's INNER native functions at their own granularity: what the pybind11 face pandora_amd.inner_cpp
 * (csrc/inner_face.cpp) binds, signature for signature, so that aggregation/cbca.py can swap its import and keep its python loop.
 * aggregation_cpp.cross_support(image, len_arms, intensity) (aggregation/cpp/src/aggregation.cpp:224-321): arms of a given image
 * (already filtered, invalid pixels +inf), int16 [H][W][4] = left, right, top, bottom. */
int pmx_cross_support_image(pmx_ctx* ctx, const float* image, int H, int W, int len_arms, float intensity, int16_t* host_out);
/* aggregation_cpp.cbca(input, cross_left, cross_right, range_col, range_col_right) (aggregation.cpp:323-356 = steps 1-4, :28-221)
 * on ONE disparity slice: input float32 [H][W], arms int16 [H][W][4] and [H][Wr][4], nvalid column pairs (left column, right
 * column); out_e = fully aggregated cost, out_n = number of support pixels WITHOUT the anchor (cbca.py:166 adds it), float32 [H][W]. */
int pmx_cbca_slice(pmx_ctx* ctx, const float* input, const int16_t* cross_left, const int16_t* cross_right, int H, int W, int Wr,
                   const int64_t* range_col, const int64_t* range_col_right, int nvalid, float* out_e, float* out_n);

/* ---- optimization --------------------------------------------------------------------------- */
/* AbstractOptimization.optimize_cv (optimization/optimization.py:104-123) for method "sgm"; the
 * arithmetic is external to the reference (pandora_plugin_libsgm==1.5.7): see DESIGN.md. */
int pmx_sgm(pmx_ctx* ctx, pmx_cv* cv, float P1, float P2, int is_max, float invalid_cost, int overcounting);
/* The same with P2 given per pixel and path direction: p2maps float32 [8][H][W] (host), directions in the order
 * (0,+1) (0,-1) (+1,0) (+1,+1) (+1,-1) (-1,0) (-1,+1) (-1,-1) of the step from p-r to p; map k holds, at pixel p, the P2 that
 * enters p's update on path k.  For the penalty methods that follow the left image's gradient along the path
 * (p2_method "negativeGradient" / "inverseGradient" of the libSGM plugin, docs/source/userguide/plugins/plugin_libsgm.rst:20-27,
 * 168-290; the plugin's Python builds the maps).  float32 kernels, one launch per direction.  PARITY UNPINNED like pmx_sgm. */
int pmx_sgm_p2maps(pmx_ctx* ctx, pmx_cv* cv, float P1, const float* p2maps, int is_max, float invalid_cost, int overcounting);

/* Debug / test hook: restrict the following pmx_sgm calls of this context to a subset of the eight paths.  Bit k of mask =
 * the k-th path of the definition's order (drow, dcol of the step towards the pixel): (0,+1) (0,-1) (+1,0) (+1,+1) (+1,-1)
 * (-1,0) (-1,+1) (-1,-1); 0xff (default) = all.  A strip of a full-size image can then be checked against the CPU oracle path
 * family by family (the horizontal paths of a row depend on that row alone).  A subset always runs the float32 kernels. */
int pmx_debug_sgm_directions(pmx_ctx* ctx, int mask);

/* ---- disparity / refinement ----------------------------------------------------------------- */
/* Upload the int64 validity mask computed by criteria.validity_mask (criteria.py:66-158);
 * NULL = zeros. */
int pmx_set_validity(pmx_ctx* ctx, const int64_t* validity);
/* The validity mask a cost volume carries into the disparity step, put together on the device instead of on the host
 * (criteria.validity_mask, criteria.py:66-158, then mask_invalid_variable_disparity_range :291-322 and mask_border :325-353):
 * validity(r, c) = base, then |= PANDORA_MSK_PIXEL_RIGHT_NODATA_OR_DISPARITY_RANGE_MISSING where `missing_of`'s snapshot
 * (pmx_cv_mark_missing; NULL: skip) is set, then the frame of `border` pixels (0: none) = PANDORA_MSK_PIXEL_LEFT_NODATA_OR_BORDER.
 * base: int64 host, base_rows == 1: one line of W flags for every row (the disparity-range criteria depend on the column only),
 * base_rows == H: a full map (input masks took part).  Replaces pmx_set_validity for that volume's pmx_wta. */
int pmx_compose_validity(pmx_ctx* ctx, const int64_t* base, int base_rows, const pmx_cv* missing_of, int border);
/* WinnerTakesAll.to_disp (disparity/disparity.py:399-516). Results stay on the device. */
int pmx_wta(pmx_ctx* ctx, const pmx_cv* cv, int is_max, float invalid_disparity);
/* Device-side copies of a result map (which: 0 disparity float32, 1 validity int64, 2 interpolated coefficient float32) as it is
 * now: the reference deep-copies 2-D results on the host where a later step overwrites them (cv["disp_indices"] =
 * disp_map["disparity_map"].copy(deep=True), disparity/disparity.py:459); here the copy stays in HBM (queued behind the kernels
 * that produce the map, nothing waits) and crosses PCIe only if it is read.  NULL on failure. */
void* pmx_map_snapshot(pmx_ctx* ctx, int which);
/* An uninitialised snapshot of the same kind: the output of a step that works on snapshots. */
void* pmx_map_snapshot_alloc(pmx_ctx* ctx, int which);
int pmx_map_snapshot_read(pmx_ctx* ctx, const void* snapshot, void* host_out);
void pmx_map_snapshot_free(pmx_ctx* ctx, void* snapshot);
/* Steps on snapshots - maps that stay in HBM from the WTA to the last filter (the reference hands numpy arrays from step to step;
 * the host-pointer forms of these steps, further down, are the same kernels between an upload and a download):
 *   pmx_maps_restore         the engine's disparity / validity maps become what two snapshots hold, e.g. the LEFT maps again after
 *                            the right side's WTA (state_machine.py:449-490 runs left and right through every step in turn);
 *   pmx_median_filter_maps   MedianFilter.filter_disparity (filter/median.py:94-131), out != in;
 *   pmx_cross_checking_maps  CrossChecking.disparity_checking (validation/validation.py:226-371): the left validity snapshot is
 *                            updated in place, the left-right distance goes to a float32 snapshot;
 *   pmx_validity_frame_map   criteria.mask_border (criteria.py:325-353) on a validity snapshot. */
int pmx_maps_restore(pmx_ctx* ctx, const void* disp_snapshot, const void* validity_snapshot);
int pmx_median_filter_maps(pmx_ctx* ctx, const void* disp_snapshot, const void* validity_snapshot, int filter_size,
                           void* out_disp_snapshot);
int pmx_cross_checking_maps(pmx_ctx* ctx, const void* disp_left, void* validity_left, const void* disp_right, int dmin, int dmax,
                            double threshold, void* conf_out);
int pmx_validity_frame_map(pmx_ctx* ctx, void* validity_snapshot, int border);
/* refinement_cpp.loop_refinement + vfit/quadratic (refinement/cpp/src/refinement.cpp:28-99,
 * vfit.cpp:28-56, quadratic.cpp:28-50) on the device-resident WTA result. */
int pmx_refine(pmx_ctx* ctx, const pmx_cv* cv, int method, int is_max);
/* refinement_cpp.loop_approximate_refinement (refinement/cpp/src/refinement.cpp:103-182; refinement_cpp.pyi:82-122; caller
 * AbstractRefinement.approximate_subpixel_refinement, refinement.py:124-158): refines the resident map as a RIGHT map found by the
 * diagonal search of the LEFT volume `cv_left` - pixel (row, col) with disparity d reads the left cells (row, col + d, -d) and its two
 * diagonal neighbours; borders of the range and of the diagonal stop the interpolation (mask += 8).  The resident map / mask / coefficient
 * are updated in place like pmx_refine's. */
int pmx_refine_approximate(pmx_ctx* ctx, const pmx_cv* cv_left, int method, int is_max);
/* Download disparity float32[H][W], validity int64[H][W], interpolated coeff float32[H][W]
 * (any pointer may be NULL). */
int pmx_get_disparity(pmx_ctx* ctx, float* disp, int64_t* validity, float* itp);
/* Upload a disparity map / validity (to refine a map that was edited on the host). */
int pmx_set_disparity(pmx_ctx* ctx, const float* disp, const int64_t* validity);

/* ---- D-sharded multi-GPU WTA ---------------------------------------------------- */
/* Per-pixel packed key (orderable cost bits << 31 | global disparity index; < 2^63, INT64_MAX = no
 * finite cost) of the local shard:
 * min over ranks of the key == np.argmin over the full volume, ties to the lowest index.
 * keys: uint64 [H][W] DEVICE pointer owned by the caller (e.g. a torch tensor fed to RCCL). */
int pmx_wta_minkey(pmx_ctx* ctx, const pmx_cv* cv, int is_max, int global_index_offset, uint64_t* dev_keys);
/* Decode reduced keys into the context's disparity/validity (d0_global = first disparity of the
 * full range). */
int pmx_wta_from_keys(pmx_ctx* ctx, const uint64_t* dev_keys, double d0_global, int subpix, float invalid_disparity);

/* ---- one pair over several GPUs: RCCL inside the library -------------------------- */
/* The reference has no collective; its only scaling convention is ROI tiles with a margin (optimization/optimization.py:43,
 * marge.py:86-101, img_tools.get_window img_tools.py:61-98).  One process per GPU; the communicator is RCCL's
 * (librccl.so, loaded on first use), created from a 128-byte id that rank 0 obtains with pmx_comm_unique_id and the launcher
 * hands to every rank (pandora_amd/comm.py does it over a TCP socket at MASTER_ADDR).  Collectives run on the context's stream
 * and touch only device buffers owned by the context ("exchange buffers"). */
enum { PMX_XBUF_KEYS = 0,          /* uint64 [H][W]   packed (cost, global index) keys of pmx_shard_minkey */
       PMX_XBUF_NANPIX = 1,        /* uint8  [H][W]   1 = the pixel is NaN for every disparity of this shard */
       PMX_XBUF_REFINE_PACK = 2,   /* float  [4][H][W] value-or-zero maps of pmx_shard_refine_pack */
       PMX_XBUF_REFINE_FLAGS = 3,  /* int64  [H][W]   validity bits the owner's refinement added */
       PMX_XBUF_FULL_DISP = 4,     /* float  [full_H][W]  row-tiled runs: the whole image's maps */
       PMX_XBUF_FULL_VALIDITY = 5, /* int64  [full_H][W] */
       PMX_XBUF_FULL_ITP = 6,      /* float  [full_H][W] */
       PMX_XBUF_SCALARS = 7,       /* double [8] */
       PMX_XBUF_FULL_VALIDITY16 = 8, /* uint16 [full_H][W]  the validity mask as it travels (and as the reference stores it) */
       PMX_XBUF_COUNT = 9 };
enum { PMX_OP_MIN = 0, PMX_OP_SUM = 1, PMX_OP_MAX = 2 };
int pmx_comm_unique_id(void* id_out, size_t bytes);                                  /* ncclGetUniqueId; bytes >= 128 */
int pmx_comm_init(pmx_ctx* ctx, const void* id, size_t bytes, int world, int rank);  /* ncclCommInitRank on the context's GPU */
int pmx_comm_destroy(pmx_ctx* ctx);
int pmx_comm_info(const pmx_ctx* ctx, int* world, int* rank);                        /* (1, 0) without a communicator */
/* what RCCL itself reports for the communicator (ncclCommCount; 1 without one): bench.py prints it beside WORLD_SIZE so that a run
 * that was meant to span N GPUs and did not cannot pass for one that did */
int pmx_comm_count(const pmx_ctx* ctx, int* nranks);
/* In-place ncclAllReduce of an exchange buffer over all ranks (keys: MIN = np.argmin over the full volume, ties to the lowest
 * index; NaN flags: MIN = NaN in every shard; refinement packs: SUM with exactly one non-zero contributor per pixel = exact). */
int pmx_comm_allreduce(pmx_ctx* ctx, int which, int op);
/* Eight host doubles through the same path (timings: MAX over ranks).  Identity without a communicator. */
int pmx_comm_allreduce_scalars(pmx_ctx* ctx, double* inout8, int op);
/* D shards (pipelines without SGM): per-pixel keys of this rank's disparity slice -> PMX_XBUF_KEYS (= pmx_wta_minkey); after the
 * MIN all-reduce pmx_shard_from_keys decodes the winners into the context's disparity / validity (= pmx_wta_from_keys). */
int pmx_shard_minkey(pmx_ctx* ctx, const pmx_cv* cv, int is_max, int global_index_offset);
int pmx_shard_from_keys(pmx_ctx* ctx, double d0_global, int subpix, float invalid_disparity);
/* np.min(np.isnan(cv), axis=2) of this shard -> PMX_XBUF_NANPIX (input of criteria.mask_invalid_variable_disparity_range,
 * criteria.py:291-322, once MIN-reduced over the shards and downloaded). */
int pmx_shard_nan_pixels(pmx_ctx* ctx, const pmx_cv* cv);
/* Sub-pixel refinement of a D-sharded volume (refinement/refinement.py:77-122 needs the winner's two neighbours: each shard
 * carries one disparity of halo): the rank whose owned disparities [own_lo, own_hi] hold a pixel's winner refines it; pack,
 * SUM all-reduce PMX_XBUF_REFINE_PACK and PMX_XBUF_REFINE_FLAGS, unpack -> the context's disparity / validity / coefficient. */
int pmx_shard_refine_pack(pmx_ctx* ctx, const pmx_cv* cv, int method, int is_max, double own_lo, double own_hi, int last_rank);
int pmx_shard_refine_unpack(pmx_ctx* ctx);
/* Row tiles (any pipeline; SGM paths are cut at the tile margin exactly as the reference's ROI runs cut them): the context's
 * maps are those of a tile starting at image row tile_lo; rows [own_lo, own_hi) go to their place in the full-size maps,
 * pmx_comm_allgather_rows (rows owned = contiguous split of full_H over the ranks, first full_H % world ranks one more) makes
 * every rank hold all rows, pmx_get_full_maps downloads them (any pointer may be NULL). */
int pmx_tile_place(pmx_ctx* ctx, int full_H, int own_lo, int own_hi, int tile_lo, int with_itp);
/* the same from host rows (a PandoraMachine run ends with its maps on the host: filters, validation); itp may be NULL */
int pmx_set_full_rows(pmx_ctx* ctx, int full_H, int own_lo, int own_hi, const float* disp, const int64_t* validity, const float* itp);
int pmx_comm_allgather_rows(pmx_ctx* ctx, int with_itp);
/* the same towards ONE rank (whoever downloads or saves the maps): a group of ncclSend / ncclRecv, every peer straight to the root over
 * its own xGMI link; the validity mask travels as uint16 (10 B/pixel with the coefficient map, 16 for the all-gather) */
int pmx_comm_gather_rows(pmx_ctx* ctx, int root, int with_itp);
int pmx_get_full_maps(pmx_ctx* ctx, float* disp, int64_t* validity, float* itp);
/* TEST TRANSPORT ONLY: host copies of an exchange buffer, so that two ranks sharing the one GPU of a test box (RCCL refuses that)
 * can reduce through the host.  *count = elements, *elem_bytes = bytes per element. */
int pmx_xbuf_info(pmx_ctx* ctx, int which, size_t* count, int* elem_bytes);
int pmx_xbuf_download(pmx_ctx* ctx, int which, void* host);
int pmx_xbuf_upload(pmx_ctx* ctx, int which, const void* host);

/* ---- measurement ---------------------------------------------------------------------------- */
/* When enabled, every kernel launch is bracketed by HIP events on the context's stream. */
int pmx_set_profiling(pmx_ctx* ctx, int enabled);
int pmx_reset_stage_times(pmx_ctx* ctx);
/* total GPU milliseconds and launch count of one stage since the last reset (syncs the stream) */
int pmx_stage_time(pmx_ctx* ctx, int stage, double* total_ms, int* launches);
/* Debug / test hook: copy the eight uint8 per-direction SGM path-cost volumes [8][H][W][Dp] of a handle
 * that is in the fused integer representation to the host.  *Dp receives the byte stride per pixel, *gl and
 * *kpl the lane map the kernels wrote them with: disparity index d of a pixel is byte
 * (k < (kpl & ~3)) ? s*(kpl & ~3) + k : nact*(kpl & ~3) + s   with s = d / kpl, k = d % kpl, nact = ceil(D / kpl).
 * Returns PMX_ERR_STATE when the handle is not in that representation. */
int pmx_debug_path_costs(pmx_ctx* ctx, const pmx_cv* cv, uint8_t* host_out, size_t host_bytes, int* Dp, int* gl, int* kpl);
/* Debug / test hook of the census + CBCA marching kernel (k_cbca.hip: cbca_census_march_kernel), whose cells end in the division of
 * two small integers: runs the kernel's division (hardware reciprocal + one Newton step) against the IEEE division for EVERY
 * numerator 0 .. 65535 and denominator 1 .. 1024 on the device; *mismatches receives how many quotients differ in any bit (0 is
 * the contract; aggregation.cpp:108-121 divides in float32). */
int pmx_debug_small_division(pmx_ctx* ctx, unsigned* mismatches);
/* Debug: the window table of the last marching pass of SGM (csrc/k_sgmfam.hip, k_sgmfam8.hip; tickets: csrc/pmx_buf.h
 * pmx_take_window): words [0..7] the first eight chunk counters (windows taken per chunk of consecutive windows; a counter may
 * overshoot its chunk's size), [8 + w] = 1 + the XCD window w ran on (0: never started).  Returns the number of words copied
 * (<= max_words) or a negative error.  No reference counterpart (the reference has no GPU code); tests/test_gpu_sgm_family.py checks
 * with it that every window of a launch was taken. */
int pmx_debug_fam_windows(pmx_ctx* ctx, unsigned* host_out, int max_words);
/* ---- validation --------------------------------------------------------------------------
 * Replaces validation.CrossCheckingAccurate.disparity_checking (src/pandora/validation/validation.py:226-371; the
 * class is registered for both "cross_checking_accurate" and "cross_checking_fast").  Host maps in/out, computed
 * on the device: validity_left int64 [H][W] is updated in place (PANDORA_MSK_PIXEL_OCCLUSION / _MISMATCH added
 * to the pixels the check rejects), conf_out float32 [H][W] receives |disp_right(q) + disp_left| (NaN where the
 * pixel is invalid or q = rint(col + disp_left) leaves the row).  [dmin, dmax] is the dataset's
 * "disparity_interval" (disparity.py:334-347).  mask_border (:368-369) stays with the caller. */
int pmx_cross_checking(pmx_ctx* ctx, const float* disp_left, int64_t* validity_left, const float* disp_right, int H, int W,
                       int dmin, int dmax, double threshold, float* conf_out);
/* Replaces validation_cpp.interpolate_occlusion_mc_cnn / interpolate_mismatch_mc_cnn / interpolate_occlusion_sgm /
 * interpolate_mismatch_sgm (src/pandora/validation/cpp/src/interpolated_disparity.cpp:232-296, :298-393, :101-139, :166-230),
 * the passes of AbstractInterpolation.interpolated_disparity (src/pandora/validation/interpolated_disparity.py:200-233 "mc-cnn":
 * occlusions then mismatches; :318-330 "sgm": mismatches then occlusions).  The n_passes passes run in the given order, each one
 * gathering from the maps the previous one produced; disp / validity: host maps, updated in place. */
enum { PMX_INTERP_OCCLUSION_MC_CNN = 0, PMX_INTERP_MISMATCH_MC_CNN = 1, PMX_INTERP_OCCLUSION_SGM = 2, PMX_INTERP_MISMATCH_SGM = 3 };
int pmx_interpolate_disparity(pmx_ctx* ctx, float* disp, int64_t* validity, int H, int W, const int* passes, int n_passes);
/* Replaces matching_cost_cpp.reverse_disp_range (matching_cost/cpp/src/matching_cost.cpp:59-132): per-pixel right
 * disparity ranges from the left ones; [global_min, global_max] must bracket every (int)left_min / (int)left_max. */
int pmx_reverse_disp_range(pmx_ctx* ctx, const float* left_min, const float* left_max, int H, int W, int global_min,
                           int global_max, float* right_min, float* right_max);
/* ---- disparity filters ---------------------------------------------------------------------
 * Replaces filter.MedianFilter.filter_disparity (src/pandora/filter/median.py:94-179): disp float32 [H][W] is
 * filtered in place on the pixels that are valid (validity & PANDORA_MSK_PIXEL_INVALID == 0) and finite, with
 * np.nanmedian over filter_size x filter_size in which invalid pixels are ignored; the frame of filter_size/2
 * pixels and the invalid pixels keep their values.  Host maps in/out, computed on the device. */
int pmx_median_filter_disparity(pmx_ctx* ctx, float* disp, const int64_t* validity, int H, int W, int filter_size);
/* Replaces filter.BilateralFilter.filter_disparity (src/pandora/filter/bilateral.py:100-255): same in-place protocol as
 * the median filter; window = min(H, W, int(3*sigma_space + 1)), float64 weighted means over the non-NaN window
 * elements (results within 1e-6 relative - 2e-6 absolute where the weighted mean cancels to about zero - of the reference's
 * float64 numpy arithmetic: the colour gaussian is a float32 exp on both sides, an ulp apart). */
int pmx_bilateral_filter_disparity(pmx_ctx* ctx, float* disp, const int64_t* validity, int H, int W, double sigma_color,
                                   double sigma_space);
/* Replaces filter.DisparityDenoiser.filter_disparity (src/pandora/filter/disparity_denoiser.py:223-313) after its get_grad
 * (:138-149, scipy's gaussian_filter + np.gradient: the caller runs them and passes the two gradient planes): a bilateral
 * filter of the distance to the local tangent plane over filter_size x filter_size windows of the maps padded with numpy's
 * "reflect"; weights = euclidian x colour x centred-planar gaussians (:290-295); in place on the pixels that are not invalid
 * and finite (:297-303).  color = the band of the left image the reference picks (:246-254), float32 [H][W].  Results within
 * 1e-6 relative of the reference's numpy arithmetic.  Host maps in/out, computed on the device. */
int pmx_denoise_disparity(pmx_ctx* ctx, float* disp, const int64_t* validity, const float* color, const float* grad_row,
                          const float* grad_col, int H, int W, int filter_size, double sigma_euclidian, double sigma_color,
                          double sigma_planar);
/* ---- multiscale ---------------------------------------------------------------------------- */
/* Replaces img_tools_cpp.interpolate_nodata_sgm (src/pandora/cpp/src/img_tools.cpp:99-155, called by
 * img_tools.fill_nodata_image, src/pandora/img_tools.py:578-613, before the pyramid is built): every pixel whose mask
 * has a bit of invalid_bits becomes the median of the first valid pixels along the 8 directions (NaN if none) and takes
 * the mask value filled_value; the others are copied.  Host maps in/out (int32 masks), computed on the device. */
int pmx_interpolate_nodata(pmx_ctx* ctx, const float* img, const int32_t* msk, int H, int W, int invalid_bits, int filled_value,
                           float* out_img, int32_t* out_msk);
/* Replaces the window search of multiscale.FixedZoomPyramid.disparity_range
 * (src/pandora/multiscale/fixed_zoom_pyramid.py:106-172) before its zoom: per valid pixel whose window fits, the
 * nanmin - marge / nanmax + marge of the valid disparities of the window; every other pixel gets
 * [global_min, global_max].  Host maps in/out, computed on the device. */
int pmx_disparity_range(pmx_ctx* ctx, const float* disp, const int64_t* validity, int H, int W, int window_size, int marge,
                        int global_min, int global_max, float* range_min, float* range_max);
/* ---- cost-volume confidence ---------------------------------------------------------------------
 * Replaces cost_volume_confidence_cpp.compute_ambiguity_and_sampled_ambiguity(..., sample_ambiguity=False)
 * (src/pandora/cost_volume_confidence/cpp/src/ambiguity.cpp:28-142): for every pixel the integral over etas of the
 * number of disparities whose normalised cost is within eta of the pixel's minimum, on the device-resident volume
 * (materialised to float32 if need be).  etas: float32 [nbr_etas] increasing (np.arange(eta_min, eta_max, eta_step));
 * grid_min / grid_max: int64 [H][W] per-pixel disparity range (NaN costs inside it count for every eta); negate != 0
 * for similarity measures (the reference flips the sign of the volume around the call, ambiguity.py:117-119).
 * ambiguity_out: float32 [H][W], not normalised.  grid_min == grid_max == NULL (here and in pmx_risk / pmx_interval_bounds)
 * means every pixel searches the volume's whole disparity range, which is what constant [min, max] inputs give; it saves the
 * upload of two int64 maps. */
int pmx_ambiguity(pmx_ctx* ctx, pmx_cv* cv, const float* etas, int nbr_etas, const int64_t* grid_min, const int64_t* grid_max,
                  int negate, float* ambiguity_out);

/* Replaces cost_volume_confidence_cpp.compute_ambiguity_and_sampled_ambiguity(..., True) followed by
 * compute_risk_and_sampled_risk(..., sample_risk=False), i.e. Risk.confidence_prediction's computation
 * (src/pandora/cost_volume_confidence/risk.py:139-166, cpp/src/risk.cpp:28-197): per pixel, the mean over etas of the span
 * of disparity indices whose normalised cost lies within eta of the pixel's minimum (risk_max), of
 * 1 + span - sampled ambiguity (risk_min), and of the disparities at the two ends of the span (disp_sup, disp_inf).
 * etas: float64, ascending, etas[0] >= 0 (the reference compares the risk in double and the ambiguity with float32 etas;
 * both are reproduced).  negate != 0 for similarity measures (risk.py:139-141).  Outputs: float32 [H][W]; NaN where
 * the pixel has no cost. */
int pmx_risk(pmx_ctx* ctx, pmx_cv* cv, const double* etas, int nbr_etas, const int64_t* grid_min, const int64_t* grid_max,
             int negate, float* risk_max, float* risk_min, float* disp_sup, float* disp_inf);

/* Replaces cost_volume_confidence_cpp.compute_interval_bounds (src/pandora/cost_volume_confidence/cpp/src/
 * interval_bounds.cpp:28-161; called by interval_bounds.py:158-170 with disp_interval == disparity_range == the volume's
 * disparities): the interval of disparities whose possibility type_factor * norm + 1 - max(type_factor * norm), taken
 * inside the pixel's [grid_min, grid_max], reaches possibility_threshold; type_factor is -1 for "min" measures and +1
 * for "max".  Outputs: float32 [H][W] disparities; NaN where the range holds no cost.  (The optional graph
 * regularisation of interval_tools.py is host-side work on the two maps and is not part of this library.) */
int pmx_interval_bounds(pmx_ctx* ctx, pmx_cv* cv, float possibility_threshold, float type_factor, const int64_t* grid_min,
                        const int64_t* grid_max, float* interval_inf, float* interval_sup);
/* Host helpers of the plugin layer (no GPU involved): the O(H*W) passes over the caller's arrays that the reference leaves to
 * numpy, on a few host threads.  pmx_host_minmax_i64: extrema of an integer disparity grid (np.nanmin / np.nanmax of
 * matching_cost.py:604-616 on an integer grid).  pmx_host_fingerprint: 64-bit content fingerprint of a buffer - "is this image the
 * resident one?" (the reference has no residency; any changed word changes the value; not cryptographic). */
int pmx_host_minmax_i64(const int64_t* a, size_t n, int64_t* out_min, int64_t* out_max);
uint64_t pmx_host_fingerprint(const void* data, size_t bytes);
/* Order statistics of a float32 array: out[i] = the ranks[i]-th smallest value (0-based; NaNs sort last, as in numpy) - what
 * np.percentile's partition provides to the percentile normalisation of the ambiguity measure
 * (cost_volume_confidence/ambiguity.py:168-184), by radix selection on the device instead of a host-side partition. */
int pmx_order_statistics(pmx_ctx* ctx, const float* values, size_t n, const size_t* ranks, int n_ranks, float* out);
/* Page-locked host memory for the caller's image / result arrays (the reference works in pageable numpy memory; over PCIe a
 * pinned buffer copies at the link rate and without first-touch page faults under the DMA).  NULL on failure. */
void* pmx_host_alloc(size_t bytes);
void pmx_host_free(void* p);
/* raw stream handle (hipStream_t) so a caller can enqueue its own work in order */
void* pmx_stream(pmx_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* PANDORA_AMD_H */
