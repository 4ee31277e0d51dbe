// pmx_internal.h - shared declarations of libpandora_amd.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "../../include/pandora_amd.h"

#define PMX_MAX_SUBPIX 4

void pmx_set_error(const char* fmt, ...);

#define PMX_HIP(expr)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            pmx_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            (void)hipGetLastError(); /* reported once: the runtime's sticky "last error" must not fail the NEXT call's launch check */ \
            return PMX_ERR_HIP;                                                                   \
        }                                                                                         \
    } while (0)

#define PMX_CHECK(cond, code, ...)      \
    do {                                \
        if (!(cond)) {                  \
            pmx_set_error(__VA_ARGS__); \
            return (code);              \
        }                               \
    } while (0)

struct pmx_fam_wta { float* disp; float* near; double d0; int subpix; float invalid_disparity; };

struct pmx_stage_rec {
    std::vector<hipEvent_t> ev;  // pairs (start, stop) not yet folded into total_ms
    double total_ms = 0.0;
    int launches = 0;
};

struct pmx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int H = 0, W = 0, subpix = 1;
    // resident pair; right[k] is the k-th shifted right image, width W (k=0) or W-1
    float* left = nullptr;
    float* right[PMX_MAX_SUBPIX] = {nullptr, nullptr, nullptr, nullptr};
    // masks: raw int16 + the cv_masked predicate (1 = invalid or dilated no-data); bad_win = window
    // the dilation was computed for (0 = stale)
    int16_t* msk_left = nullptr;
    int16_t* msk_right = nullptr;
    int valid_value = 0, nodata_value = 1;
    uint8_t* bad_left = nullptr;
    uint8_t* bad_right = nullptr;
    int bad_win = 0;
    double* grid_min = nullptr;
    double* grid_max = nullptr;
    // WTA / refinement results
    float* disp = nullptr;
    float* itp = nullptr;
    int64_t* validity = nullptr;
    void* near = nullptr;  // float4 [H][W]: (S[k-1], S[k], S[k+1], k) of the last WTA winner (fast path)
    bool disp_ready = false;  // a disparity map is resident for the current pair (pmx_wta / pmx_wta_from_keys / pmx_set_disparity)
    const void* near_owner = nullptr;  // the volume handle that cache was computed from (nullptr = stale)
    bool near_exact = false;  // the cache matches the resident disparity map pixel for pixel (no host edit since the WTA that wrote it)
    // a second winner cache (allocated when a second volume of the pair runs its WTA while the first still owns the other: the
    // left and right sides of a cross-checked run); pmx_near_select makes the one that belongs to a volume the active one
    void* near2 = nullptr;
    const void* near2_owner = nullptr;
    bool near2_exact = false;
    // scratch volume reused across calls (SGM accumulator, CBCA intermediate)
    float* scratch = nullptr;
    size_t scratch_bytes = 0;
    // generic small scratch (census codes, arms, medians)
    void* small = nullptr;
    size_t small_bytes = 0;
    // the gather of a row-tiled run on a stream of its own (pmx_comm_gather_rows): the next pair's kernels run under it
    hipStream_t comm_stream = nullptr;
    hipEvent_t placed_ev = nullptr, gathered_ev = nullptr;
    bool gather_pending = false;  // gathered_ev has been recorded and somebody may still have to wait for it
    // a second stream for kernels that are independent of each other inside ONE entry point (the horizontal pair of the integer SGM
    // runs beside the marching kernel of the vertical families, k_sgm8.hip): forked from and joined back into `stream` by events
    hipStream_t aux_stream = nullptr;
    hipEvent_t aux_fork = nullptr, aux_join = nullptr;
    // pinned staging of pmx_set_images (both images of a pair)
    char* stage_host = nullptr;
    size_t stage_cap = 0;
    // pinned staging of pmx_compose_validity's line (the copy is asynchronous; line_ev says when the buffer is free again)
    // (a ring of kLineSlots lines: a slot is only waited for when the ring comes round to it while its copy is still queued)
    static constexpr int kLineSlots = 8;
    int64_t* line_host = nullptr;
    int64_t* line_dev = nullptr;
    size_t line_cap = 0;  // elements per slot
    hipEvent_t line_ev[kLineSlots] = {};
    int line_next = 0;
    bool profiling = false;
    bool lazy = true;
    void* probe_sink = nullptr;   // 64 bytes the placement probe may write to
    int placement_trials = 6;  // pmx_set_placement_trials: candidates probed for every new volume-sized buffer (round 6: on by default)
    pmx_stage_rec stages[PMX_STAGE_COUNT];
    // caching allocator for the big per-volume buffers (a hipMalloc / hipFree of a few GB costs ~100 ms each on this
    // stack; a stream of pairs allocates and frees the same sizes over and over).  Single stream per context, so a
    // block handed out again is ordered after the kernels that used it before.
    std::unordered_map<void*, size_t> pool_live;       // blocks handed out -> their true size
    std::vector<std::pair<size_t, void*>> pool_free;   // cached blocks
    size_t pool_free_bytes = 0;
    // float32 SGM, family schedule (k_sgmfam.hip): the strip-to-strip hand-off buffer of tagged granules.  It belongs to the
    // context, is zeroed when (re)allocated and never lent to anything else, so a tag equal to the current epoch can only have
    // been written by the current launch; fam_epoch counts launches.  fam_ctl: [0] ticket counter, [1] error word (device).
    unsigned long long* fam_halo = nullptr;
    size_t fam_halo_bytes = 0;
    unsigned fam_epoch = 0;            // launches of the marching kernels (both kinds share the buffer and this counter)
    unsigned* fam_ctl = nullptr;
    unsigned* fam_xtab = nullptr;      // marching kernels: chunk counters of the window tickets + the windows' XCDs (pmx_buf.h pmx_take_window)
    size_t fam_xtab_words = 0;
    size_t fam_xtab_flags = 0;         // ... where the windows' XCD words start in it (behind the chunk counters)
    unsigned* fam_err_host = nullptr;  // pinned copy of the error word, filled behind every family launch
    int sgm_dir_mask = 0xff;           // pmx_debug_sgm_directions
    const float* sgm_p2maps = nullptr; // pmx_sgm_p2maps, for the duration of its call: P2 per pixel and direction (device)
    // one pair over several GPUs (pmx_comm.hip): RCCL communicator and the device buffers the collectives work on
    struct pmx_comm* comm = nullptr;
    void* xbuf[PMX_XBUF_COUNT] = {};
    size_t xbuf_bytes[PMX_XBUF_COUNT] = {};
    int full_H = 0;                    // rows of the whole image in a row-tiled run (pmx_tile_place)
    void* refine_saved[2] = {nullptr, nullptr};  // merged disparity / validity before the owner's refinement
    size_t refine_saved_bytes = 0;
    // kernel-route / tuning options (pmx_set_option; seeded once from PMX_<name> by pmx_create): slot i belongs to kPmxOptNames[i]
    static constexpr int kMaxOpts = 48;
    std::string opt_val[kMaxOpts];
    bool opt_set[kMaxOpts] = {};
};

// the value of a route option or nullptr (what getenv("PMX_<name>") used to answer, per context and without the environment)
const char* pmx_opt(const pmx_ctx* ctx, const char* name);

void pmx_comm_release(pmx_ctx* ctx);  // frees the exchange buffers

// an in-kernel hand-off that gave up (k_sgmfam.hip) is reported by the next call that synchronises the stream
int pmx_check_async_error(pmx_ctx* ctx, const char* where);

hipError_t pmx_pool_alloc(pmx_ctx* ctx, void** p, size_t bytes);
void pmx_pool_free(pmx_ctx* ctx, void* p);
void pmx_pool_release(pmx_ctx* ctx);  // hipFree every cached block

// exact representations a cost-volume handle can be in (see pmx_set_lazy in the public header)
// PMX_REPR_SGM_UP_PENDING (float32 family schedule, lazy mode): `data` still holds the matching costs, `spart` (and `spart2`) the
// sums of the horizontal and downward paths; the upward family has not run.  pmx_wta runs it in WTA mode (S is never written), anything that
// needs the optimised volume runs it in store mode first (pmx_cv_materialize).
enum { PMX_REPR_FLOAT = 0, PMX_REPR_ALL_NAN = 1, PMX_REPR_CENSUS_DEFERRED = 2, PMX_REPR_SGM_U8X8 = 3, PMX_REPR_SGM_UP_PENDING = 4 };

struct pmx_cv {
    pmx_ctx* ctx = nullptr;
    float* data = nullptr;
    size_t bytes = 0;  // capacity of data
    int H = 0, W = 0, D = 0, d0 = 0, subpix = 1;
    int repr = PMX_REPR_ALL_NAN;
    bool nonneg = true;  // every cost that is a number is >= +0 (census, SAD, SSD and what CBCA makes of them; all-NaN too)
    // census codes kept with the volume (fast path + deferred cost kernel)
    int win = 0;
    uint32_t* codes = nullptr;  // allocation: [pad | left | pad | right | pad]
    size_t codes_bytes = 0;
    const uint32_t* codeL = nullptr;
    const uint32_t* codeR = nullptr;
    // eight per-direction uint8 path-cost volumes [8][H][W][Dp]
    uint8_t* ldir = nullptr;
    size_t ldir_bytes = 0;
    int Dp = 0;
    size_t dstride = 0;  // bytes from one direction's volume to the next (H*W*Dp + a skew, see pmx_dir_stride)
    int gl = 0, kpl = 0;  // lane map of the fused kernels: gl lanes per scanline/pixel, kpl disparities per lane
    int nvol = 8;         // byte volumes behind PMX_REPR_SGM_U8X8: eight paths, or three direction families (k_sgmfam8.hip)
    // uint8 matching costs [H][W][Dp] in the same lane-map order (packed-arithmetic SGM path, k_sgm8.hip)
    uint8_t* cost8 = nullptr;
    size_t cost8_bytes = 0;
    // cv_masked on the integer fast path (per-pixel disparity grids and / or a left mask): the cells that are numbers are the
    // index interval [lo, hi) of each pixel, snapshot of geometry x grids x left mask taken when pmx_cv_masked ran;
    // range[pixel] = lo | hi << 16.  nullptr = census geometry alone.
    uint32_t* range = nullptr;
    size_t range_bytes = 0;
    bool has_range = false;
    // pmx_cv_mark_missing: snapshot of "the cost is NaN for every disparity" per pixel, uint8 [H][W], taken when it was called
    uint8_t* missing = nullptr;
    size_t missing_bytes = 0;
    bool has_missing = false;
    // PMX_REPR_SGM_UP_PENDING: the partial sum volumes (kept with the handle for the next pair) and what pmx_sgm was asked for:
    // spart = the horizontal pair's sum (or, when the downward family ran behind it, the sum of both), spart2 = the downward
    // family's own sum when it ran beside the pair (pending.two)
    float* spart = nullptr;
    size_t spart_bytes = 0;
    float* spart2 = nullptr;
    size_t spart2_bytes = 0;
    struct { float P1, P2, invalid_cost; int is_max, overcounting, two; } pending = {0.f, 0.f, 0.f, 0, 0, 0};
    size_t cells() const { return (size_t)H * (size_t)W * (size_t)D; }
};

// RAII-less helper: brackets a launch with events when profiling is on
struct pmx_stage_scope {
    pmx_ctx* ctx;
    int stage;
    hipStream_t st;  // the stream the bracketed work is queued on (the context's own unless said otherwise)
    hipEvent_t a = nullptr, b = nullptr;
    pmx_stage_scope(pmx_ctx* c, int s, hipStream_t on = nullptr) : ctx(c), stage(s), st(on ? on : c->stream) {
        // a failed event only loses a timing sample: pmx_stage_time skips null events
        if (ctx->profiling && hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) {
            if (hipEventRecord(a, st) != hipSuccess) drop();
        } else {
            drop();
        }
    }
    ~pmx_stage_scope() {
        if (a && b) {
            if (hipEventRecord(b, st) == hipSuccess) {
                ctx->stages[stage].ev.push_back(a);
                ctx->stages[stage].ev.push_back(b);
            } else {
                drop();
            }
        }
    }
    void drop() {
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
        a = b = nullptr;
    }
};

int pmx_need_scratch(pmx_ctx* ctx, size_t bytes);
int pmx_need_small(pmx_ctx* ctx, size_t bytes);
int pmx_update_bad_masks(pmx_ctx* ctx, int win);

static inline int pmx_shifted_width(int W, int k) { return k == 0 ? W : W - 1; }

// parameters shared by every matching-cost kernel: geometry + the cv_masked predicate inputs
constexpr size_t kImgGuardBytes = 256;  // zeroed bytes before and after every device image

struct pmx_mc_params {
    int H, W, D, d0, subpix, win;
    const float* left;
    const float* right[PMX_MAX_SUBPIX];
    const uint8_t* bad_left;   // may be null
    const uint8_t* bad_right;  // may be null
    const double* grid_min;    // may be null
    const double* grid_max;
    int apply_mask;
};

// kernels (one translation unit each)
int pmx_launch_shift_right(pmx_ctx* ctx, const float* R, int H, int W, int subpix, int k, float* out);
int pmx_launch_census(pmx_ctx* ctx, pmx_cv* cv, int win, bool defer_costs);
int pmx_launch_census_costs(pmx_ctx* ctx, pmx_cv* cv);  // cost kernel from the codes held by cv
int pmx_cv_materialize(pmx_ctx* ctx, pmx_cv* cv);       // any representation -> float32 volume
int pmx_cv_ensure_data(pmx_ctx* ctx, pmx_cv* cv);       // allocate the float32 storage on first need
bool pmx_fused_sgm_eligible(const pmx_ctx* ctx, const pmx_cv* cv, float P1, float P2, int is_max, float invalid_cost, int overcounting);
int pmx_launch_sgm_fused(pmx_ctx* ctx, pmx_cv* cv, float P1, float P2, float invalid_cost);
int pmx_launch_sum8_wta(pmx_ctx* ctx, const pmx_cv* cv, float invalid_disparity);
bool pmx_sgm8_supported(int gl, int kpl, int nw);
size_t pmx_dir_stride(int H, int W, int Dp);  // spacing of the eight path volumes
int pmx_launch_build_range(pmx_ctx* ctx, pmx_cv* cv);        // geometry x ctx grids x ctx bad_left -> cv->range
int pmx_launch_range_nan(pmx_ctx* ctx, pmx_cv* cv);          // float volume: NaN outside cv->range
int pmx_launch_sgm8(pmx_ctx* ctx, pmx_cv* cv, int kpl, uint32_t P1, uint32_t P2, uint32_t invalid_cost);  // cost8 + 8 path volumes
int pmx_launch_sum8_refine(pmx_ctx* ctx, const pmx_cv* cv, int method);
int pmx_launch_sum8_to_float(pmx_ctx* ctx, pmx_cv* cv);
int pmx_launch_census_nan_pixels(pmx_ctx* ctx, const pmx_cv* cv, uint8_t* dev_out);
int pmx_launch_cross_checking(pmx_ctx* ctx, const float* dl, int64_t* validity, const float* dr, int H, int W, int dmin, int dmax,
                              double threshold, float* conf);
int pmx_launch_interpolate_disparity(pmx_ctx* ctx, int pass, const float* disp, const int64_t* valid, int H, int W, float* out_disp,
                                     int64_t* out_valid);
int pmx_launch_reverse_disp_range(pmx_ctx* ctx, const float* lmin, const float* lmax, int H, int W, int gmin, int gmax, float* rmin,
                                  float* rmax);
int pmx_launch_denoise_disparity(pmx_ctx* ctx, const float* in, const int64_t* validity, const float* color, const float* grad_row,
                                 const float* grad_col, int H, int W, int ws, const double* ge, double sigma_color, double sigma_planar,
                                 float* out);
int pmx_launch_bilateral_disparity(pmx_ctx* ctx, const float* in, const int64_t* validity, int H, int W, int win, const double* gs,
                                   double sigma_color, float* out);
int pmx_launch_disparity_range(pmx_ctx* ctx, const float* disp, const int64_t* validity, int H, int W, int win, int marge, int gmin,
                               int gmax, float* out_min, float* out_max);
int pmx_launch_scale_pixels(pmx_ctx* ctx, pmx_cv* cv, const float* d_weights);
int pmx_launch_ambiguity(pmx_ctx* ctx, pmx_cv* cv, const float* d_etas, int nbr_etas, const int64_t* d_gmin, const int64_t* d_gmax,
                         int negate, uint32_t* d_mm, float* d_amb);
int pmx_launch_risk(pmx_ctx* ctx, pmx_cv* cv, const double* d_etas, int nbr_etas, const int64_t* d_gmin, const int64_t* d_gmax,
                    int negate, uint32_t* d_mm, float* d_out4);
int pmx_launch_interval_bounds(pmx_ctx* ctx, pmx_cv* cv, float threshold, float type_factor, const int64_t* d_gmin,
                               const int64_t* d_gmax, uint32_t* d_mm, float* d_out2);
int pmx_launch_interpolate_nodata(pmx_ctx* ctx, const float* img, const int* msk, int H, int W, int invalid_bits, int filled,
                                  float* out_img, int* out_msk);
int pmx_launch_median_disparity(pmx_ctx* ctx, const float* in, const int64_t* validity, int H, int W, int size, float* out);
int pmx_launch_sad_ssd(pmx_ctx* ctx, pmx_cv* cv, int win, int squared);
int pmx_launch_zncc(pmx_ctx* ctx, pmx_cv* cv, int win);
int pmx_launch_cv_masked(pmx_ctx* ctx, pmx_cv* cv, int win);
int pmx_launch_mask_dilate(pmx_ctx* ctx, const int16_t* msk, int H, int W, int win, int valid, int nodata, uint8_t* bad);
int pmx_launch_fill_nan(pmx_ctx* ctx, float* p, size_t n);
int pmx_launch_sgm(pmx_ctx* ctx, pmx_cv* cv, float P1, float P2, int is_max, float invalid_cost, int overcounting);
bool pmx_sgm_family_supported(const pmx_ctx* ctx, const pmx_cv* cv);
// One marching pass (k_sgmfam.hip): family `fam` (0: the three downward paths, 1: the three upward ones), the paths of `bits`
// (bit 0 vertical, 1 predecessor column c-1, 2 predecessor column c+1) summed on their own and written as out = [in1 +] [in2 +]
// that sum; epilogue: the pass finishes S.  wta != nullptr: `out` is not written, the pass reduces over D (WTA mode).  st: the stream
// (nullptr = the context's own).
int pmx_launch_sgm_family(pmx_ctx* ctx, pmx_cv* cv, int fam, const float* in1, const float* in2, float* out, float P1, float P2,
                          int is_max, float invalid_cost, int overcounting, int bits, bool epilogue, const pmx_fam_wta* wta,
                          hipStream_t st);
int pmx_cus_per_xcd();  // CUs of one XCD of the current device
int pmx_sgm_family_prepare(pmx_ctx* ctx, const pmx_cv* cv);  // the hand-off buffer of the float32 marching passes, on the context's stream
int pmx_fam_prepare(pmx_ctx* ctx, size_t halo_bytes);  // hand-off buffer + ticket / error words of the marching kernels
// integer path as direction families (k_sgmfam8.hip): the vertical families' byte sums into out + f * dstride
int pmx_fam8_waves(const pmx_ctx* ctx, int W);
bool pmx_fam8_supported(int kpl, int H);
// The tag a launch's hand-off blocks carry: the launch count scrambled over all 32 bits, top bit set (a zeroed buffer never
// matches).  A block is taken when its tag word equals the tag (k_sgmfam8.hip) or the tag XOR its payload (k_sgmfam.hip): with
// consecutive small epochs a stale or half-arrived block could pass such a test by coincidence of small numbers (a stale packed
// byte block whose two halves XOR to the epoch, a torn block whose old and new payloads differ by epoch ^ epoch'); with scrambled
// tags a coincidence needs 31 matching pseudo-random bits.  Distinct launches within 2^31 of each other have distinct tags.
static inline unsigned pmx_fam_tag(unsigned epoch) { return (epoch * 0x9E3779B9u) | 0x80000000u; }
int pmx_launch_sgm_fam8(pmx_ctx* ctx, pmx_cv* cv, int kpl, bool five, int Dc, uint8_t* out, size_t dstride, uint32_t P1, uint32_t P2,
                        int fams, bool from_codes, uint32_t invalid_cost);
int pmx_launch_wta(pmx_ctx* ctx, const pmx_cv* cv, int is_max, float invalid_disparity);
int pmx_launch_refine(pmx_ctx* ctx, const pmx_cv* cv, int method, int is_max);
int pmx_launch_approx_refine(pmx_ctx* ctx, const pmx_cv* cv, int method, int is_max);
int pmx_near_select(pmx_ctx* ctx, const pmx_cv* cv, bool for_write);  // the winner cache of `cv` becomes ctx->near (see pmx_api.hip)
void pmx_near_forget(pmx_ctx* ctx, const pmx_cv* cv);                 // the volume changed: no cache describes it any more
int pmx_launch_near_refine(pmx_ctx* ctx, const pmx_cv* cv, int method, int is_max);  // from the winner's three values (ctx->near)
int pmx_launch_wta_fixup(pmx_ctx* ctx, const pmx_cv* cv);                          // validity of the pixels a fused WTA found all-NaN
int pmx_sgm_finish_pending(pmx_ctx* ctx, pmx_cv* cv, const pmx_fam_wta* wta);      // runs the upward family of a pending volume
int pmx_launch_cbca(pmx_ctx* ctx, pmx_cv* cv, int offset, float intensity, int distance, bool census_src);
bool pmx_cbca_can_fuse_census(const pmx_ctx* ctx, const pmx_cv* cv, int offset, int distance);
int pmx_launch_small_division_check(pmx_ctx* ctx, unsigned* host_count);
int pmx_launch_cross_support(pmx_ctx* ctx, int side, int offset, float intensity, int distance, int16_t* dev_out);
int pmx_comm_join(pmx_ctx* ctx);  // the context's stream waits for a gather still running on the communication stream
int pmx_launch_nan_pixels(pmx_ctx* ctx, const pmx_cv* cv, uint8_t* dev_out);
int pmx_launch_order_statistic(pmx_ctx* ctx, const float* dev_values, size_t n, size_t rank, uint32_t* dev_hist, uint32_t* host_hist, float* out);
int pmx_launch_compose_validity(pmx_ctx* ctx, const int64_t* dev_base, int base_rows, const uint8_t* dev_missing, int border);
int pmx_launch_compose_validity_into(pmx_ctx* ctx, const int64_t* dev_base, int base_rows, const uint8_t* dev_missing, int border,
                                     int64_t* dev_out);
int pmx_launch_reverse(pmx_ctx* ctx, const pmx_cv* in, int min_disp, pmx_cv* out);
int pmx_launch_minkey(pmx_ctx* ctx, const pmx_cv* cv, int is_max, int index_offset, uint64_t* keys);
int pmx_launch_from_keys(pmx_ctx* ctx, const uint64_t* keys, double d0, int subpix, float invalid_disparity);
