// pmx_api.hip - context, residency and the extern "C" surface of libpandora_amd.so.
// Every entry point is declared (with the reference interface it replaces) in include/pandora_amd.h.
#include <cstdarg>
#include <cstring>

#include <cstdio>
#include <thread>
#include <utility>
#include <vector>
#include <cstdlib>

#include "pmx_internal.h"

static thread_local char g_err[512] = "";

void pmx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* pmx_last_error(void) { return g_err; }

extern "C" int pmx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---- kernel-route / tuning options ---------------------------------------------------------------------------------------------------
// Every name the library looks at (DESIGN 7b says what each one forces).  The ONE place the environment is read: pmx_create.
static const char* const kPmxOptNames[] = {
    "SGM_SCHED", "SGM_PAR", "SGM_HFUSED", "SGM_PENDING", "SGM_FAM_SHAPE", "SGM_FAM_PAR", "SGM_FAM_XCD",
    "SGM8", "FUSED_MAP", "COST5", "WTA3", "SGM8_FAM", "SGM8_HPAIR", "SGM8_CODES", "SGM8_FAMCODES", "SGM8_OVERLAP", "SGM8_HF",
    "SGM8_FAM_NW", "SGM8_FAM_PRIO", "SGM8_FAM_XCD",
    "CBCA_ARMS_FLAT", "CBCA_ROWS", "CBCA_FAST", "CBCA_FUSE", "CBCA_MARCH", "CBCA_VBUF", "CBCA_SIGN", "CBCA_GEO", "CBCA_VBS",
    "CBCA_ROWDESC", "COMM_OVERLAP",
};
static constexpr int kPmxOptCount = (int)(sizeof(kPmxOptNames) / sizeof(kPmxOptNames[0]));
static_assert(kPmxOptCount <= pmx_ctx::kMaxOpts, "option slots");
static int opt_index(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < kPmxOptCount; ++i)
        if (!strcmp(name, kPmxOptNames[i])) return i;
    return -1;
}
const char* pmx_opt(const pmx_ctx* ctx, const char* name) {
    const int i = opt_index(name);
    if (i < 0) {  // a kernel hook that is not in the table (a typo, a new hook): the route a test forces would silently be the default one
        fprintf(stderr, "libpandora_amd: pmx_opt(\"%s\") is not in kPmxOptNames (pmx_api.hip)\n", name ? name : "(null)");
        abort();
    }
    return (i >= 0 && ctx && ctx->opt_set[i]) ? ctx->opt_val[i].c_str() : nullptr;
}
extern "C" int pmx_set_option(pmx_ctx* ctx, const char* name, const char* value) {
    PMX_CHECK(ctx, PMX_ERR_ARG, "pmx_set_option: null context");
    const int i = opt_index(name);
    PMX_CHECK(i >= 0, PMX_ERR_ARG, "pmx_set_option: unknown option '%s'", name ? name : "(null)");
    ctx->opt_set[i] = value != nullptr;
    ctx->opt_val[i] = value ? value : "";
    return PMX_OK;
}
extern "C" const char* pmx_get_option(const pmx_ctx* ctx, const char* name) {
    return opt_index(name) >= 0 ? pmx_opt(ctx, name) : nullptr;  // (a caller's unknown name answers "not set"; only the library's own lookups abort)
}
extern "C" const char* pmx_option_name(int index) { return index >= 0 && index < kPmxOptCount ? kPmxOptNames[index] : nullptr; }

extern "C" pmx_ctx* pmx_create(int device) {
    int n = pmx_device_count();
    if (device < 0 || device >= n) {
        pmx_set_error("pmx_create: device %d not available (%d HIP devices visible)", device, n);
        return nullptr;
    }
    if (hipSetDevice(device) != hipSuccess) {
        pmx_set_error("pmx_create: hipSetDevice(%d) failed", device);
        return nullptr;
    }
    pmx_ctx* ctx = new pmx_ctx();
    ctx->device = device;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        pmx_set_error("pmx_create: hipStreamCreate failed");
        delete ctx;
        return nullptr;
    }
    for (int i = 0; i < kPmxOptCount; ++i) {  // the options' start values: PMX_<name> of the environment, read here and nowhere else
        const std::string var = std::string("PMX_") + kPmxOptNames[i];
        if (const char* v = getenv(var.c_str())) {
            ctx->opt_set[i] = true;
            ctx->opt_val[i] = v;
        }
    }
    return ctx;
}

// Images sit between two zeroed guards of kImgGuardBytes so that wide loads which start a few elements before
// the first row or run past the last one stay inside the allocation.
static hipError_t img_alloc(pmx_ctx* ctx, float** p, size_t n) {
    char* raw = nullptr;
    hipError_t e = hipMalloc((void**)&raw, n * sizeof(float) + 2 * kImgGuardBytes);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(raw, 0, n * sizeof(float) + 2 * kImgGuardBytes, ctx->stream);
    *p = (float*)(raw + kImgGuardBytes);
    return e;
}
static void img_free(float*& p) {
    if (p) hipFree((char*)p - kImgGuardBytes);
    p = nullptr;
}

// masks and disparity grids belong to one pair
static void free_pair_extras(pmx_ctx* ctx) {
    hipFree(ctx->msk_left); ctx->msk_left = nullptr;
    hipFree(ctx->msk_right); ctx->msk_right = nullptr;
    hipFree(ctx->bad_left); ctx->bad_left = nullptr;
    hipFree(ctx->bad_right); ctx->bad_right = nullptr;
    hipFree(ctx->grid_min); ctx->grid_min = nullptr;
    hipFree(ctx->grid_max); ctx->grid_max = nullptr;
    ctx->bad_win = 0;
}

static void free_images(pmx_ctx* ctx) {
    img_free(ctx->left);
    for (int k = 0; k < PMX_MAX_SUBPIX; ++k) img_free(ctx->right[k]);
    hipFree(ctx->msk_left); ctx->msk_left = nullptr;
    hipFree(ctx->msk_right); ctx->msk_right = nullptr;
    hipFree(ctx->bad_left); ctx->bad_left = nullptr;
    hipFree(ctx->bad_right); ctx->bad_right = nullptr;
    hipFree(ctx->grid_min); ctx->grid_min = nullptr;
    hipFree(ctx->grid_max); ctx->grid_max = nullptr;
    hipFree(ctx->disp); ctx->disp = nullptr;
    hipFree(ctx->itp); ctx->itp = nullptr;
    hipFree(ctx->validity); ctx->validity = nullptr;
    hipFree(ctx->near); ctx->near = nullptr;
    hipFree(ctx->near2); ctx->near2 = nullptr;
    ctx->near2_owner = nullptr;
    ctx->bad_win = 0;
}

extern "C" void pmx_destroy(pmx_ctx* ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    free_images(ctx);
    pmx_pool_free(ctx, ctx->scratch);
    hipFree(ctx->small);
    hipFree(ctx->probe_sink);
    pmx_comm_destroy(ctx);
    pmx_comm_release(ctx);
    if (ctx->aux_stream) {
        hipStreamSynchronize(ctx->aux_stream);
        hipStreamDestroy(ctx->aux_stream);
    }
    if (ctx->aux_fork) hipEventDestroy(ctx->aux_fork);
    if (ctx->aux_join) hipEventDestroy(ctx->aux_join);
    hipFree(ctx->fam_halo);
    hipFree(ctx->fam_ctl);
    hipFree(ctx->fam_xtab);
    if (ctx->fam_err_host) hipHostFree(ctx->fam_err_host);
    if (ctx->line_host) hipHostFree(ctx->line_host);
    if (ctx->stage_host) hipHostFree(ctx->stage_host);
    if (ctx->line_dev) hipFree(ctx->line_dev);
    for (hipEvent_t e : ctx->line_ev)
        if (e) hipEventDestroy(e);
    pmx_pool_release(ctx);
    for (auto& s : ctx->stages)
        for (auto e : s.ev) hipEventDestroy(e);
    hipStreamDestroy(ctx->stream);
    delete ctx;
}

// Reads the error word of the in-kernel hand-offs (k_sgmfam.hip) once the stream is idle.  A hand-off only gives up after seconds
// of polling, which means a broken build or a dying device, never a property of the data: the results on the device are then
// incomplete and every synchronising entry point reports it.
int pmx_check_async_error(pmx_ctx* ctx, const char* where) {
    if (!ctx->fam_err_host || *ctx->fam_err_host == 0) return PMX_OK;
    const unsigned code = *ctx->fam_err_host;
    *ctx->fam_err_host = 0;
    (void)hipMemsetAsync(ctx->fam_ctl + 1, 0, sizeof(unsigned), ctx->stream);
    pmx_set_error("%s: the float32 SGM family kernel gave up waiting for a neighbouring window (code %u): results are incomplete", where,
                  code);
    return PMX_ERR_STATE;
}

extern "C" int pmx_sync(pmx_ctx* ctx) {
    PMX_CHECK(ctx, PMX_ERR_ARG, "pmx_sync: null context");
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->comm_stream) PMX_HIP(hipStreamSynchronize(ctx->comm_stream));
    return pmx_check_async_error(ctx, "pmx_sync");
}

extern "C" int pmx_debug_sgm_directions(pmx_ctx* ctx, int mask) {
    PMX_CHECK(ctx, PMX_ERR_ARG, "pmx_debug_sgm_directions: null context");
    PMX_CHECK(mask > 0 && mask <= 0xff, PMX_ERR_ARG, "pmx_debug_sgm_directions: mask must be in 1..255, got %d", mask);
    ctx->sgm_dir_mask = mask;
    return PMX_OK;
}

// page-locked host memory for the caller's result / input arrays: DMA at full PCIe rate, no first-touch page faults under the copy
extern "C" void* pmx_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        pmx_set_error("pmx_host_alloc: hipHostMalloc(%zu) failed", bytes);
        return nullptr;
    }
    return p;
}

extern "C" void pmx_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

// ---- host helpers of the plugin layer: the O(H*W) passes over the caller's arrays that the reference does in numpy, spread
// over a few host threads (a 4 Mpx int64 grid is 32 MB: one core reads it in a millisecond or two, eight in a fraction) ---------
namespace {
constexpr int kHostThreads = 8;  // (a 32 MB scan stops gaining at 8 on the MI355X hosts; every thread costs ~20 us to start)
template <typename F>
void host_chunks(size_t n, size_t min_chunk, F&& body) {  // body(chunk index, begin, end); chunk count <= kHostThreads
    unsigned hw = std::thread::hardware_concurrency();
    static const int cap = getenv("PMX_HOST_THREADS") ? atoi(getenv("PMX_HOST_THREADS")) : kHostThreads;
    size_t nt = hw ? (hw < (unsigned)cap ? hw : cap) : 1;
    if (nt > (size_t)kHostThreads) nt = kHostThreads;
    if (nt < 1) nt = 1;
    if (n / (min_chunk ? min_chunk : 1) < nt) nt = n / (min_chunk ? min_chunk : 1);
    if (nt <= 1) {
        body(0, (size_t)0, n);
        return;
    }
    std::vector<std::thread> th;
    const size_t per = (n + nt - 1) / nt;
    for (size_t t = 1; t < nt; ++t) th.emplace_back([&, t] { body(t, t * per < n ? t * per : n, (t + 1) * per < n ? (t + 1) * per : n); });
    body(0, (size_t)0, per < n ? per : n);
    for (auto& x : th) x.join();
}
constexpr uint64_t kMix = 0x9E3779B97F4A7C15ull;
}  // namespace

extern "C" int pmx_host_minmax_i64(const int64_t* a, size_t n, int64_t* out_min, int64_t* out_max) {
    PMX_CHECK(a && n && out_min && out_max, PMX_ERR_ARG, "pmx_host_minmax_i64: empty input or null output");
    int64_t mn[kHostThreads], mx[kHostThreads];
    for (int i = 0; i < kHostThreads; ++i) { mn[i] = INT64_MAX; mx[i] = INT64_MIN; }
    host_chunks(n, 1u << 17, [&](size_t t, size_t b, size_t e) {
        int64_t lo = INT64_MAX, hi = INT64_MIN;
        for (size_t i = b; i < e; ++i) {
            lo = a[i] < lo ? a[i] : lo;
            hi = a[i] > hi ? a[i] : hi;
        }
        mn[t] = lo;
        mx[t] = hi;
    });
    int64_t lo = mn[0], hi = mx[0];
    for (int i = 1; i < kHostThreads; ++i) { lo = mn[i] < lo ? mn[i] : lo; hi = mx[i] > hi ? mx[i] : hi; }
    *out_min = lo;
    *out_max = hi;
    return PMX_OK;
}

// Content fingerprint of a buffer (is the array the caller hands over the one that is resident?): eight interleaved lanes of
// h = (h ^ word) * odd - every step is a bijection of the lane, so changing any one word changes the result for certain, several
// words with probability 1 - 2^-64.  Not a cryptographic hash.
static uint64_t fingerprint_copy(void* dst, const void* data, size_t bytes) {  // dst: optional copy made in the same pass
    if (!data || !bytes) return kMix;
    const uint8_t* p = (const uint8_t*)data;
    uint8_t* q = (uint8_t*)dst;
    uint64_t part[kHostThreads] = {};
    const size_t blocks = bytes / 64;
    host_chunks(blocks, 1u << 14, [&](size_t t, size_t b, size_t e) {
        uint64_t h[8];
        for (int l = 0; l < 8; ++l) h[l] = kMix * (uint64_t)(l + 1);
        for (size_t i = b; i < e; ++i) {
            uint64_t w[8];
            memcpy(w, p + i * 64, 64);
            if (q) memcpy(q + i * 64, w, 64);
            for (int l = 0; l < 8; ++l) h[l] = (h[l] ^ w[l]) * kMix;
        }
        uint64_t acc = 0;
        for (int l = 0; l < 8; ++l) acc = (acc ^ h[l]) * kMix + (h[l] >> 29);
        part[t] = acc;
    });
    uint64_t acc = bytes;
    // (the value depends on how the buffer was cut into chunks, i.e. on the thread count - constant within a process, and
    //  fingerprints are only ever compared within one)
    for (int t = 0; t < kHostThreads; ++t) acc = (acc ^ part[t]) * kMix + (acc >> 31);
    for (size_t i = blocks * 64; i < bytes; ++i) {
        if (q) q[i] = p[i];
        acc = (acc ^ p[i]) * kMix;
    }
    return acc ^ (acc >> 32);
}

extern "C" uint64_t pmx_host_fingerprint(const void* data, size_t bytes) { return fingerprint_copy(nullptr, data, bytes); }

extern "C" void* pmx_stream(pmx_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

// ---- placement-aware allocation (pmx_set_placement_trials; on by default since round 6) ---------------------------------------------------
// On MI355X the bandwidth a kernel gets from a hipMalloc'd buffer is a property of that buffer: of several multi-GB buffers
// of one process some read 8 % faster than the others, reproducibly (tools/ubench/streams8.hip).  A context that
// will reuse its volumes for many pairs can afford to choose: allocate up to `placement_trials` candidates (all held until the
// choice is made, so that they are different memory), time one streaming fill and read of each, keep the fastest.
__global__ __launch_bounds__(256) void placement_probe_kernel(const uint4* __restrict__ p, size_t n, uint32_t* __restrict__ sink) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t step = (size_t)gridDim.x * 256;
    uint32_t acc = 0;
    for (; i < n; i += step) {
        const uint4 v = p[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x9e3779b9u) sink[0] = acc;  // (keeps the loads alive)
}

__global__ __launch_bounds__(256) void stream_fill_kernel(uint4* __restrict__ p, size_t n, uint32_t v);
static float placement_probe_ms(pmx_ctx* ctx, const void* buf, size_t bytes) {
    if (!ctx->probe_sink && hipMalloc(&ctx->probe_sink, 64) != hipSuccess) return -1.f;
    hipEvent_t a = nullptr, b = nullptr;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1.f;
    // a fill and a read, best of two each after a first pass that also faults the pages in; their sum is the candidate's mark.
    // (The fill separates the faster from the slower stretches of a device's memory more sharply than the read - 6.4 against
    // 6.0 TB/s with 2 % between passes, tools/ubench/phys_map.hip - the read is what the consumers of a volume do; alternated on
    // one box with six candidates: read only 12.54 / 12.79 / 12.77 / 13.53 ms per headline step, fill only 12.78 / 12.72 / 12.89 /
    // 12.81, both 12.59 / 12.56 / 12.89 / 12.89, plain hipMalloc 13.5 - 14.1: profiles/r05_k_probe_kinds.txt.)
    float best_r = -1.f, best_w = -1.f;
    for (int rep = 0; rep < 3; ++rep) {
        for (int kind = 0; kind < 2; ++kind) {
            (void)hipEventRecord(a, ctx->stream);
            if (kind == 0) hipLaunchKernelGGL(stream_fill_kernel, dim3(16384), dim3(256), 0, ctx->stream, (uint4*)buf, bytes / 16, (uint32_t)rep);
            else hipLaunchKernelGGL(placement_probe_kernel, dim3(16384), dim3(256), 0, ctx->stream, (const uint4*)buf, bytes / 16, (uint32_t*)ctx->probe_sink);
            (void)hipEventRecord(b, ctx->stream);
            if (hipEventSynchronize(b) != hipSuccess) break;
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, a, b) == hipSuccess && rep > 0) {
                float& best = kind == 0 ? best_w : best_r;
                if (best < 0.f || ms < best) best = ms;
            }
        }
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    return (best_r > 0.f && best_w > 0.f) ? best_r + best_w : -1.f;
}

// ---- what plain streaming kernels reach on this device, measured when asked (bench.py: roofline.peak_measured) ------------
__global__ __launch_bounds__(256) void stream_fill_kernel(uint4* __restrict__ p, size_t n, uint32_t v) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t step = (size_t)gridDim.x * 256;
    const uint4 w = make_uint4(v, v + 1u, v + 2u, v + 3u);
    for (; i < n; i += step) p[i] = w;
}
// The yardstick kernels of pmx_measure_hbm.  Round 6: the one-load-per-iteration grid-stride copy of round 5 got
// 4.9 - 5.1 TB/s where the guide measures 6.29 for a float4 copy; tools/ubench/hbm_probe.hip tried the shapes (16 B per lane
// throughout; 4 GiB; one box): copy, a block on a contiguous chunk, four loads in flight per lane, non-temporal: 5.50 - 5.55 TB/s
// (grid-stride, one load: 4.89; hipMemcpyAsync: 5.10); read, eight loads in flight, 65536 blocks: 6.31 (four, 16384 blocks: 5.99);
// fill, 16384 blocks: 5.86 - 5.93.  These are the shapes below; the copy does not reach the guide's figure on this pool's boxes,
// the read does.
typedef unsigned int pmx_v4u __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stream_copy_kernel(const pmx_v4u* __restrict__ src, pmx_v4u* __restrict__ dst, size_t n) {
    const size_t per = (n + gridDim.x - 1) / gridDim.x;
    const size_t b0 = (size_t)blockIdx.x * per, b1 = b0 + per < n ? b0 + per : n;
    size_t i = b0 + threadIdx.x;
    for (; i + 3 * 256 < b1; i += 4 * 256) {
        pmx_v4u v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(src + i + u * 256);
#pragma unroll
        for (int u = 0; u < 4; ++u) __builtin_nontemporal_store(v[u], dst + i + u * 256);
    }
    for (; i < b1; i += 256) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void stream_read_kernel(const pmx_v4u* __restrict__ p, size_t n, uint32_t* __restrict__ sink) {
    const size_t step = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
    for (; i + 7 * step < n; i += 8 * step) {
        pmx_v4u v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[i + u * step];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n; i += step) acc += p[i].x;
    if (acc == 0x9e3779b9u) sink[0] = acc;  // (keeps the loads alive)
}

extern "C" int pmx_release_caches(pmx_ctx* ctx, size_t* free_bytes, size_t* total_bytes) {
    PMX_CHECK(ctx, PMX_ERR_ARG, "pmx_release_caches: no context");
    PMX_HIP(hipSetDevice(ctx->device));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->aux_stream) PMX_HIP(hipStreamSynchronize(ctx->aux_stream));
    pmx_pool_free(ctx, ctx->scratch);  // (the SGM accumulator: pmx_need_scratch brings it back)
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
    pmx_pool_release(ctx);
    if (ctx->fam_halo) {  // pmx_fam_prepare allocates and zeroes a new one, epochs start over
        PMX_HIP(hipFree(ctx->fam_halo));
        ctx->fam_halo = nullptr;
        ctx->fam_halo_bytes = 0;
        ctx->fam_epoch = 0;
    }
    size_t f = 0, t = 0;
    PMX_HIP(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return PMX_OK;
}

extern "C" int pmx_measure_hbm(pmx_ctx* ctx, size_t bytes, double* read_gbs, double* write_gbs, double* copy_gbs) {
    PMX_CHECK(ctx && bytes >= ((size_t)1 << 20), PMX_ERR_ARG, "pmx_measure_hbm: a context and at least 1 MB");
    PMX_HIP(hipSetDevice(ctx->device));
    bytes &= ~(size_t)15;
    void *a = nullptr, *b = nullptr;
    PMX_HIP(hipMalloc(&a, bytes));
    if (hipMalloc(&b, bytes) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(a);
        PMX_CHECK(false, PMX_ERR_HIP, "pmx_measure_hbm: no room for two buffers of %zu bytes", bytes);
    }
    if (!ctx->probe_sink && hipMalloc(&ctx->probe_sink, 64) != hipSuccess) {
        (void)hipGetLastError();
        ctx->probe_sink = nullptr;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const size_t n = bytes / 16;
    double best[3] = {0, 0, 0};
    for (int rep = 0; rep < 4; ++rep) {  // (the first pass faults the pages in and is not counted)
        for (int kind = 0; kind < 3; ++kind) {
            (void)hipEventRecord(e0, ctx->stream);
            if (kind == 0) hipLaunchKernelGGL(stream_fill_kernel, dim3(16384), dim3(256), 0, ctx->stream, (uint4*)a, n, (uint32_t)rep);
            else if (kind == 1) hipLaunchKernelGGL(stream_read_kernel, dim3(65536), dim3(256), 0, ctx->stream, (const pmx_v4u*)a, n, (uint32_t*)ctx->probe_sink);
            else hipLaunchKernelGGL(stream_copy_kernel, dim3(16384), dim3(256), 0, ctx->stream, (const pmx_v4u*)a, (pmx_v4u*)b, n);
            (void)hipEventRecord(e1, ctx->stream);
            float ms = 0.f;
            if (hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && rep > 0 && ms > 0.f) {
                const double gbs = (double)bytes * (kind == 2 ? 2.0 : 1.0) / (ms * 1e-3) / 1e9;
                if (gbs > best[kind]) best[kind] = gbs;
            }
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(a);
    (void)hipFree(b);
    PMX_HIP(hipGetLastError());
    if (write_gbs) *write_gbs = best[0];
    if (read_gbs) *read_gbs = best[1];
    if (copy_gbs) *copy_gbs = best[2];
    return PMX_OK;
}

static hipError_t placed_alloc(pmx_ctx* ctx, void** out, size_t bytes) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return hipErrorUnknown;
    int trials = ctx->placement_trials;
    while (trials > 1 && (size_t)trials * bytes > free_b / 2) --trials;  // never hold more than half of what is free
    std::vector<std::pair<float, void*>> cand;
    for (int t = 0; t < trials; ++t) {
        void* p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) {
            (void)hipGetLastError();
            break;
        }
        const float ms = placement_probe_ms(ctx, p, bytes);
        cand.emplace_back(ms, p);
        float lo = ms, hi = ms;
        for (auto& c : cand) { lo = c.first < lo ? c.first : lo; hi = c.first > hi ? c.first : hi; }
        if (ms > 0.f && cand.size() > 1 && ms <= lo * 1.01f && hi > lo * 1.04f) break;  // this one is of the fast kind: stop looking
    }
    if (cand.empty()) return hipErrorOutOfMemory;
    size_t best = 0;
    for (size_t i = 1; i < cand.size(); ++i)
        if (cand[i].first > 0.f && (cand[best].first <= 0.f || cand[i].first < cand[best].first)) best = i;
    for (size_t i = 0; i < cand.size(); ++i)
        if (i != best) (void)hipFree(cand[i].second);
    *out = cand[best].second;
    return hipSuccess;
}

// ---- caching allocator (see pmx_ctx) -------------------------------------------------------------------------
hipError_t pmx_pool_alloc(pmx_ctx* ctx, void** p, size_t bytes) {
    *p = nullptr;
    // smallest cached block that fits without wasting more than a quarter
    int best = -1;
    for (int i = 0; i < (int)ctx->pool_free.size(); ++i) {
        const size_t sz = ctx->pool_free[i].first;
        if (sz >= bytes && sz <= bytes + bytes / 4 && (best < 0 || sz < ctx->pool_free[best].first)) best = i;
    }
    if (best >= 0) {
        *p = ctx->pool_free[best].second;
        ctx->pool_live[*p] = ctx->pool_free[best].first;
        ctx->pool_free_bytes -= ctx->pool_free[best].first;
        ctx->pool_free.erase(ctx->pool_free.begin() + best);
        return hipSuccess;
    }
    if (ctx->placement_trials > 1 && bytes >= ((size_t)256 << 20)) {
        hipError_t pe = placed_alloc(ctx, p, bytes);
        if (pe == hipSuccess) {
            ctx->pool_live[*p] = bytes;
            return hipSuccess;
        }
        (void)hipGetLastError();  // fall through to the plain path
        *p = nullptr;
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {  // give the cache back to the driver and retry once
        (void)hipGetLastError();
        hipStreamSynchronize(ctx->stream);
        pmx_pool_release(ctx);
        e = hipMalloc(p, bytes);
    }
    if (e == hipSuccess) ctx->pool_live[*p] = bytes;
    return e;
}

void pmx_pool_free(pmx_ctx* ctx, void* p) {
    if (!p) return;
    auto it = ctx->pool_live.find(p);
    if (it == ctx->pool_live.end()) {  // not ours (allocated before the pool existed)
        hipFree(p);
        return;
    }
    const size_t sz = it->second;
    ctx->pool_live.erase(it);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) total_b = 0;
    // keep at most a third of the device in the cache
    if (total_b && ctx->pool_free_bytes + sz > total_b / 3) {
        hipStreamSynchronize(ctx->stream);
        hipFree(p);
        return;
    }
    ctx->pool_free.emplace_back(sz, p);
    ctx->pool_free_bytes += sz;
}

void pmx_pool_release(pmx_ctx* ctx) {
    for (auto& b : ctx->pool_free) hipFree(b.second);
    ctx->pool_free.clear();
    ctx->pool_free_bytes = 0;
}

int pmx_need_scratch(pmx_ctx* ctx, size_t bytes) {
    if (ctx->scratch_bytes >= bytes) return PMX_OK;
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    pmx_pool_free(ctx, ctx->scratch);
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
    PMX_HIP(pmx_pool_alloc(ctx, (void**)&ctx->scratch, bytes));
    ctx->scratch_bytes = bytes;
    return PMX_OK;
}

int pmx_need_small(pmx_ctx* ctx, size_t bytes) {
    if (ctx->small_bytes >= bytes) return PMX_OK;
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->small) PMX_HIP(hipFree(ctx->small));
    ctx->small = nullptr;
    ctx->small_bytes = 0;
    PMX_HIP(hipMalloc(&ctx->small, bytes));
    ctx->small_bytes = bytes;
    return PMX_OK;
}

extern "C" int pmx_set_images(pmx_ctx* ctx, const float* left, const float* right, int H, int W, int subpix) {
    return pmx_set_images_fingerprinted(ctx, left, right, H, W, subpix, nullptr, nullptr);
}

extern "C" int pmx_set_images_fingerprinted(pmx_ctx* ctx, const float* left, const float* right, int H, int W, int subpix,
                                            uint64_t* fp_left, uint64_t* fp_right) {
    PMX_CHECK(ctx && left && right, PMX_ERR_ARG, "pmx_set_images: null argument");
    PMX_CHECK(H > 0 && W > 1, PMX_ERR_ARG, "pmx_set_images: bad shape %dx%d", H, W);
    PMX_CHECK(subpix == 1 || subpix == 2 || subpix == 4, PMX_ERR_ARG,
              "pmx_set_images: subpix must be 1, 2 or 4 (matching_cost.py:70), got %d", subpix);
    PMX_HIP(hipSetDevice(ctx->device));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    size_t n = (size_t)H * W;
    // a stream of pairs of one shape (the production case) keeps its device buffers: no hipFree / hipMalloc per pair
    const bool same_shape = ctx->left && ctx->H == H && ctx->W == W && ctx->subpix == subpix;
    if (same_shape) {
        free_pair_extras(ctx);
    } else {
        free_images(ctx);
        ctx->H = H; ctx->W = W; ctx->subpix = subpix;
        // all of the pair's buffers or none: a context whose allocation failed half way (out of memory) holds no pair - the next call
        // with the same shape must not take the "same shape, buffers kept" way onto buffers that are not there
        auto alloc_all = [&]() -> int {
            PMX_HIP(img_alloc(ctx, &ctx->left, n));
            PMX_HIP(img_alloc(ctx, &ctx->right[0], n));
            for (int k = 1; k < subpix; ++k) PMX_HIP(img_alloc(ctx, &ctx->right[k], (size_t)H * (W - 1)));
            PMX_HIP(hipMalloc((void**)&ctx->disp, n * sizeof(float)));
            PMX_HIP(hipMalloc((void**)&ctx->itp, n * sizeof(float)));
            PMX_HIP(hipMalloc((void**)&ctx->validity, n * sizeof(int64_t)));
            PMX_HIP(hipMalloc(&ctx->near, n * 16));
            return PMX_OK;
        };
        if (int rca = alloc_all()) {
            free_images(ctx);
            ctx->H = ctx->W = 0;
            ctx->disp_ready = false;
            return rca;
        }
    }
    ctx->near_owner = nullptr;
    ctx->near2_owner = nullptr;
    ctx->disp_ready = false;
    // The caller's arrays are pageable.  Pairs up to kStageMax bytes go through a pinned staging buffer: a few host threads copy
    // them in (faster than the runtime's own single-threaded staging), the DMA is queued and the call returns - the transfer
    // overlaps whatever the host does next (the machine: sizing the cost volume from the disparity grids).
    constexpr size_t kStageMax = 256u << 20;
    const size_t img_bytes = n * sizeof(float);
    const bool staged = 2 * img_bytes <= kStageMax;
    if (staged) {
        if (ctx->stage_cap < 2 * img_bytes) {
            if (ctx->stage_host) PMX_HIP(hipHostFree(ctx->stage_host));
            ctx->stage_host = nullptr;
            ctx->stage_cap = 0;
            PMX_HIP(hipHostMalloc((void**)&ctx->stage_host, 2 * img_bytes, hipHostMallocDefault));
            ctx->stage_cap = 2 * img_bytes;
        }
        // (the stream was drained above: the previous pair's transfer has left the buffer)
        // left image staged -> its transfer is queued and runs while the right image is staged
        const uint64_t fl = fingerprint_copy(ctx->stage_host, left, img_bytes);  // (pmx_host_fingerprint, taken in the copying pass)
        PMX_HIP(hipMemcpyAsync(ctx->left, ctx->stage_host, img_bytes, hipMemcpyHostToDevice, ctx->stream));
        const uint64_t fr = fingerprint_copy(ctx->stage_host + img_bytes, right, img_bytes);
        PMX_HIP(hipMemcpyAsync(ctx->right[0], ctx->stage_host + img_bytes, img_bytes, hipMemcpyHostToDevice, ctx->stream));
        if (fp_left) *fp_left = fl;
        if (fp_right) *fp_right = fr;
    } else {
        if (fp_left) *fp_left = fingerprint_copy(nullptr, left, img_bytes);
        if (fp_right) *fp_right = fingerprint_copy(nullptr, right, img_bytes);
        PMX_HIP(hipMemcpyAsync(ctx->left, left, img_bytes, hipMemcpyHostToDevice, ctx->stream));
        PMX_HIP(hipMemcpyAsync(ctx->right[0], right, img_bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    for (int k = 1; k < subpix; ++k) {
        int rc = pmx_launch_shift_right(ctx, ctx->right[0], H, W, subpix, k, ctx->right[k]);
        if (rc) return rc;
    }
    PMX_HIP(hipMemsetAsync(ctx->near, 0xff, n * 16, ctx->stream));
    PMX_HIP(hipMemsetAsync(ctx->validity, 0, n * sizeof(int64_t), ctx->stream));
    if (!staged) PMX_HIP(hipStreamSynchronize(ctx->stream));  // host buffers may be released by the caller
    return PMX_OK;
}

// The right-side volume of a cross-checked run is computed from the SAME two images in the other order
// (state_machine.py:311-331, matching_cost_run): both are in HBM already, the pair is swapped where it is.
extern "C" int pmx_swap_images(pmx_ctx* ctx) {
    PMX_CHECK(ctx && ctx->left, PMX_ERR_STATE, "pmx_swap_images: call pmx_set_images first");
    PMX_HIP(hipSetDevice(ctx->device));
    std::swap(ctx->left, ctx->right[0]);
    std::swap(ctx->msk_left, ctx->msk_right);
    std::swap(ctx->bad_left, ctx->bad_right);
    for (int k = 1; k < ctx->subpix; ++k) {  // the sub-pixel images of the NEW right image (img_tools.py:713-752, order 1)
        int rc = pmx_launch_shift_right(ctx, ctx->right[0], ctx->H, ctx->W, ctx->subpix, k, ctx->right[k]);
        if (rc) return rc;
    }
    if (ctx->grid_min || ctx->grid_max) {  // per-pixel ranges belong to the side they were set for
        PMX_HIP(hipStreamSynchronize(ctx->stream));
        hipFree(ctx->grid_min); ctx->grid_min = nullptr;
        hipFree(ctx->grid_max); ctx->grid_max = nullptr;
    }
    const size_t n = (size_t)ctx->H * ctx->W;
    ctx->near_owner = nullptr;
    ctx->near2_owner = nullptr;
    ctx->disp_ready = false;
    PMX_HIP(hipMemsetAsync(ctx->near, 0xff, n * 16, ctx->stream));
    PMX_HIP(hipMemsetAsync(ctx->validity, 0, n * sizeof(int64_t), ctx->stream));
    return PMX_OK;
}

extern "C" int pmx_set_shifted_right(pmx_ctx* ctx, int k, const float* shifted) {
    PMX_CHECK(ctx && ctx->left && shifted, PMX_ERR_STATE, "pmx_set_shifted_right: call pmx_set_images first");
    PMX_CHECK(k >= 1 && k < ctx->subpix, PMX_ERR_ARG, "pmx_set_shifted_right: image %d of a pair with subpix %d", k, ctx->subpix);
    PMX_HIP(hipSetDevice(ctx->device));
    PMX_HIP(hipMemcpyAsync(ctx->right[k], shifted, (size_t)ctx->H * (ctx->W - 1) * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    ctx->bad_win = 0;
    return PMX_OK;
}

extern "C" int pmx_set_masks(pmx_ctx* ctx, const int16_t* msk_left, const int16_t* msk_right, int valid_value,
                             int nodata_value) {
    PMX_CHECK(ctx && ctx->left, PMX_ERR_STATE, "pmx_set_masks: call pmx_set_images first");
    PMX_HIP(hipSetDevice(ctx->device));
    if (!msk_left && !msk_right && !ctx->msk_left && !ctx->msk_right) {  // no masks before, none now: nothing to wait for
        ctx->bad_win = 0;
        ctx->valid_value = valid_value;
        ctx->nodata_value = nodata_value;
        return PMX_OK;
    }
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    size_t n = (size_t)ctx->H * ctx->W;
    hipFree(ctx->msk_left); ctx->msk_left = nullptr;
    hipFree(ctx->msk_right); ctx->msk_right = nullptr;
    hipFree(ctx->bad_left); ctx->bad_left = nullptr;
    hipFree(ctx->bad_right); ctx->bad_right = nullptr;
    ctx->bad_win = 0;
    ctx->valid_value = valid_value;
    ctx->nodata_value = nodata_value;
    if (msk_left) {
        PMX_HIP(hipMalloc((void**)&ctx->msk_left, n * sizeof(int16_t)));
        PMX_HIP(hipMalloc((void**)&ctx->bad_left, n));
        PMX_HIP(hipMemcpy(ctx->msk_left, msk_left, n * sizeof(int16_t), hipMemcpyHostToDevice));
    }
    if (msk_right) {
        PMX_HIP(hipMalloc((void**)&ctx->msk_right, n * sizeof(int16_t)));
        PMX_HIP(hipMalloc((void**)&ctx->bad_right, n));
        PMX_HIP(hipMemcpy(ctx->msk_right, msk_right, n * sizeof(int16_t), hipMemcpyHostToDevice));
    }
    return PMX_OK;
}

int pmx_update_bad_masks(pmx_ctx* ctx, int win) {
    if (ctx->bad_win == win) return PMX_OK;
    if (ctx->msk_left) {
        int rc = pmx_launch_mask_dilate(ctx, ctx->msk_left, ctx->H, ctx->W, win, ctx->valid_value, ctx->nodata_value,
                                        ctx->bad_left);
        if (rc) return rc;
    }
    if (ctx->msk_right) {
        int rc = pmx_launch_mask_dilate(ctx, ctx->msk_right, ctx->H, ctx->W, win, ctx->valid_value, ctx->nodata_value,
                                        ctx->bad_right);
        if (rc) return rc;
    }
    ctx->bad_win = win;
    return PMX_OK;
}

extern "C" int pmx_set_disparity_grids(pmx_ctx* ctx, const double* disp_min, const double* disp_max) {
    PMX_CHECK(ctx && ctx->left, PMX_ERR_STATE, "pmx_set_disparity_grids: call pmx_set_images first");
    PMX_CHECK((disp_min == nullptr) == (disp_max == nullptr), PMX_ERR_ARG,
              "pmx_set_disparity_grids: give both grids or neither");
    PMX_HIP(hipSetDevice(ctx->device));
    if (!disp_min && !ctx->grid_min && !ctx->grid_max) return PMX_OK;  // no grids before, none now: nothing to wait for
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    hipFree(ctx->grid_min); ctx->grid_min = nullptr;
    hipFree(ctx->grid_max); ctx->grid_max = nullptr;
    if (disp_min) {
        size_t n = (size_t)ctx->H * ctx->W * sizeof(double);
        PMX_HIP(hipMalloc((void**)&ctx->grid_min, n));
        PMX_HIP(hipMalloc((void**)&ctx->grid_max, n));
        PMX_HIP(hipMemcpy(ctx->grid_min, disp_min, n, hipMemcpyHostToDevice));
        PMX_HIP(hipMemcpy(ctx->grid_max, disp_max, n, hipMemcpyHostToDevice));
    }
    return PMX_OK;
}

// ---- cost volumes ---------------------------------------------------------------------------
extern "C" pmx_cv* pmx_cv_alloc(pmx_ctx* ctx, int D, int d0) {
    if (!ctx || !ctx->left) {
        pmx_set_error("pmx_cv_alloc: call pmx_set_images first");
        return nullptr;
    }
    if (D <= 0) {
        pmx_set_error("pmx_cv_alloc: D must be positive, got %d", D);
        return nullptr;
    }
    hipSetDevice(ctx->device);
    pmx_cv* cv = new pmx_cv();
    cv->ctx = ctx;
    cv->H = ctx->H; cv->W = ctx->W; cv->D = D; cv->d0 = d0; cv->subpix = ctx->subpix;
    cv->bytes = cv->cells() * sizeof(float) + 256;  // tail pad: wide per-lane loads of the last pixel stay in bounds
    // allocate_cost_volume's NaN fill - and, on the lazy path, the float32 storage itself (2.2 GB at C3, 52 GB at C5) - is
    // deferred until something needs it: the integer fast path never does
    cv->repr = PMX_REPR_ALL_NAN;
    if (!ctx->lazy && pmx_cv_materialize(ctx, cv) != PMX_OK) {
        pmx_pool_free(ctx, cv->data);
        delete cv;
        return nullptr;
    }
    return cv;
}

// the float32 storage of a handle, allocated on first need
int pmx_cv_ensure_data(pmx_ctx* ctx, pmx_cv* cv) {
    if (cv->data) return PMX_OK;
    PMX_HIP(hipSetDevice(ctx->device));
    hipError_t e = pmx_pool_alloc(ctx, (void**)&cv->data, cv->bytes);
    if (e != hipSuccess) {
        cv->data = nullptr;
        pmx_set_error("cost volume: hipMalloc(%zu bytes) failed: %s", cv->bytes, hipGetErrorString(e));
        (void)hipGetLastError();  // (reported: the runtime's sticky "last error" must not fail the next call's launch check)
        return PMX_ERR_HIP;
    }
    return PMX_OK;
}

extern "C" void pmx_cv_free(pmx_ctx* ctx, pmx_cv* cv) {
    if (!cv) return;
    if (ctx) pmx_near_forget(ctx, cv);
    if (ctx) {
        hipSetDevice(ctx->device);
        pmx_pool_free(ctx, cv->data);
        pmx_pool_free(ctx, cv->spart);
        pmx_pool_free(ctx, cv->spart2);
        pmx_pool_free(ctx, cv->codes);
        pmx_pool_free(ctx, cv->ldir);
        pmx_pool_free(ctx, cv->cost8);
        pmx_pool_free(ctx, cv->range);
        pmx_pool_free(ctx, cv->missing);
    } else {
        hipFree(cv->data);
        hipFree(cv->spart);
        hipFree(cv->spart2);
        hipFree(cv->codes);
        hipFree(cv->ldir);
    }
    delete cv;
}

// Bring a handle back to a plain float32 [H][W][D] volume, whatever exact form it is held in.
int pmx_cv_materialize(pmx_ctx* ctx, pmx_cv* cv) {
    if (cv->repr == PMX_REPR_FLOAT && cv->data) return PMX_OK;
    if (cv->repr == PMX_REPR_SGM_UP_PENDING) return pmx_sgm_finish_pending(ctx, cv, nullptr);  // the optimised volume is wanted after all
    if (int rc = pmx_cv_ensure_data(ctx, cv)) return rc;
    switch (cv->repr) {
        case PMX_REPR_FLOAT: return PMX_OK;
        case PMX_REPR_ALL_NAN: {
            int rc = pmx_launch_fill_nan(ctx, cv->data, cv->cells());
            if (rc == PMX_OK) cv->repr = PMX_REPR_FLOAT;
            return rc;
        }
        case PMX_REPR_CENSUS_DEFERRED: {
            int rc = pmx_launch_census_costs(ctx, cv);  // sets FLOAT
            if (rc == PMX_OK && cv->has_range) rc = pmx_launch_range_nan(ctx, cv);  // cv_masked had run on the codes
            return rc;
        }
        case PMX_REPR_SGM_U8X8: {
            int rc = pmx_launch_sum8_to_float(ctx, cv);
            if (rc == PMX_OK) cv->repr = PMX_REPR_FLOAT;
            return rc;
        }
    }
    pmx_set_error("pmx_cv_materialize: corrupt handle");
    return PMX_ERR_STATE;
}

extern "C" int pmx_set_placement_trials(pmx_ctx* ctx, int trials) {
    PMX_CHECK(ctx && trials >= 1 && trials <= 8, PMX_ERR_ARG, "pmx_set_placement_trials: 1 <= trials <= 8");
    ctx->placement_trials = trials;
    return PMX_OK;
}

extern "C" int pmx_set_lazy(pmx_ctx* ctx, int enabled) {
    PMX_CHECK(ctx, PMX_ERR_ARG, "pmx_set_lazy: null context");
    ctx->lazy = enabled != 0;
    return PMX_OK;
}

extern "C" int pmx_cv_fill_nan(pmx_ctx* ctx, pmx_cv* cv) {
    PMX_CHECK(ctx && cv, PMX_ERR_ARG, "pmx_cv_fill_nan: null argument");
    cv->repr = PMX_REPR_ALL_NAN;
    cv->nonneg = true;
    return ctx->lazy ? PMX_OK : pmx_cv_materialize(ctx, cv);
}

extern "C" int pmx_cv_upload(pmx_ctx* ctx, pmx_cv* cv, const float* host) {
    PMX_CHECK(ctx && cv && host, PMX_ERR_ARG, "pmx_cv_upload: null argument");
    PMX_HIP(hipSetDevice(ctx->device));
    if (int rc0 = pmx_cv_ensure_data(ctx, cv)) return rc0;
    PMX_HIP(hipMemcpyAsync(cv->data, host, cv->cells() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    cv->repr = PMX_REPR_FLOAT;
    cv->nonneg = false;  // whatever the caller computed
    return PMX_OK;
}

extern "C" int pmx_cv_download(pmx_ctx* ctx, pmx_cv* cv, float* host) {
    PMX_CHECK(ctx && cv && host, PMX_ERR_ARG, "pmx_cv_download: null argument");
    PMX_HIP(hipSetDevice(ctx->device));
    {
        int rc = pmx_cv_materialize(ctx, cv);
        if (rc) return rc;
    }
    PMX_HIP(hipMemcpyAsync(host, cv->data, cv->cells() * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return pmx_check_async_error(ctx, "pmx_cv_download");
}

extern "C" int pmx_cv_download_rows(pmx_ctx* ctx, pmx_cv* cv, int row_lo, int row_hi, float* host) {
    PMX_CHECK(ctx && cv && host, PMX_ERR_ARG, "pmx_cv_download_rows: null argument");
    PMX_CHECK(0 <= row_lo && row_lo < row_hi && row_hi <= cv->H, PMX_ERR_ARG, "pmx_cv_download_rows: rows [%d, %d) of %d", row_lo, row_hi, cv->H);
    PMX_HIP(hipSetDevice(ctx->device));
    {
        int rc = pmx_cv_materialize(ctx, cv);
        if (rc) return rc;
    }
    const size_t row = (size_t)cv->W * cv->D;
    PMX_HIP(hipMemcpyAsync(host, cv->data + (size_t)row_lo * row, (size_t)(row_hi - row_lo) * row * sizeof(float), hipMemcpyDeviceToHost,
                           ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return pmx_check_async_error(ctx, "pmx_cv_download_rows");
}

extern "C" int pmx_cv_dims(const pmx_cv* cv, int* H, int* W, int* D, int* d0, int* subpix) {
    PMX_CHECK(cv, PMX_ERR_ARG, "pmx_cv_dims: null cv");
    if (H) *H = cv->H;
    if (W) *W = cv->W;
    if (D) *D = cv->D;
    if (d0) *d0 = cv->d0;
    if (subpix) *subpix = cv->subpix;
    return PMX_OK;
}

static int check_cv(pmx_ctx* ctx, const pmx_cv* cv, const char* who) {
    PMX_CHECK(ctx && cv, PMX_ERR_ARG, "%s: null argument", who);
    PMX_CHECK(cv->ctx == ctx, PMX_ERR_ARG, "%s: cost volume belongs to another context", who);
    PMX_CHECK(ctx->left && cv->H == ctx->H && cv->W == ctx->W && cv->subpix == ctx->subpix, PMX_ERR_STATE,
              "%s: cost volume does not match the resident images", who);
    PMX_HIP(hipSetDevice(ctx->device));
    return PMX_OK;
}

static bool census_window_ok(int win) { return win == 3 || win == 5 || win == 7 || win == 9 || win == 11 || win == 13; }

extern "C" int pmx_census(pmx_ctx* ctx, pmx_cv* cv, int win) {
    int rc = check_cv(ctx, cv, "pmx_census");
    if (rc) return rc;
    PMX_CHECK(census_window_ok(win), PMX_ERR_ARG, "pmx_census: window_size must be in (3,5,7,9,11,13) (census.py:68), got %d", win);
    // the Hamming costs can stay implicit (codes only) while nothing needs the float volume; only
    // worth it where the fused SGM path can consume them
    const int nw = (win * win + 31) / 32;
    // (a right mask makes cv_masked a per-cell pattern: float volume; grids and a left mask are intervals per pixel and stay lazy)
    const bool defer = ctx->lazy && cv->subpix == 1 && nw <= 6 && cv->D < 320 && abs(cv->d0) + cv->D <= 1024 / nw - 32 && !ctx->msk_right;
    cv->has_range = false;
    cv->nonneg = true;
    if (!defer) {
        rc = pmx_cv_ensure_data(ctx, cv);
        if (rc) return rc;
    }
    return pmx_launch_census(ctx, cv, win, defer);
}

extern "C" int pmx_sad_ssd(pmx_ctx* ctx, pmx_cv* cv, int win, int squared) {
    int rc = check_cv(ctx, cv, "pmx_sad_ssd");
    if (rc) return rc;
    PMX_CHECK(win > 0 && (win & 1), PMX_ERR_ARG, "pmx_sad_ssd: window_size must be odd and > 0 (sad_ssd.py:69), got %d", win);
    rc = pmx_cv_ensure_data(ctx, cv);
    if (rc) return rc;
    rc = pmx_launch_sad_ssd(ctx, cv, win, squared);
    cv->nonneg = true;
    if (rc == PMX_OK) cv->repr = PMX_REPR_FLOAT;
    return rc;
}

extern "C" int pmx_zncc(pmx_ctx* ctx, pmx_cv* cv, int win) {
    int rc = check_cv(ctx, cv, "pmx_zncc");
    if (rc) return rc;
    PMX_CHECK(win > 0 && (win & 1), PMX_ERR_ARG, "pmx_zncc: window_size must be odd and > 0, got %d", win);
    rc = pmx_cv_ensure_data(ctx, cv);
    if (rc) return rc;
    rc = pmx_launch_zncc(ctx, cv, win);
    cv->nonneg = false;
    if (rc == PMX_OK) cv->repr = PMX_REPR_FLOAT;
    return rc;
}

extern "C" int pmx_cv_masked(pmx_ctx* ctx, pmx_cv* cv, int win) {
    int rc = check_cv(ctx, cv, "pmx_cv_masked");
    if (rc) return rc;
    if (!ctx->msk_left && !ctx->msk_right && !ctx->grid_min) return PMX_OK;  // nothing to inject: NaN pattern is complete
    rc = pmx_update_bad_masks(ctx, win);
    if (rc) return rc;
    if (cv->repr == PMX_REPR_CENSUS_DEFERRED && !ctx->msk_right && win == cv->win)
        return pmx_launch_build_range(ctx, cv);  // stays on the integer path: the kernels take the valid interval of each pixel
    rc = pmx_cv_materialize(ctx, cv);
    if (rc) return rc;
    return pmx_launch_cv_masked(ctx, cv, win);
}

extern "C" int pmx_cv_scale_pixels(pmx_ctx* ctx, pmx_cv* cv, const float* weights) {
    int rc = check_cv(ctx, cv, "pmx_cv_scale_pixels");
    if (rc) return rc;
    PMX_CHECK(weights, PMX_ERR_ARG, "pmx_cv_scale_pixels: null weights");
    const size_t n = (size_t)cv->H * cv->W;
    rc = pmx_need_small(ctx, n * sizeof(float));
    if (rc) return rc;
    rc = pmx_cv_materialize(ctx, cv);
    if (rc) return rc;
    PMX_HIP(hipMemcpyAsync(ctx->small, weights, n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    cv->nonneg = false;  // the weights are the caller's
    rc = pmx_launch_scale_pixels(ctx, cv, (const float*)ctx->small);
    if (rc) return rc;
    PMX_HIP(hipStreamSynchronize(ctx->stream));  // the host weights may be released by the caller
    return PMX_OK;
}

static int launch_missing(pmx_ctx* ctx, const pmx_cv* cv, uint8_t* dev_out) {
    if (cv->repr == PMX_REPR_CENSUS_DEFERRED || cv->repr == PMX_REPR_SGM_U8X8)
        return pmx_launch_census_nan_pixels(ctx, cv, dev_out);  // NaN pattern = census geometry (x cv_masked's snapshot)
    int rc = pmx_cv_materialize(ctx, const_cast<pmx_cv*>(cv));
    if (rc) return rc;
    return pmx_launch_nan_pixels(ctx, cv, dev_out);
}

extern "C" int pmx_nan_pixels(pmx_ctx* ctx, const pmx_cv* cv, uint8_t* host_out) {
    int rc = check_cv(ctx, cv, "pmx_nan_pixels");
    if (rc) return rc;
    PMX_CHECK(host_out, PMX_ERR_ARG, "pmx_nan_pixels: null output");
    size_t n = (size_t)cv->H * cv->W;
    rc = pmx_need_small(ctx, n);
    if (rc) return rc;
    rc = launch_missing(ctx, cv, (uint8_t*)ctx->small);
    if (rc) return rc;
    PMX_HIP(hipMemcpyAsync(host_out, ctx->small, n, hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return PMX_OK;
}

extern "C" int pmx_cv_mark_missing(pmx_ctx* ctx, pmx_cv* cv) {
    int rc = check_cv(ctx, cv, "pmx_cv_mark_missing");
    if (rc) return rc;
    const size_t n = (size_t)cv->H * cv->W;
    if (cv->missing_bytes < n) {
        pmx_pool_free(ctx, cv->missing);
        cv->missing = nullptr;
        cv->missing_bytes = 0;
        cv->has_missing = false;
        PMX_HIP(pmx_pool_alloc(ctx, (void**)&cv->missing, n));
        cv->missing_bytes = n;
    }
    rc = launch_missing(ctx, cv, cv->missing);
    if (rc) return rc;
    cv->has_missing = true;
    return PMX_OK;
}

extern "C" int pmx_cv_get_missing(pmx_ctx* ctx, const pmx_cv* cv, uint8_t* host_out) {
    int rc = check_cv(ctx, cv, "pmx_cv_get_missing");
    if (rc) return rc;
    PMX_CHECK(host_out, PMX_ERR_ARG, "pmx_cv_get_missing: null output");
    PMX_CHECK(cv->has_missing, PMX_ERR_STATE, "pmx_cv_get_missing: pmx_cv_mark_missing has not been called on this volume");
    PMX_HIP(hipMemcpyAsync(host_out, cv->missing, (size_t)cv->H * cv->W, hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return pmx_check_async_error(ctx, "pmx_cv_get_missing");
}

extern "C" int pmx_compose_validity(pmx_ctx* ctx, const int64_t* base, int base_rows, const pmx_cv* missing_of, int border) {
    PMX_CHECK(ctx && ctx->left, PMX_ERR_STATE, "pmx_compose_validity: call pmx_set_images first");
    PMX_CHECK(base && (base_rows == 1 || base_rows == ctx->H), PMX_ERR_ARG,
              "pmx_compose_validity: the base is one line of W flags (base_rows 1) or a full H x W map (base_rows H)");
    PMX_CHECK(border >= 0, PMX_ERR_ARG, "pmx_compose_validity: negative border");
    if (missing_of) {
        int rc = check_cv(ctx, missing_of, "pmx_compose_validity");
        if (rc) return rc;
        PMX_CHECK(missing_of->has_missing, PMX_ERR_STATE, "pmx_compose_validity: pmx_cv_mark_missing has not been called on the volume");
    }
    PMX_HIP(hipSetDevice(ctx->device));
    const uint8_t* miss = missing_of ? missing_of->missing : nullptr;
    if (base_rows == 1) {
        // the line goes through a pinned buffer: the copy is queued, nothing waits for the kernels in front of it
        const size_t n = (size_t)ctx->W;
        constexpr int kSlots = pmx_ctx::kLineSlots;
        if (ctx->line_cap < n) {
            PMX_HIP(hipStreamSynchronize(ctx->stream));
            if (ctx->line_host) PMX_HIP(hipHostFree(ctx->line_host));
            if (ctx->line_dev) PMX_HIP(hipFree(ctx->line_dev));
            ctx->line_host = ctx->line_dev = nullptr;
            ctx->line_cap = 0;
            PMX_HIP(hipHostMalloc((void**)&ctx->line_host, kSlots * n * sizeof(int64_t), hipHostMallocDefault));
            PMX_HIP(hipMalloc((void**)&ctx->line_dev, kSlots * n * sizeof(int64_t)));
            ctx->line_cap = n;
        }
        const int slot = ctx->line_next;
        ctx->line_next = (slot + 1) % kSlots;
        if (!ctx->line_ev[slot]) PMX_HIP(hipEventCreateWithFlags(&ctx->line_ev[slot], hipEventDisableTiming));
        else PMX_HIP(hipEventSynchronize(ctx->line_ev[slot]));  // this slot's previous line was used kSlots calls ago
        int64_t* hl = ctx->line_host + (size_t)slot * ctx->line_cap;
        int64_t* dl = ctx->line_dev + (size_t)slot * ctx->line_cap;
        memcpy(hl, base, n * sizeof(int64_t));
        PMX_HIP(hipMemcpyAsync(dl, hl, n * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
        int rc = pmx_launch_compose_validity(ctx, dl, 1, miss, border);
        PMX_HIP(hipEventRecord(ctx->line_ev[slot], ctx->stream));  // (behind the kernel that reads the device line)
        return rc;
    }
    // a full map (input masks took part): uploaded into the validity buffer itself, then finished in place
    const size_t bytes = (size_t)ctx->H * ctx->W * sizeof(int64_t);
    PMX_HIP(hipMemcpyAsync(ctx->validity, base, bytes, hipMemcpyHostToDevice, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));  // the caller's buffer is pageable: it may change as soon as we return
    return pmx_launch_compose_validity(ctx, ctx->validity, ctx->H, miss, border);
}

extern "C" pmx_cv* pmx_reverse_cost_volume(pmx_ctx* ctx, const pmx_cv* left_cv, int min_disp) {
    if (check_cv(ctx, left_cv, "pmx_reverse_cost_volume")) return nullptr;
    if (pmx_cv_materialize(ctx, const_cast<pmx_cv*>(left_cv))) return nullptr;
    pmx_cv* out = pmx_cv_alloc(ctx, left_cv->D, min_disp);
    if (!out) return nullptr;
    if (pmx_cv_ensure_data(ctx, out) != PMX_OK) {
        pmx_cv_free(ctx, out);
        return nullptr;
    }
    out->repr = PMX_REPR_FLOAT;  // the kernel writes every cell
    out->nonneg = left_cv->nonneg;
    if (pmx_launch_reverse(ctx, left_cv, min_disp, out) != PMX_OK) {
        pmx_cv_free(ctx, out);
        return nullptr;
    }
    return out;
}

extern "C" int pmx_cbca(pmx_ctx* ctx, pmx_cv* cv, int offset, float intensity, int distance) {
    int rc = check_cv(ctx, cv, "pmx_cbca");
    if (rc) return rc;
    PMX_CHECK(offset >= 0 && distance >= 1 && intensity > 0.f, PMX_ERR_ARG,
              "pmx_cbca: need offset >= 0, cbca_distance >= 1, cbca_intensity > 0 (cbca.py:59-82)");
    PMX_CHECK(distance <= 32, PMX_ERR_UNSUPPORTED, "pmx_cbca: cbca_distance > 32 not supported");
    // census costs still implicit (codes only): pass H computes them on the fly and the float volume first exists as the
    // aggregated one
    bool census_src = pmx_cbca_can_fuse_census(ctx, cv, offset, distance);
    if (census_src && pmx_need_scratch(ctx, cv->cells() * sizeof(float) + 256) != PMX_OK) {
        (void)hipGetLastError();  // no room for the E_h volume: the in-place kernels of pmx_launch_cbca work on a materialised volume
        census_src = false;
    }
    rc = census_src ? pmx_cv_ensure_data(ctx, cv) : pmx_cv_materialize(ctx, cv);
    if (rc) return rc;
    rc = pmx_launch_cbca(ctx, cv, offset, intensity, distance, census_src);
    if (rc == PMX_OK && census_src) cv->repr = PMX_REPR_FLOAT;
    return rc;
}

extern "C" int pmx_cross_support(pmx_ctx* ctx, int side, int offset, float intensity, int distance, int16_t* host_out) {
    PMX_CHECK(ctx && ctx->left && host_out, PMX_ERR_ARG, "pmx_cross_support: bad argument");
    PMX_CHECK(side >= 0 && side <= ctx->subpix, PMX_ERR_ARG, "pmx_cross_support: side out of range");
    PMX_HIP(hipSetDevice(ctx->device));
    int Wd = side <= 1 ? ctx->W : ctx->W - 1;
    size_t n = (size_t)(ctx->H - 2 * offset) * (Wd - 2 * offset) * 4 * sizeof(int16_t);
    int16_t* dev = nullptr;
    PMX_HIP(hipMalloc((void**)&dev, n));
    int rc = pmx_launch_cross_support(ctx, side, offset, intensity, distance, dev);
    if (rc == PMX_OK) {
        hipError_t e = hipMemcpyAsync(host_out, dev, n, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            pmx_set_error("pmx_cross_support: copy failed: %s", hipGetErrorString(e));
            rc = PMX_ERR_HIP;
        }
    }
    hipFree(dev);
    return rc;
}

extern "C" int pmx_sgm(pmx_ctx* ctx, pmx_cv* cv, float P1, float P2, int is_max, float invalid_cost, int overcounting) {
    int rc = check_cv(ctx, cv, "pmx_sgm");
    if (rc) return rc;
    PMX_CHECK(P1 > 0.f && P2 > P1, PMX_ERR_ARG, "pmx_sgm: need 0 < P1 < P2 (plugin_libsgm.rst:170-185), got %g %g", P1, P2);
    PMX_CHECK(cv->D <= 512, PMX_ERR_UNSUPPORTED, "pmx_sgm: D = %d > 512 disparities not supported", cv->D);
    cv->nonneg = false;  // (true for costs >= 0, but nothing downstream of SGM asks)
    if (ctx->lazy && ctx->sgm_dir_mask == 0xff && pmx_fused_sgm_eligible(ctx, cv, P1, P2, is_max, invalid_cost, overcounting))
        return pmx_launch_sgm_fused(ctx, cv, P1, P2, invalid_cost);  // integer fast path, bit-identical
    rc = pmx_cv_materialize(ctx, cv);
    if (rc) return rc;
    return pmx_launch_sgm(ctx, cv, P1, P2, is_max, invalid_cost, overcounting);
}

extern "C" int pmx_sgm_p2maps(pmx_ctx* ctx, pmx_cv* cv, float P1, const float* p2maps, int is_max, float invalid_cost, int overcounting) {
    int rc = check_cv(ctx, cv, "pmx_sgm_p2maps");
    if (rc) return rc;
    PMX_CHECK(p2maps, PMX_ERR_ARG, "pmx_sgm_p2maps: null penalty maps");
    PMX_CHECK(P1 > 0.f, PMX_ERR_ARG, "pmx_sgm_p2maps: need P1 > 0, got %g", P1);
    PMX_CHECK(cv->D <= 512, PMX_ERR_UNSUPPORTED, "pmx_sgm_p2maps: D = %d > 512 disparities not supported", cv->D);
    PMX_CHECK(ctx->sgm_dir_mask == 0xff, PMX_ERR_STATE, "pmx_sgm_p2maps: the direction mask of pmx_debug_sgm_directions is set");
    cv->nonneg = false;
    rc = pmx_cv_materialize(ctx, cv);  // float32 kernels only: the integer fast path has the constant penalty in its arithmetic
    if (rc) return rc;
    const size_t n = (size_t)8 * cv->H * cv->W;
    float* dev = nullptr;
    PMX_HIP(pmx_pool_alloc(ctx, (void**)&dev, n * sizeof(float)));
    hipError_t e = hipMemcpyAsync(dev, p2maps, n * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // the caller may release the maps
    if (e != hipSuccess) {
        pmx_pool_free(ctx, dev);
        PMX_HIP(e);
    }
    ctx->sgm_p2maps = dev;
    rc = pmx_launch_sgm(ctx, cv, P1, 0.f, is_max, invalid_cost, overcounting);
    ctx->sgm_p2maps = nullptr;
    pmx_pool_free(ctx, dev);  // stream-ordered reuse: the kernels above are queued on ctx->stream
    return rc;
}

extern "C" int pmx_set_validity(pmx_ctx* ctx, const int64_t* validity) {
    PMX_CHECK(ctx && ctx->left, PMX_ERR_STATE, "pmx_set_validity: call pmx_set_images first");
    PMX_HIP(hipSetDevice(ctx->device));
    size_t n = (size_t)ctx->H * ctx->W * sizeof(int64_t);
    if (validity) {
        PMX_HIP(hipMemcpyAsync(ctx->validity, validity, n, hipMemcpyHostToDevice, ctx->stream));
        PMX_HIP(hipStreamSynchronize(ctx->stream));
    } else {
        PMX_HIP(hipMemsetAsync(ctx->validity, 0, n, ctx->stream));
    }
    return PMX_OK;
}

// ---- the winner cache (ctx->near: what the WTA left of its winner, so that the refinement does not gather it again) belongs to
// ONE volume.  A cross-checked run takes the left and the right volume through WTA and refinement in turn: with one cache the
// left refinement would find the right side's winners in it and gather again (0.9 ms at 2048^2 x 129), so there are two, and the
// one that belongs to the volume at hand is made the active one.
int pmx_near_select(pmx_ctx* ctx, const pmx_cv* cv, bool for_write) {
    if (ctx->near_owner == cv) return PMX_OK;
    auto swap_slots = [&] {
        std::swap(ctx->near, ctx->near2);
        std::swap(ctx->near_owner, ctx->near2_owner);
        std::swap(ctx->near_exact, ctx->near2_exact);
    };
    if (ctx->near2 && ctx->near2_owner == cv) {
        swap_slots();
        return PMX_OK;
    }
    if (!for_write || ctx->near_owner == nullptr) return PMX_OK;  // reading: no cache of this volume; writing: the active one is free
    // the active cache belongs to another volume: write into the other one (taking it from whoever had it)
    if (!ctx->near2) {
        const size_t bytes = (size_t)ctx->H * ctx->W * 16;
        if (hipMalloc(&ctx->near2, bytes) != hipSuccess) {  // no room for a second cache: overwrite the only one, as before
            (void)hipGetLastError();
            ctx->near2 = nullptr;
            return PMX_OK;
        }
        PMX_HIP(hipMemsetAsync(ctx->near2, 0xff, bytes, ctx->stream));
    }
    ctx->near2_owner = nullptr;
    ctx->near2_exact = false;
    swap_slots();
    return PMX_OK;
}

void pmx_near_forget(pmx_ctx* ctx, const pmx_cv* cv) {
    if (ctx->near_owner == cv) ctx->near_owner = nullptr;
    if (ctx->near2_owner == cv) ctx->near2_owner = nullptr;
}

extern "C" int pmx_wta(pmx_ctx* ctx, const pmx_cv* cv, int is_max, float invalid_disparity) {
    int rc = check_cv(ctx, cv, "pmx_wta");
    if (rc) return rc;
    ctx->disp_ready = true;
    ctx->near_exact = ctx->near2_exact = false;
    const bool up_pending = cv->repr == PMX_REPR_SGM_UP_PENDING && (is_max != 0) == (cv->pending.is_max != 0);
    if ((cv->repr == PMX_REPR_SGM_U8X8 && !is_max) || up_pending) {  // the two routes that leave a winner cache
        rc = pmx_near_select(ctx, cv, true);
        if (rc) return rc;
    }
    if (cv->repr == PMX_REPR_SGM_U8X8 && !is_max) return pmx_launch_sum8_wta(ctx, cv, invalid_disparity);
    if (up_pending) {
        // the last SGM pass and the WTA in one kernel: the optimised volume is neither written nor read
        const pmx_fam_wta w = {ctx->disp, (float*)ctx->near, (double)cv->d0, cv->subpix, invalid_disparity};
        rc = pmx_sgm_finish_pending(ctx, const_cast<pmx_cv*>(cv), &w);
        if (rc) return rc;
        ctx->near_owner = cv;
        ctx->near_exact = true;
        return pmx_launch_wta_fixup(ctx, cv);
    }
    rc = pmx_cv_materialize(ctx, const_cast<pmx_cv*>(cv));
    if (rc) return rc;
    return pmx_launch_wta(ctx, cv, is_max, invalid_disparity);
}

extern "C" int pmx_refine(pmx_ctx* ctx, const pmx_cv* cv, int method, int is_max) {
    int rc = check_cv(ctx, cv, "pmx_refine");
    if (rc) return rc;
    PMX_CHECK(method == PMX_REFINE_VFIT || method == PMX_REFINE_QUADRATIC, PMX_ERR_ARG,
              "pmx_refine: unknown refinement method %d", method);
    PMX_CHECK(ctx->disp_ready, PMX_ERR_STATE, "pmx_refine: no disparity map for this pair yet (run pmx_wta or pmx_set_disparity first)");
    rc = pmx_near_select(ctx, cv, false);
    if (rc) return rc;
    if (cv->repr == PMX_REPR_SGM_U8X8 && !is_max) return pmx_launch_sum8_refine(ctx, cv, method);
    if (cv->repr == PMX_REPR_SGM_UP_PENDING && ctx->near_owner == cv && ctx->near_exact) {
        ctx->near_exact = false;  // the refined map is no longer the WTA's
        return pmx_launch_near_refine(ctx, cv, method, is_max);
    }
    rc = pmx_cv_materialize(ctx, const_cast<pmx_cv*>(cv));
    if (rc) return rc;
    return pmx_launch_refine(ctx, cv, method, is_max);
}

extern "C" int pmx_refine_approximate(pmx_ctx* ctx, const pmx_cv* cv_left, int method, int is_max) {
    int rc = check_cv(ctx, cv_left, "pmx_refine_approximate");
    if (rc) return rc;
    PMX_CHECK(method == PMX_REFINE_VFIT || method == PMX_REFINE_QUADRATIC, PMX_ERR_ARG,
              "pmx_refine_approximate: unknown refinement method %d", method);
    PMX_CHECK(ctx->disp_ready, PMX_ERR_STATE, "pmx_refine_approximate: no disparity map resident (pmx_set_disparity with the right map first)");
    rc = pmx_cv_materialize(ctx, const_cast<pmx_cv*>(cv_left));
    if (rc) return rc;
    ctx->near_exact = false;
    ctx->near2_exact = false;
    return pmx_launch_approx_refine(ctx, cv_left, method, is_max);
}

extern "C" int pmx_get_disparity(pmx_ctx* ctx, float* disp, int64_t* validity, float* itp) {
    PMX_CHECK(ctx && ctx->left, PMX_ERR_STATE, "pmx_get_disparity: nothing resident");
    PMX_HIP(hipSetDevice(ctx->device));
    size_t n = (size_t)ctx->H * ctx->W;
    if (disp) PMX_HIP(hipMemcpyAsync(disp, ctx->disp, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    if (validity) PMX_HIP(hipMemcpyAsync(validity, ctx->validity, n * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    if (itp) PMX_HIP(hipMemcpyAsync(itp, ctx->itp, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return pmx_check_async_error(ctx, "pmx_get_disparity");
}

// np.percentile's order statistics of a float32 map, without its sort (the percentile normalisation of the ambiguity measure,
// ambiguity.py:168-184, partitions a 4 Mpx map four times per run: 120 ms of host time at 2048^2)
extern "C" int pmx_order_statistics(pmx_ctx* ctx, const float* values, size_t n, const size_t* ranks, int n_ranks, float* out) {
    PMX_CHECK(ctx && values && ranks && out && n > 0 && n_ranks > 0, PMX_ERR_ARG, "pmx_order_statistics: null or empty argument");
    for (int i = 0; i < n_ranks; ++i) PMX_CHECK(ranks[i] < n, PMX_ERR_ARG, "pmx_order_statistics: rank %zu of %zu values", ranks[i], n);
    PMX_HIP(hipSetDevice(ctx->device));
    int rc = pmx_need_small(ctx, n * sizeof(float) + 2048 * sizeof(uint32_t));
    if (rc) return rc;
    float* dev = (float*)ctx->small;
    uint32_t* dev_hist = (uint32_t*)(dev + n);
    std::vector<uint32_t> host_hist(2048);
    PMX_HIP(hipMemcpyAsync(dev, values, n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    for (int i = 0; i < n_ranks; ++i) {
        rc = pmx_launch_order_statistic(ctx, dev, n, ranks[i], dev_hist, host_hist.data(), out + i);
        if (rc) return rc;
    }
    return PMX_OK;
}

// ---- snapshots of the result maps: a device-side copy that outlives the next step's overwrite and is only brought to the
// host if somebody reads it (the reference's deep copies of 2-D results, e.g. cv["disp_indices"], disparity.py:459) ------------
struct pmx_snap {
    void* dev = nullptr;
    size_t bytes = 0;
};

extern "C" void* pmx_map_snapshot(pmx_ctx* ctx, int which) {
    if (!ctx || !ctx->left || which < 0 || which > 2) {
        pmx_set_error("pmx_map_snapshot: nothing resident, or unknown map %d (0 disparity, 1 validity, 2 interpolated coefficient)", which);
        return nullptr;
    }
    if (hipSetDevice(ctx->device) != hipSuccess) return nullptr;
    const size_t n = (size_t)ctx->H * ctx->W;
    const void* src = which == 0 ? (const void*)ctx->disp : which == 1 ? (const void*)ctx->validity : (const void*)ctx->itp;
    pmx_snap* s = new pmx_snap;
    s->bytes = n * (which == 1 ? sizeof(int64_t) : sizeof(float));
    if (pmx_pool_alloc(ctx, &s->dev, s->bytes) != hipSuccess ||
        hipMemcpyAsync(s->dev, src, s->bytes, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) {
        (void)hipGetLastError();
        pmx_set_error("pmx_map_snapshot: no device memory for %zu bytes", s->bytes);
        pmx_pool_free(ctx, s->dev);
        delete s;
        return nullptr;
    }
    return s;
}

extern "C" void* pmx_map_snapshot_alloc(pmx_ctx* ctx, int which) {  // uninitialised: the output of a step that works on snapshots
    if (!ctx || !ctx->left || which < 0 || which > 2) {
        pmx_set_error("pmx_map_snapshot_alloc: nothing resident, or unknown map %d", which);
        return nullptr;
    }
    if (hipSetDevice(ctx->device) != hipSuccess) return nullptr;
    pmx_snap* s = new pmx_snap;
    s->bytes = (size_t)ctx->H * ctx->W * (which == 1 ? sizeof(int64_t) : sizeof(float));
    if (pmx_pool_alloc(ctx, &s->dev, s->bytes) != hipSuccess) {
        (void)hipGetLastError();
        pmx_set_error("pmx_map_snapshot_alloc: no device memory for %zu bytes", s->bytes);
        delete s;
        return nullptr;
    }
    return s;
}

static int check_snap(pmx_ctx* ctx, const void* snapshot, size_t elem, const char* who, const char* what) {
    PMX_CHECK(ctx && ctx->left, PMX_ERR_STATE, "%s: nothing resident", who);
    PMX_CHECK(snapshot, PMX_ERR_ARG, "%s: null %s", who, what);
    PMX_CHECK(((const pmx_snap*)snapshot)->bytes == (size_t)ctx->H * ctx->W * elem, PMX_ERR_ARG,
              "%s: %s is not a %zu-byte-per-pixel map of the resident %dx%d pair", who, what, elem, ctx->H, ctx->W);
    return PMX_OK;
}
#define PMX_SNAP(p) (((const pmx_snap*)(p))->dev)

// The engine's disparity / validity maps become what two snapshots hold (device-to-device): the step that follows (pmx_refine,
// pmx_get_disparity ...) works on them as if the WTA had just produced them.
extern "C" int pmx_maps_restore(pmx_ctx* ctx, const void* disp_snapshot, const void* validity_snapshot) {
    if (int rc = check_snap(ctx, disp_snapshot, sizeof(float), "pmx_maps_restore", "disparity snapshot")) return rc;
    if (int rc = check_snap(ctx, validity_snapshot, sizeof(int64_t), "pmx_maps_restore", "validity snapshot")) return rc;
    PMX_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)ctx->H * ctx->W;
    PMX_HIP(hipMemcpyAsync(ctx->disp, PMX_SNAP(disp_snapshot), n * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    PMX_HIP(hipMemcpyAsync(ctx->validity, PMX_SNAP(validity_snapshot), n * sizeof(int64_t), hipMemcpyDeviceToDevice, ctx->stream));
    ctx->disp_ready = true;
    ctx->near_exact = ctx->near2_exact = false;  // a winner cache describes its WTA's map; whether this is that map nobody knows
    return PMX_OK;
}

// median.py:94-131 / validation.py:226-371 / criteria.py:325-353 on snapshots: nothing crosses PCIe, nothing waits
extern "C" int pmx_median_filter_maps(pmx_ctx* ctx, const void* disp_snapshot, const void* validity_snapshot, int filter_size,
                                      void* out_disp_snapshot) {
    if (int rc = check_snap(ctx, disp_snapshot, sizeof(float), "pmx_median_filter_maps", "disparity snapshot")) return rc;
    if (int rc = check_snap(ctx, validity_snapshot, sizeof(int64_t), "pmx_median_filter_maps", "validity snapshot")) return rc;
    if (int rc = check_snap(ctx, out_disp_snapshot, sizeof(float), "pmx_median_filter_maps", "output snapshot")) return rc;
    PMX_CHECK(out_disp_snapshot != disp_snapshot, PMX_ERR_ARG, "pmx_median_filter_maps: the filter does not work in place");
    PMX_CHECK(filter_size >= 1 && (filter_size & 1) && filter_size <= 15, PMX_ERR_ARG,
              "pmx_median_filter_maps: filter_size must be odd, >= 1 (median.py:86) and <= 15, got %d", filter_size);
    PMX_CHECK(filter_size <= ctx->H && filter_size <= ctx->W, PMX_ERR_ARG, "pmx_median_filter_maps: filter_size %d exceeds the %dx%d map",
              filter_size, ctx->H, ctx->W);
    PMX_HIP(hipSetDevice(ctx->device));
    return pmx_launch_median_disparity(ctx, (const float*)PMX_SNAP(disp_snapshot), (const int64_t*)PMX_SNAP(validity_snapshot), ctx->H,
                                       ctx->W, filter_size, (float*)PMX_SNAP(out_disp_snapshot));
}

extern "C" int pmx_cross_checking_maps(pmx_ctx* ctx, const void* disp_left, void* validity_left, const void* disp_right, int dmin,
                                       int dmax, double threshold, void* conf_out) {
    if (int rc = check_snap(ctx, disp_left, sizeof(float), "pmx_cross_checking_maps", "left disparity snapshot")) return rc;
    if (int rc = check_snap(ctx, validity_left, sizeof(int64_t), "pmx_cross_checking_maps", "left validity snapshot")) return rc;
    if (int rc = check_snap(ctx, disp_right, sizeof(float), "pmx_cross_checking_maps", "right disparity snapshot")) return rc;
    if (int rc = check_snap(ctx, conf_out, sizeof(float), "pmx_cross_checking_maps", "confidence snapshot")) return rc;
    PMX_CHECK(dmin <= dmax, PMX_ERR_ARG, "pmx_cross_checking_maps: bad interval [%d,%d]", dmin, dmax);
    PMX_HIP(hipSetDevice(ctx->device));
    return pmx_launch_cross_checking(ctx, (const float*)PMX_SNAP(disp_left), (int64_t*)PMX_SNAP(validity_left),
                                     (const float*)PMX_SNAP(disp_right), ctx->H, ctx->W, dmin, dmax, threshold, (float*)PMX_SNAP(conf_out));
}

extern "C" int pmx_validity_frame_map(pmx_ctx* ctx, void* validity_snapshot, int border) {
    if (int rc = check_snap(ctx, validity_snapshot, sizeof(int64_t), "pmx_validity_frame_map", "validity snapshot")) return rc;
    PMX_CHECK(border >= 0, PMX_ERR_ARG, "pmx_validity_frame_map: negative border");
    PMX_HIP(hipSetDevice(ctx->device));
    return pmx_launch_compose_validity_into(ctx, (const int64_t*)PMX_SNAP(validity_snapshot), ctx->H, nullptr, border,
                                            (int64_t*)PMX_SNAP(validity_snapshot));
}

extern "C" int pmx_map_snapshot_read(pmx_ctx* ctx, const void* snapshot, void* host_out) {
    PMX_CHECK(ctx && snapshot && host_out, PMX_ERR_ARG, "pmx_map_snapshot_read: null argument");
    const pmx_snap* s = (const pmx_snap*)snapshot;
    PMX_HIP(hipSetDevice(ctx->device));
    PMX_HIP(hipMemcpyAsync(host_out, s->dev, s->bytes, hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return pmx_check_async_error(ctx, "pmx_map_snapshot_read");
}

extern "C" void pmx_map_snapshot_free(pmx_ctx* ctx, void* snapshot) {
    if (!snapshot) return;
    pmx_snap* s = (pmx_snap*)snapshot;
    if (ctx) {
        hipSetDevice(ctx->device);
        pmx_pool_free(ctx, s->dev);
    }
    delete s;
}

extern "C" int pmx_set_disparity(pmx_ctx* ctx, const float* disp, const int64_t* validity) {
    PMX_CHECK(ctx && ctx->left, PMX_ERR_STATE, "pmx_set_disparity: call pmx_set_images first");
    PMX_HIP(hipSetDevice(ctx->device));
    size_t n = (size_t)ctx->H * ctx->W;
    if (disp) {
        PMX_HIP(hipMemcpyAsync(ctx->disp, disp, n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        ctx->disp_ready = true;
        ctx->near_exact = ctx->near2_exact = false;  // an edited map: the winner cache of a fused WTA no longer describes it
    }
    if (validity) PMX_HIP(hipMemcpyAsync(ctx->validity, validity, n * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return PMX_OK;
}

extern "C" int pmx_wta_minkey(pmx_ctx* ctx, const pmx_cv* cv, int is_max, int global_index_offset, uint64_t* dev_keys) {
    int rc = check_cv(ctx, cv, "pmx_wta_minkey");
    if (rc) return rc;
    PMX_CHECK(dev_keys, PMX_ERR_ARG, "pmx_wta_minkey: null key buffer");
    rc = pmx_cv_materialize(ctx, const_cast<pmx_cv*>(cv));
    if (rc) return rc;
    return pmx_launch_minkey(ctx, cv, is_max, global_index_offset, dev_keys);
}

extern "C" int pmx_wta_from_keys(pmx_ctx* ctx, const uint64_t* dev_keys, double d0_global, int subpix,
                                 float invalid_disparity) {
    PMX_CHECK(ctx && ctx->left && dev_keys, PMX_ERR_ARG, "pmx_wta_from_keys: bad argument");
    PMX_HIP(hipSetDevice(ctx->device));
    ctx->disp_ready = true;
    ctx->near_exact = false;
    return pmx_launch_from_keys(ctx, dev_keys, d0_global, subpix, invalid_disparity);
}

extern "C" int pmx_debug_small_division(pmx_ctx* ctx, unsigned* mismatches) {
    PMX_CHECK(ctx && mismatches, PMX_ERR_ARG, "pmx_debug_small_division: null argument");
    PMX_HIP(hipSetDevice(ctx->device));
    return pmx_launch_small_division_check(ctx, mismatches);
}

extern "C" int pmx_debug_path_costs(pmx_ctx* ctx, const pmx_cv* cv, uint8_t* host_out, size_t host_bytes, int* Dp, int* gl,
                                    int* kpl) {
    PMX_CHECK(ctx && cv, PMX_ERR_ARG, "pmx_debug_path_costs: null argument");
    PMX_CHECK(cv->repr == PMX_REPR_SGM_U8X8 && cv->ldir && cv->nvol == 8, PMX_ERR_STATE,
              "pmx_debug_path_costs: volume is not in the fused SGM representation with eight path volumes");
    if (Dp) *Dp = cv->Dp;
    if (gl) *gl = cv->gl;
    if (kpl) *kpl = cv->kpl;
    size_t n = (size_t)8 * cv->H * cv->W * cv->Dp;
    if (!host_out) return PMX_OK;
    PMX_CHECK(host_bytes >= n, PMX_ERR_ARG, "pmx_debug_path_costs: host buffer too small (%zu < %zu)", host_bytes, n);
    PMX_HIP(hipSetDevice(ctx->device));
    for (int k = 0; k < 8; ++k)  // the device volumes are dstride apart, the host copy is dense
        PMX_HIP(hipMemcpyAsync(host_out + (size_t)k * (n / 8), cv->ldir + (size_t)k * cv->dstride, n / 8, hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return PMX_OK;
}

// ---- measurement ----------------------------------------------------------------------------
extern "C" int pmx_set_profiling(pmx_ctx* ctx, int enabled) {
    PMX_CHECK(ctx, PMX_ERR_ARG, "pmx_set_profiling: null context");
    ctx->profiling = enabled != 0;
    return PMX_OK;
}

static int fold_stage(pmx_ctx* ctx, int stage) {
    pmx_stage_rec& s = ctx->stages[stage];
    for (size_t i = 0; i + 1 < s.ev.size(); i += 2) {
        float ms = 0.f;
        PMX_HIP(hipEventElapsedTime(&ms, s.ev[i], s.ev[i + 1]));
        s.total_ms += ms;
        s.launches += 1;
        hipEventDestroy(s.ev[i]);
        hipEventDestroy(s.ev[i + 1]);
    }
    s.ev.clear();
    return PMX_OK;
}

extern "C" int pmx_reset_stage_times(pmx_ctx* ctx) {
    PMX_CHECK(ctx, PMX_ERR_ARG, "pmx_reset_stage_times: null context");
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < PMX_STAGE_COUNT; ++i) {
        int rc = fold_stage(ctx, i);
        if (rc) return rc;
        ctx->stages[i].total_ms = 0.0;
        ctx->stages[i].launches = 0;
    }
    return PMX_OK;
}

extern "C" int pmx_stage_time(pmx_ctx* ctx, int stage, double* total_ms, int* launches) {
    PMX_CHECK(ctx && stage >= 0 && stage < PMX_STAGE_COUNT, PMX_ERR_ARG, "pmx_stage_time: bad argument");
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    int rc = fold_stage(ctx, stage);
    if (rc) return rc;
    if (total_ms) *total_ms = ctx->stages[stage].total_ms;
    if (launches) *launches = ctx->stages[stage].launches;
    return PMX_OK;
}


// ---- validation on the device ---------------------------------------------------------
extern "C" int pmx_cross_checking(pmx_ctx* ctx, const float* disp_left, int64_t* validity_left, const float* disp_right, int H, int W,
                                  int dmin, int dmax, double threshold, float* conf_out) {
    PMX_CHECK(ctx && disp_left && validity_left && disp_right && conf_out, PMX_ERR_ARG, "pmx_cross_checking: null argument");
    PMX_CHECK(H > 0 && W > 0 && dmin <= dmax, PMX_ERR_ARG, "pmx_cross_checking: bad shape %dx%d or interval [%d,%d]", H, W, dmin, dmax);
    PMX_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)H * W;
    int rc = pmx_need_small(ctx, n * (4 + 4 + 4 + 8));
    if (rc) return rc;
    char* base = (char*)ctx->small;
    int64_t* d_val = (int64_t*)base;
    float* d_dl = (float*)(base + n * 8);
    float* d_dr = d_dl + n;
    float* d_conf = d_dr + n;
    PMX_HIP(hipMemcpyAsync(d_val, validity_left, n * 8, hipMemcpyHostToDevice, ctx->stream));
    PMX_HIP(hipMemcpyAsync(d_dl, disp_left, n * 4, hipMemcpyHostToDevice, ctx->stream));
    PMX_HIP(hipMemcpyAsync(d_dr, disp_right, n * 4, hipMemcpyHostToDevice, ctx->stream));
    rc = pmx_launch_cross_checking(ctx, d_dl, d_val, d_dr, H, W, dmin, dmax, threshold, d_conf);
    if (rc) return rc;
    PMX_HIP(hipMemcpyAsync(validity_left, d_val, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipMemcpyAsync(conf_out, d_conf, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return PMX_OK;
}

extern "C" int pmx_interpolate_disparity(pmx_ctx* ctx, float* disp, int64_t* validity, int H, int W, const int* passes, int n_passes) {
    PMX_CHECK(ctx && disp && validity && passes, PMX_ERR_ARG, "pmx_interpolate_disparity: null argument");
    PMX_CHECK(H > 0 && W > 0 && n_passes > 0, PMX_ERR_ARG, "pmx_interpolate_disparity: bad shape %dx%d or pass count %d", H, W, n_passes);
    for (int k = 0; k < n_passes; ++k)
        PMX_CHECK(passes[k] >= PMX_INTERP_OCCLUSION_MC_CNN && passes[k] <= PMX_INTERP_MISMATCH_SGM, PMX_ERR_ARG,
                  "pmx_interpolate_disparity: unknown pass %d", passes[k]);
    PMX_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)H * W;
    int rc = pmx_need_small(ctx, n * 24);
    if (rc) return rc;
    int64_t* d_val[2] = {(int64_t*)ctx->small, (int64_t*)ctx->small + n};
    float* d_disp[2] = {(float*)(d_val[1] + n), (float*)(d_val[1] + n) + n};
    PMX_HIP(hipMemcpyAsync(d_val[0], validity, n * 8, hipMemcpyHostToDevice, ctx->stream));
    PMX_HIP(hipMemcpyAsync(d_disp[0], disp, n * 4, hipMemcpyHostToDevice, ctx->stream));
    int cur = 0;
    for (int k = 0; k < n_passes; ++k, cur ^= 1) {  // every pass reads the previous pass' maps only
        rc = pmx_launch_interpolate_disparity(ctx, passes[k], d_disp[cur], d_val[cur], H, W, d_disp[cur ^ 1], d_val[cur ^ 1]);
        if (rc) return rc;
    }
    PMX_HIP(hipMemcpyAsync(validity, d_val[cur], n * 8, hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipMemcpyAsync(disp, d_disp[cur], n * 4, hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return PMX_OK;
}

extern "C" int pmx_reverse_disp_range(pmx_ctx* ctx, const float* left_min, const float* left_max, int H, int W, int global_min,
                                      int global_max, float* right_min, float* right_max) {
    PMX_CHECK(ctx && left_min && left_max && right_min && right_max, PMX_ERR_ARG, "pmx_reverse_disp_range: null argument");
    PMX_CHECK(H > 0 && W > 0, PMX_ERR_ARG, "pmx_reverse_disp_range: bad shape %dx%d", H, W);
    PMX_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)H * W;
    int rc = pmx_need_small(ctx, n * 16);
    if (rc) return rc;
    float* d = (float*)ctx->small;
    PMX_HIP(hipMemcpyAsync(d, left_min, n * 4, hipMemcpyHostToDevice, ctx->stream));
    PMX_HIP(hipMemcpyAsync(d + n, left_max, n * 4, hipMemcpyHostToDevice, ctx->stream));
    rc = pmx_launch_reverse_disp_range(ctx, d, d + n, H, W, global_min, global_max, d + 2 * n, d + 3 * n);
    if (rc) return rc;
    PMX_HIP(hipMemcpyAsync(right_min, d + 2 * n, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipMemcpyAsync(right_max, d + 3 * n, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return PMX_OK;
}


// ---- disparity filters on the device -------------------------------------------------------
extern "C" int pmx_median_filter_disparity(pmx_ctx* ctx, float* disp, const int64_t* validity, int H, int W, int filter_size) {
    PMX_CHECK(ctx && disp && validity, PMX_ERR_ARG, "pmx_median_filter_disparity: null argument");
    PMX_CHECK(H > 0 && W > 0, PMX_ERR_ARG, "pmx_median_filter_disparity: bad shape %dx%d", H, W);
    PMX_CHECK(filter_size >= 1 && (filter_size & 1) && filter_size <= 15, PMX_ERR_ARG,
              "pmx_median_filter_disparity: filter_size must be odd, >= 1 (median.py:86) and <= 15, got %d", filter_size);
    PMX_CHECK(filter_size <= H && filter_size <= W, PMX_ERR_ARG, "pmx_median_filter_disparity: filter_size %d exceeds the %dx%d map",
              filter_size, H, W);
    PMX_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)H * W;
    int rc = pmx_need_small(ctx, n * (8 + 4 + 4));
    if (rc) return rc;
    char* base = (char*)ctx->small;
    int64_t* d_val = (int64_t*)base;
    float* d_in = (float*)(base + n * 8);
    float* d_out = d_in + n;
    PMX_HIP(hipMemcpyAsync(d_val, validity, n * 8, hipMemcpyHostToDevice, ctx->stream));
    PMX_HIP(hipMemcpyAsync(d_in, disp, n * 4, hipMemcpyHostToDevice, ctx->stream));
    rc = pmx_launch_median_disparity(ctx, d_in, d_val, H, W, filter_size, d_out);
    if (rc) return rc;
    PMX_HIP(hipMemcpyAsync(disp, d_out, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return PMX_OK;
}

extern "C" int pmx_bilateral_filter_disparity(pmx_ctx* ctx, float* disp, const int64_t* validity, int H, int W, double sigma_color,
                                              double sigma_space) {
    PMX_CHECK(ctx && disp && validity, PMX_ERR_ARG, "pmx_bilateral_filter_disparity: null argument");
    PMX_CHECK(H > 0 && W > 0, PMX_ERR_ARG, "pmx_bilateral_filter_disparity: bad shape %dx%d", H, W);
    PMX_CHECK(sigma_color > 0 && sigma_space > 0, PMX_ERR_ARG, "pmx_bilateral_filter_disparity: sigmas must be > 0 (bilateral.py:92-93)");
    PMX_HIP(hipSetDevice(ctx->device));
    int win = (int)(3 * sigma_space + 1);  // bilateral.py:171
    if (win > H) win = H;
    if (win > W) win = W;
    const size_t n = (size_t)H * W, ng = (size_t)win * win;
    std::vector<double> gs(ng);
    const double norm = sigma_space * sqrt(2 * 3.14159265358979323846);
    for (int i = 0; i < win; ++i)
        for (int j = 0; j < win; ++j) {
            const double dist = sqrt((double)((i - win / 2) * (i - win / 2) + (j - win / 2) * (j - win / 2)));  // bilateral.py:203-214
            gs[(size_t)i * win + j] = exp(-((dist / sigma_space) * (dist / sigma_space)) * 0.5) / norm;
        }
    int rc = pmx_need_small(ctx, n * (8 + 4 + 4) + ng * 8);
    if (rc) return rc;
    char* base = (char*)ctx->small;
    int64_t* d_val = (int64_t*)base;
    double* d_gs = (double*)(base + n * 8);
    float* d_in = (float*)(base + n * 8 + ng * 8);
    float* d_out = d_in + n;
    PMX_HIP(hipMemcpyAsync(d_val, validity, n * 8, hipMemcpyHostToDevice, ctx->stream));
    PMX_HIP(hipMemcpyAsync(d_gs, gs.data(), ng * 8, hipMemcpyHostToDevice, ctx->stream));
    PMX_HIP(hipMemcpyAsync(d_in, disp, n * 4, hipMemcpyHostToDevice, ctx->stream));
    rc = pmx_launch_bilateral_disparity(ctx, d_in, d_val, H, W, win, d_gs, sigma_color, d_out);
    if (rc) return rc;
    PMX_HIP(hipMemcpyAsync(disp, d_out, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));  // (also keeps `gs` alive until its copy has left)
    return PMX_OK;
}


extern "C" int pmx_denoise_disparity(pmx_ctx* ctx, float* disp, const int64_t* validity, const float* color, const float* grad_row,
                                     const float* grad_col, int H, int W, int filter_size, double sigma_euclidian, double sigma_color,
                                     double sigma_planar) {
    PMX_CHECK(ctx && disp && validity && color && grad_row && grad_col, PMX_ERR_ARG, "pmx_denoise_disparity: null argument");
    PMX_CHECK(H > 0 && W > 0, PMX_ERR_ARG, "pmx_denoise_disparity: bad shape %dx%d", H, W);
    PMX_CHECK(filter_size > 0 && (filter_size & 1), PMX_ERR_ARG,
              "pmx_denoise_disparity: filter_size must be odd and > 0 (disparity_denoiser.py:80,120), got %d", filter_size);
    PMX_CHECK(sigma_euclidian > 0 && sigma_color > 0 && sigma_planar > 0, PMX_ERR_ARG,
              "pmx_denoise_disparity: sigmas must be > 0 (disparity_denoiser.py:121-123)");
    PMX_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)H * W, ng = (size_t)filter_size * filter_size;
    std::vector<double> ge(ng);
    const int o = filter_size / 2;
    for (int i = -o; i <= o; ++i)
        for (int j = -o; j <= o; ++j) {
            const double e = sqrt((double)(i * i + j * j)) / sigma_euclidian;  // disparity_denoiser.py:280-283, :38-48
            ge[(size_t)(i + o) * filter_size + (j + o)] = exp(-(e * e) / 2.0);
        }
    int rc = pmx_need_small(ctx, n * (8 + 5 * 4) + ng * 8);
    if (rc) return rc;
    char* base = (char*)ctx->small;
    int64_t* d_val = (int64_t*)base;
    double* d_ge = (double*)(base + n * 8);
    float* d_in = (float*)(base + n * 8 + ng * 8);
    float *d_col = d_in + n, *d_g0 = d_in + 2 * n, *d_g1 = d_in + 3 * n, *d_out = d_in + 4 * n;
    PMX_HIP(hipMemcpyAsync(d_val, validity, n * 8, hipMemcpyHostToDevice, ctx->stream));
    PMX_HIP(hipMemcpyAsync(d_ge, ge.data(), ng * 8, hipMemcpyHostToDevice, ctx->stream));
    PMX_HIP(hipMemcpyAsync(d_in, disp, n * 4, hipMemcpyHostToDevice, ctx->stream));
    PMX_HIP(hipMemcpyAsync(d_col, color, n * 4, hipMemcpyHostToDevice, ctx->stream));
    PMX_HIP(hipMemcpyAsync(d_g0, grad_row, n * 4, hipMemcpyHostToDevice, ctx->stream));
    PMX_HIP(hipMemcpyAsync(d_g1, grad_col, n * 4, hipMemcpyHostToDevice, ctx->stream));
    rc = pmx_launch_denoise_disparity(ctx, d_in, d_val, d_col, d_g0, d_g1, H, W, filter_size, d_ge, sigma_color, sigma_planar, d_out);
    if (rc) return rc;
    PMX_HIP(hipMemcpyAsync(disp, d_out, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));  // (also keeps `ge` alive until its copy has left)
    return PMX_OK;
}

// ---- multiscale -------------------------------------------------------------------------------------
extern "C" int pmx_interpolate_nodata(pmx_ctx* ctx, const float* img, const int32_t* msk, int H, int W, int invalid_bits,
                                      int filled_value, float* out_img, int32_t* out_msk) {
    PMX_CHECK(ctx && img && msk && out_img && out_msk, PMX_ERR_ARG, "pmx_interpolate_nodata: null argument");
    PMX_CHECK(H > 0 && W > 0, PMX_ERR_ARG, "pmx_interpolate_nodata: bad shape %dx%d", H, W);
    PMX_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)H * W;
    int rc = pmx_need_small(ctx, n * 16);
    if (rc) return rc;
    float* d_img = (float*)ctx->small;
    int* d_msk = (int*)(d_img + n);
    float* d_oimg = (float*)(d_msk + n);
    int* d_omsk = (int*)(d_oimg + n);
    PMX_HIP(hipMemcpyAsync(d_img, img, n * 4, hipMemcpyHostToDevice, ctx->stream));
    PMX_HIP(hipMemcpyAsync(d_msk, msk, n * 4, hipMemcpyHostToDevice, ctx->stream));
    rc = pmx_launch_interpolate_nodata(ctx, d_img, d_msk, H, W, invalid_bits, filled_value, d_oimg, d_omsk);
    if (rc) return rc;
    PMX_HIP(hipMemcpyAsync(out_img, d_oimg, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipMemcpyAsync(out_msk, d_omsk, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return PMX_OK;
}

extern "C" int pmx_disparity_range(pmx_ctx* ctx, const float* disp, const int64_t* validity, int H, int W, int window_size, int marge,
                                   int global_min, int global_max, float* range_min, float* range_max) {
    PMX_CHECK(ctx && disp && validity && range_min && range_max, PMX_ERR_ARG, "pmx_disparity_range: null argument");
    PMX_CHECK(H > 0 && W > 0 && window_size >= 1 && marge >= 0, PMX_ERR_ARG, "pmx_disparity_range: bad shape / window / marge");
    PMX_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)H * W;
    int rc = pmx_need_small(ctx, n * (8 + 4 + 4 + 4));
    if (rc) return rc;
    char* base = (char*)ctx->small;
    int64_t* d_val = (int64_t*)base;
    float* d_disp = (float*)(base + n * 8);
    float* d_lo = d_disp + n;
    float* d_hi = d_lo + n;
    PMX_HIP(hipMemcpyAsync(d_val, validity, n * 8, hipMemcpyHostToDevice, ctx->stream));
    PMX_HIP(hipMemcpyAsync(d_disp, disp, n * 4, hipMemcpyHostToDevice, ctx->stream));
    rc = pmx_launch_disparity_range(ctx, d_disp, d_val, H, W, window_size, marge, global_min, global_max, d_lo, d_hi);
    if (rc) return rc;
    PMX_HIP(hipMemcpyAsync(range_min, d_lo, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipMemcpyAsync(range_max, d_hi, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return PMX_OK;
}


// ---- cost-volume confidence ------------------------------------------------------------------------
extern "C" int pmx_ambiguity(pmx_ctx* ctx, pmx_cv* cv, const float* etas, int nbr_etas, const int64_t* grid_min, const int64_t* grid_max,
                             int negate, float* ambiguity_out) {
    int rc = check_cv(ctx, cv, "pmx_ambiguity");
    if (rc) return rc;
    PMX_CHECK(etas && ambiguity_out && !grid_min == !grid_max, PMX_ERR_ARG, "pmx_ambiguity: null argument (grids: both or neither)");
    PMX_CHECK(nbr_etas > 0 && nbr_etas <= 1024, PMX_ERR_ARG, "pmx_ambiguity: nbr_etas must be in 1..1024, got %d", nbr_etas);
    for (int i = 1; i < nbr_etas; ++i)  // the kernel counts admitted etas with a binary search: exact only for sorted thresholds
        PMX_CHECK(etas[i] >= etas[i - 1], PMX_ERR_ARG, "pmx_ambiguity: etas must be ascending (etas[%d] = %g < %g)", i, etas[i], etas[i - 1]);
    rc = pmx_cv_materialize(ctx, cv);  // the measure is defined on the float32 costs
    if (rc) return rc;
    const size_t n = (size_t)cv->H * cv->W;
    rc = pmx_need_small(ctx, n * (8 + 8 + 4) + (size_t)nbr_etas * 4 + 64);
    if (rc) return rc;
    char* base = (char*)ctx->small;
    int64_t* d_gmin = (int64_t*)base;
    int64_t* d_gmax = d_gmin + n;
    float* d_amb = (float*)(d_gmax + n);
    float* d_etas = d_amb + n;
    uint32_t* d_mm = (uint32_t*)(d_etas + nbr_etas);
    if (grid_min) {
        PMX_HIP(hipMemcpyAsync(d_gmin, grid_min, n * 8, hipMemcpyHostToDevice, ctx->stream));
        PMX_HIP(hipMemcpyAsync(d_gmax, grid_max, n * 8, hipMemcpyHostToDevice, ctx->stream));
    } else {
        d_gmin = d_gmax = nullptr;  // the volume's whole disparity range for every pixel
    }
    PMX_HIP(hipMemcpyAsync(d_etas, etas, (size_t)nbr_etas * 4, hipMemcpyHostToDevice, ctx->stream));
    rc = pmx_launch_ambiguity(ctx, cv, d_etas, nbr_etas, d_gmin, d_gmax, negate, d_mm, d_amb);
    if (rc) return rc;
    PMX_HIP(hipMemcpyAsync(ambiguity_out, d_amb, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return PMX_OK;
}

extern "C" int pmx_risk(pmx_ctx* ctx, pmx_cv* cv, const double* etas, int nbr_etas, const int64_t* grid_min, const int64_t* grid_max,
                        int negate, float* risk_max, float* risk_min, float* disp_sup, float* disp_inf) {
    int rc = check_cv(ctx, cv, "pmx_risk");
    if (rc) return rc;
    PMX_CHECK(etas && risk_max && risk_min && disp_sup && disp_inf && !grid_min == !grid_max, PMX_ERR_ARG, "pmx_risk: null argument (grids: both or neither)");
    PMX_CHECK(nbr_etas > 0 && nbr_etas <= 1024, PMX_ERR_ARG, "pmx_risk: nbr_etas must be in 1..1024, got %d", nbr_etas);
    PMX_CHECK(etas[0] >= 0, PMX_ERR_ARG, "pmx_risk: etas must start at >= 0 (the reference's scan is undefined otherwise), got %g", etas[0]);
    for (int i = 1; i < nbr_etas; ++i)
        PMX_CHECK(etas[i] >= etas[i - 1], PMX_ERR_ARG, "pmx_risk: etas must be ascending (etas[%d] = %g < %g)", i, etas[i], etas[i - 1]);
    rc = pmx_cv_materialize(ctx, cv);
    if (rc) return rc;
    const size_t n = (size_t)cv->H * cv->W;
    rc = pmx_need_small(ctx, n * (8 + 8 + 16) + (size_t)nbr_etas * 8 + 64);
    if (rc) return rc;
    char* base = (char*)ctx->small;
    int64_t* d_gmin = (int64_t*)base;
    int64_t* d_gmax = d_gmin + n;
    double* d_etas = (double*)(d_gmax + n);
    float* d_out = (float*)(d_etas + nbr_etas);
    uint32_t* d_mm = (uint32_t*)(d_out + 4 * n);
    if (grid_min) {
        PMX_HIP(hipMemcpyAsync(d_gmin, grid_min, n * 8, hipMemcpyHostToDevice, ctx->stream));
        PMX_HIP(hipMemcpyAsync(d_gmax, grid_max, n * 8, hipMemcpyHostToDevice, ctx->stream));
    } else {
        d_gmin = d_gmax = nullptr;  // the volume's whole disparity range for every pixel
    }
    PMX_HIP(hipMemcpyAsync(d_etas, etas, (size_t)nbr_etas * 8, hipMemcpyHostToDevice, ctx->stream));
    rc = pmx_launch_risk(ctx, cv, d_etas, nbr_etas, d_gmin, d_gmax, negate, d_mm, d_out);
    if (rc) return rc;
    float* outs[4] = {risk_max, risk_min, disp_sup, disp_inf};
    for (int k = 0; k < 4; ++k) PMX_HIP(hipMemcpyAsync(outs[k], d_out + k * n, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return PMX_OK;
}

extern "C" int pmx_interval_bounds(pmx_ctx* ctx, pmx_cv* cv, float possibility_threshold, float type_factor, const int64_t* grid_min,
                                   const int64_t* grid_max, float* interval_inf, float* interval_sup) {
    int rc = check_cv(ctx, cv, "pmx_interval_bounds");
    if (rc) return rc;
    PMX_CHECK(interval_inf && interval_sup && !grid_min == !grid_max, PMX_ERR_ARG, "pmx_interval_bounds: null argument (grids: both or neither)");
    rc = pmx_cv_materialize(ctx, cv);
    if (rc) return rc;
    const size_t n = (size_t)cv->H * cv->W;
    rc = pmx_need_small(ctx, n * (8 + 8 + 8) + 64);
    if (rc) return rc;
    char* base = (char*)ctx->small;
    int64_t* d_gmin = (int64_t*)base;
    int64_t* d_gmax = d_gmin + n;
    float* d_out = (float*)(d_gmax + n);
    uint32_t* d_mm = (uint32_t*)(d_out + 2 * n);
    if (grid_min) {
        PMX_HIP(hipMemcpyAsync(d_gmin, grid_min, n * 8, hipMemcpyHostToDevice, ctx->stream));
        PMX_HIP(hipMemcpyAsync(d_gmax, grid_max, n * 8, hipMemcpyHostToDevice, ctx->stream));
    } else {
        d_gmin = d_gmax = nullptr;  // the volume's whole disparity range for every pixel
    }
    rc = pmx_launch_interval_bounds(ctx, cv, possibility_threshold, type_factor, d_gmin, d_gmax, d_mm, d_out);
    if (rc) return rc;
    PMX_HIP(hipMemcpyAsync(interval_inf, d_out, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipMemcpyAsync(interval_sup, d_out + n, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return PMX_OK;
}
