// k_disparity.hip - winner-takes-all over D, sub-pixel refinement, the packed-key variant used
// for D-sharded multi-GPU WTA, and reverse_cost_volume.  gfx950.
//
// WTA reads the volume once (4 B/cell algorithmic) and writes O(H*W): HBM-bound streaming
// reduction; one wavefront per pixel, 64 disparities per coalesced 256-B load, (value, index)
// lexicographic reduction with wave shuffles so ties resolve to the lowest index like np.argmin /
// np.argmax (disparity/disparity.py:511,548).
#include "pmx_internal.h"

static constexpr int kBlock = 256;
#define MSK_INVALID 0x3C3LL /* constants.py:31 */
#define MSK_STOPPED 0x8LL   /* constants.py:40 */

__device__ __forceinline__ float d_inf() { return __int_as_float(0x7f800000); }
__device__ __forceinline__ float d_nan() { return __int_as_float(0x7fc00000); }

// (v, idx) <- better of (v, idx) and (ov, oidx); "better" = smaller v for min (larger for max), ties
// to the smaller index
template <bool IS_MAX>
__device__ __forceinline__ void take_better(float& v, int& idx, float ov, int oidx) {
    bool better = IS_MAX ? (ov > v) : (ov < v);
    bool tie = (ov == v) && (oidx < idx);
    if (better || tie) { v = ov; idx = oidx; }
}

template <bool IS_MAX>
__device__ __forceinline__ void wave_argbest(const float* __restrict__ p, int D, int lane, float& v, int& idx, int& any) {
    v = IS_MAX ? -d_inf() : d_inf();
    idx = 0x7fffffff;
    any = 0;
    for (int k = lane; k < D; k += 64) {
        float x = p[k];
        if (x == x) {
            any = 1;
        } else {
            x = IS_MAX ? -d_inf() : d_inf();  // disparity.py:434-446
        }
        bool better = IS_MAX ? (x > v) : (x < v);
        if (better || idx == 0x7fffffff) { v = x; idx = k; }
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        float ov = __shfl_xor(v, s);
        int oi = __shfl_xor(idx, s);
        int oa = __shfl_xor(any, s);
        take_better<IS_MAX>(v, idx, ov, oi);
        any |= oa;
    }
}

// orderable(f): unsigned key with the same order as the float (NaN never reaches here)
__device__ __forceinline__ uint32_t orderable_f(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

template <int CTRL>
__device__ __forceinline__ uint64_t dpp64(uint64_t v) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = (uint32_t)__builtin_amdgcn_update_dpp((int)lo, (int)lo, CTRL, 0xf, 0xf, false);
    hi = (uint32_t)__builtin_amdgcn_update_dpp((int)hi, (int)hi, CTRL, 0xf, 0xf, false);
    return ((uint64_t)hi << 32) | lo;
}

// min over each 16-lane row, result in every lane of the row
__device__ __forceinline__ uint64_t row_allmin_u64(uint64_t v) {
    uint64_t o;
    o = dpp64<0x128>(v); v = o < v ? o : v;  // row_ror:8
    o = dpp64<0x124>(v); v = o < v ? o : v;  // row_ror:4
    o = dpp64<0x122>(v); v = o < v ? o : v;  // row_ror:2
    o = dpp64<0x121>(v); v = o < v ? o : v;  // row_ror:1
    return v;
}

// WTA on a float32 volume: four pixels per wavefront (one per 16-lane DPP row), lane `sub` owns NE = 4*NB
// consecutive disparities read with NB 16-byte loads (few wide loads: the texture addresser charges per
// instruction).  (orderable cost, index) is one uint64 key per disparity, so a row min-reduce yields the
// FIRST extremum exactly like np.argmin / np.argmax (disparity.py:511,548); NaN counts as +/-inf (:434-446).
template <bool IS_MAX, int NB>
__global__ __launch_bounds__(kBlock) void wta_kernel(const float* __restrict__ cv, size_t npix, int D, double d0, int subpix,
                                                     float invalid_disparity, float* __restrict__ disp,
                                                     int64_t* __restrict__ validity) {
    constexpr int NE = 4 * NB;
    const int lane = threadIdx.x & 63;
    const int sub = lane & 15, grp = lane >> 4;
    const size_t wave = (size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const size_t nwaves = (size_t)gridDim.x * (kBlock / 64);
    const int d_first = sub * NE;
    const bool lane_active = d_first < D;
    for (size_t quad = wave; quad * 4 < npix; quad += nwaves) {
        const size_t pix = min(quad * 4 + grp, npix - 1);  // surplus rows repeat the last pixel (same values)
        float x[NE];
        __builtin_memcpy(x, cv + pix * (size_t)D + (lane_active ? d_first : 0), sizeof(float) * NE);  // the volume has a tail pad
        uint64_t key = ~0ull;
        bool any = false;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            float v = x[e];
            const bool in = lane_active && (d_first + e < D);
            const bool finite = in && (v == v);
            any = any || finite;
            if (!(v == v)) v = IS_MAX ? -d_inf() : d_inf();
            if (IS_MAX) v = -v;
            if (v == 0.f) v = 0.f;  // -0 and +0 must tie
            const uint64_t kk = ((uint64_t)orderable_f(v) << 32) | (uint32_t)(d_first + e);
            if (in) key = kk < key ? kk : key;
        }
        key = row_allmin_u64(key);
        const unsigned long long bal = __ballot(any);
        const bool row_any = ((bal >> (grp * 16)) & 0xffffull) != 0;
        if (sub == 0) {
            if (!row_any) {
                disp[pix] = invalid_disparity;  // disparity.py:452-455
                int64_t m = validity[pix];
                if ((m & MSK_INVALID) == 0) validity[pix] = MSK_INVALID;  // disparity.py:471-474
            } else {
                disp[pix] = (float)(d0 + (double)(uint32_t)key / (double)subpix);
            }
        }
    }
}

int pmx_launch_wta(pmx_ctx* ctx, const pmx_cv* cv, int is_max, float invalid_disparity) {
    size_t npix = (size_t)cv->H * cv->W;
    size_t want = (npix + 15) / 16;
    int grid = (int)(want < 65536 ? want : 65536);
    const int nb = (cv->D + 63) / 64;
    pmx_stage_scope t(ctx, PMX_STAGE_WTA);
#define PMX_WTA(MAXV, NBV)                                                                                             \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(wta_kernel<MAXV, NBV>), dim3(grid), dim3(kBlock), 0, ctx->stream, cv->data, npix, \
                       cv->D, (double)cv->d0, cv->subpix, invalid_disparity, ctx->disp, ctx->validity)
#define PMX_WTA_NB(MAXV)                 \
    switch (nb) {                        \
        case 1: PMX_WTA(MAXV, 1); break; \
        case 2: PMX_WTA(MAXV, 2); break; \
        case 3: PMX_WTA(MAXV, 3); break; \
        case 4: PMX_WTA(MAXV, 4); break; \
        case 5: PMX_WTA(MAXV, 5); break; \
        case 6: PMX_WTA(MAXV, 6); break; \
        case 7: PMX_WTA(MAXV, 7); break; \
        default: PMX_WTA(MAXV, 8); break; \
    }
    if (cv->D > 512) {
        pmx_set_error("pmx_wta: D = %d > 512 disparities not supported", cv->D);
        return PMX_ERR_UNSUPPORTED;
    }
    if (is_max) { PMX_WTA_NB(true) } else { PMX_WTA_NB(false) }
#undef PMX_WTA_NB
#undef PMX_WTA
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

// ---- sub-pixel refinement (refinement.cpp:28-99, vfit.cpp:28-56, quadratic.cpp:28-50) ----------
__device__ __forceinline__ bool validate_costs(float c0, float c1, float c2, bool is_max, float& ic0, float& ic2) {
    if (c0 != c0 || c2 != c2) return false;
    float inv = is_max ? -1.f : 1.f;
    ic0 = inv * c0;
    float ic1 = inv * c1;
    ic2 = inv * c2;
    return !(ic1 > ic0 || ic1 > ic2);
}

__global__ __launch_bounds__(kBlock) void refine_kernel(const float* __restrict__ cv, size_t npix, int D, double d_min,
                                                        double d_max, int subpix, int is_max, int method,
                                                        float* __restrict__ disp, int64_t* __restrict__ validity,
                                                        float* __restrict__ itp) {
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= npix) return;
    int64_t m = validity[i];
    if ((m & MSK_INVALID) != 0) { itp[i] = d_nan(); return; }
    float raw = disp[i];
    // a disparity outside the volume (a map edited by the caller; the reference would read out of bounds, refinement.cpp:56-60)
    // is left alone, with a NaN coefficient
    if (!((double)raw >= d_min && (double)raw <= d_max)) { itp[i] = d_nan(); return; }
    int k = (int)(((double)raw - d_min) * (double)subpix);
    const float* p = cv + i * (size_t)D;
    float c1 = p[k];
    if (c1 != c1) { itp[i] = c1; return; }
    if ((double)raw == d_min || (double)raw == d_max) { itp[i] = c1; validity[i] = m + MSK_STOPPED; return; }
    float c0 = p[k - 1], c2 = p[k + 1];
    float ic0, ic2, sd, sc;
    int64_t flag = 0;
    if (!validate_costs(c0, c1, c2, is_max != 0, ic0, ic2)) {
        sd = 0.f; sc = c1; flag = MSK_STOPPED;
    } else if (method == PMX_REFINE_VFIT) {
        float a = ic0 > ic2 ? c0 - c1 : c2 - c1;
        if (fabs((double)a) < 1.0e-15) {
            sd = 0.f; sc = c1;
        } else {
            sd = (c0 - c2) / (2 * a);
            sc = a * (sd - 1) + c2;
        }
    } else {
        float alpha = (c0 - 2.f * c1 + c2) / 2.f;
        float beta = (c2 - c0) / 2.f;
        float x = -beta / (2.f * alpha);
        float mx = (-1.f < x) ? x : -1.f;  // std::max(-1.f, x)
        sd = (mx < 1.f) ? mx : 1.f;        // std::min(1.f, mx)
        sc = (alpha * sd * sd) + (beta * sd) + c1;
    }
    disp[i] = raw + sd / (float)subpix;
    itp[i] = sc;
    validity[i] = m + flag;
}

int pmx_launch_refine(pmx_ctx* ctx, const pmx_cv* cv, int method, int is_max) {
    size_t npix = (size_t)cv->H * cv->W;
    double d_min = (double)cv->d0;
    double d_max = (double)cv->d0 + (double)(cv->D - 1) / (double)cv->subpix;
    pmx_stage_scope t(ctx, PMX_STAGE_REFINE);
    hipLaunchKernelGGL(refine_kernel, dim3((unsigned)((npix + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, cv->data,
                       npix, cv->D, d_min, d_max, cv->subpix, is_max, method, ctx->disp, ctx->validity, ctx->itp);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

// ---- refinement of a RIGHT map made by the diagonal search of the left volume (refinement_cpp.loop_approximate_refinement,
// refinement/cpp/src/refinement.cpp:103-182; caller AbstractRefinement.approximate_subpixel_refinement, refinement.py:124-158) ----
// Right pixel (row, col) with disparity raw looks at the LEFT volume's cell (row, diag = int(col + raw), dsp = int((-raw - d_min)
// subpix)) and its diagonal neighbours (diag - 1, dsp + subpix), (diag + 1, dsp - subpix).  The reference indexes them unchecked:
// dsp -/+ subpix outside [0, D) lands in the neighbouring pixel's run of the contiguous volume, and that is what its compiled
// module answers - restated here as flat offsets; only an offset outside the whole volume (undefined there) and a winner outside
// the volume are refused: coefficient NaN, map and mask left alone.
__global__ __launch_bounds__(kBlock) void approx_refine_kernel(const float* __restrict__ cv, int H, int W, int D, double d_min, double d_max,
                                                               int subpix, int is_max, int method, float* __restrict__ disp,
                                                               int64_t* __restrict__ validity, float* __restrict__ itp) {
    const size_t npix = (size_t)H * W;
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= npix) return;
    int64_t m = validity[i];
    if ((m & MSK_INVALID) != 0) { itp[i] = d_nan(); return; }
    const int col = (int)(i % (size_t)W);
    const size_t row = i / (size_t)W;
    const float raw = disp[i];
    const double fd = ((double)(-raw) - d_min) * (double)subpix;
    const float fdiag = (float)col + raw;  // (size_t + float in the reference: float arithmetic)
    if (!(fd > -1.0 && fd < (double)D && fdiag > -1.f && fdiag < (float)W)) { itp[i] = d_nan(); return; }
    const int dsp = (int)fd, diag = (int)fdiag;
    const long long total = (long long)npix * D;
    const long long at = ((long long)row * W + diag) * D + dsp;
    const float c1 = cv[at];
    if (c1 != c1) { itp[i] = c1; return; }
    if ((double)raw == d_min || (double)raw == d_max || diag == 0 || diag == W - 1) { itp[i] = c1; validity[i] = m + MSK_STOPPED; return; }
    const long long a0 = at - D + subpix, a2 = at + D - subpix;
    if (a0 < 0 || a2 >= total) { itp[i] = d_nan(); return; }
    const float c0 = cv[a0], c2 = cv[a2];
    float ic0, ic2, sd, sc;
    int64_t flag = 0;
    if (!validate_costs(c0, c1, c2, is_max != 0, ic0, ic2)) {
        sd = 0.f; sc = c1; flag = MSK_STOPPED;
    } else if (method == PMX_REFINE_VFIT) {
        float a = ic0 > ic2 ? c0 - c1 : c2 - c1;
        if (fabs((double)a) < 1.0e-15) {
            sd = 0.f; sc = c1;
        } else {
            sd = (c0 - c2) / (2 * a);
            sc = a * (sd - 1) + c2;
        }
    } else {
        float alpha = (c0 - 2.f * c1 + c2) / 2.f;
        float beta = (c2 - c0) / 2.f;
        float x = -beta / (2.f * alpha);
        float mx = (-1.f < x) ? x : -1.f;
        sd = (mx < 1.f) ? mx : 1.f;
        sc = (alpha * sd * sd) + (beta * sd) + c1;
    }
    disp[i] = raw + sd / (float)subpix;
    itp[i] = sc;
    validity[i] = m + flag;
}

int pmx_launch_approx_refine(pmx_ctx* ctx, const pmx_cv* cv, int method, int is_max) {
    size_t npix = (size_t)cv->H * cv->W;
    double d_min = (double)cv->d0;
    double d_max = (double)cv->d0 + (double)(cv->D - 1) / (double)cv->subpix;
    pmx_stage_scope t(ctx, PMX_STAGE_REFINE);
    hipLaunchKernelGGL(approx_refine_kernel, dim3((unsigned)((npix + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, cv->data, cv->H,
                       cv->W, cv->D, d_min, d_max, cv->subpix, is_max, method, ctx->disp, ctx->validity, ctx->itp);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

// ---- WTA fused into the last float32 SGM pass (k_sgmfam.hip) ----------------------------------------------------------------
// The pass left disp and, per pixel, near = (S[k-1], S[k], S[k+1], k) of the winner in the output domain, k = -1 where every cost
// is NaN.  This kernel applies to_disp's validity rule for those pixels (disparity.py:471-474).
__global__ __launch_bounds__(kBlock) void wta_fixup_kernel(const float4* __restrict__ near, size_t npix, int64_t* __restrict__ validity) {
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= npix) return;
    if (__float_as_int(near[i].w) == -1) {
        int64_t m = validity[i];
        if ((m & MSK_INVALID) == 0) validity[i] = MSK_INVALID;
    }
}

int pmx_launch_wta_fixup(pmx_ctx* ctx, const pmx_cv* cv) {
    size_t npix = (size_t)cv->H * cv->W;
    pmx_stage_scope t(ctx, PMX_STAGE_WTA);
    hipLaunchKernelGGL(wta_fixup_kernel, dim3((unsigned)((npix + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, (const float4*)ctx->near, npix,
                       ctx->validity);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

// refine_kernel on the three values the fused WTA kept instead of the volume (which was never written)
__global__ __launch_bounds__(kBlock) void near_refine_kernel(const float4* __restrict__ near, size_t npix, double d_min, double d_max,
                                                             int subpix, int is_max, int method, float* __restrict__ disp,
                                                             int64_t* __restrict__ validity, float* __restrict__ itp) {
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= npix) return;
    int64_t m = validity[i];
    if ((m & MSK_INVALID) != 0) { itp[i] = d_nan(); return; }
    float raw = disp[i];
    if (!((double)raw >= d_min && (double)raw <= d_max)) { itp[i] = d_nan(); return; }
    const float4 nb = near[i];
    float c1 = nb.y;
    if (c1 != c1) { itp[i] = c1; return; }
    if ((double)raw == d_min || (double)raw == d_max) { itp[i] = c1; validity[i] = m + MSK_STOPPED; return; }
    float c0 = nb.x, c2 = nb.z;
    float ic0, ic2, sd, sc;
    int64_t flag = 0;
    if (!validate_costs(c0, c1, c2, is_max != 0, ic0, ic2)) {
        sd = 0.f; sc = c1; flag = MSK_STOPPED;
    } else if (method == PMX_REFINE_VFIT) {
        float a = ic0 > ic2 ? c0 - c1 : c2 - c1;
        if (fabs((double)a) < 1.0e-15) {
            sd = 0.f; sc = c1;
        } else {
            sd = (c0 - c2) / (2 * a);
            sc = a * (sd - 1) + c2;
        }
    } else {
        float alpha = (c0 - 2.f * c1 + c2) / 2.f;
        float beta = (c2 - c0) / 2.f;
        float x = -beta / (2.f * alpha);
        float mx = (-1.f < x) ? x : -1.f;
        sd = (mx < 1.f) ? mx : 1.f;
        sc = (alpha * sd * sd) + (beta * sd) + c1;
    }
    disp[i] = raw + sd / (float)subpix;
    itp[i] = sc;
    validity[i] = m + flag;
}

int pmx_launch_near_refine(pmx_ctx* ctx, const pmx_cv* cv, int method, int is_max) {
    size_t npix = (size_t)cv->H * cv->W;
    double d_min = (double)cv->d0;
    double d_max = (double)cv->d0 + (double)(cv->D - 1) / (double)cv->subpix;
    pmx_stage_scope t(ctx, PMX_STAGE_REFINE);
    hipLaunchKernelGGL(near_refine_kernel, dim3((unsigned)((npix + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, (const float4*)ctx->near,
                       npix, d_min, d_max, cv->subpix, is_max, method, ctx->disp, ctx->validity, ctx->itp);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

// ---- packed (cost, index) keys for D-sharded WTA ------------------------------------
// key = orderable(cost) << 31 | global index, so that min over ranks of the uint64 key is the
// lexicographic (cost, index) minimum.  For "max" measures the cost is negated first.
// The cost takes bits 62..31 and the index bits 30..0, so keys are < 2^63 and can travel as int64
// (RCCL / gloo reduce int64 MIN).  All-NaN shard -> key = INT64_MAX.
__device__ __forceinline__ uint32_t orderable(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

template <bool IS_MAX>
__global__ __launch_bounds__(kBlock) void minkey_kernel(const float* __restrict__ cv, size_t npix, int D, int index_offset,
                                                        uint64_t* __restrict__ keys) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const size_t nwaves = (size_t)gridDim.x * (kBlock / 64);
    for (size_t pix = wave; pix < npix; pix += nwaves) {
        float v;
        int idx, any;
        wave_argbest<IS_MAX>(cv + pix * (size_t)D, D, lane, v, idx, any);
        if (lane == 0) {
            uint64_t key = 0x7fffffffffffffffull;
            if (any) {
                float f = IS_MAX ? -v : v;
                if (f == 0.f) f = 0.f;  // -0 and +0 must order equal
                key = ((uint64_t)orderable(f) << 31) | (uint64_t)((uint32_t)(idx + index_offset) & 0x7fffffffu);
            }
            keys[pix] = key;
        }
    }
}

int pmx_launch_minkey(pmx_ctx* ctx, const pmx_cv* cv, int is_max, int index_offset, uint64_t* keys) {
    size_t npix = (size_t)cv->H * cv->W;
    size_t want = (npix + 3) / 4;
    int grid = (int)(want < 16384 ? want : 16384);
    pmx_stage_scope t(ctx, PMX_STAGE_MINKEY);
    if (is_max)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(minkey_kernel<true>), dim3(grid), dim3(kBlock), 0, ctx->stream, cv->data, npix, cv->D,
                           index_offset, keys);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(minkey_kernel<false>), dim3(grid), dim3(kBlock), 0, ctx->stream, cv->data, npix,
                           cv->D, index_offset, keys);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

__global__ __launch_bounds__(kBlock) void from_keys_kernel(const uint64_t* __restrict__ keys, size_t npix, double d0, int subpix,
                                                           float invalid_disparity, float* __restrict__ disp,
                                                           int64_t* __restrict__ validity) {
    size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= npix) return;
    uint64_t key = keys[i];
    if (key == 0x7fffffffffffffffull) {
        disp[i] = invalid_disparity;
        int64_t m = validity[i];
        if ((m & MSK_INVALID) == 0) validity[i] = MSK_INVALID;
    } else {
        uint32_t idx = (uint32_t)(key & 0x7fffffffull);
        disp[i] = (float)(d0 + (double)idx / (double)subpix);
    }
}

int pmx_launch_from_keys(pmx_ctx* ctx, const uint64_t* keys, double d0, int subpix, float invalid_disparity) {
    size_t npix = (size_t)ctx->H * ctx->W;
    pmx_stage_scope t(ctx, PMX_STAGE_MINKEY);
    hipLaunchKernelGGL(from_keys_kernel, dim3((unsigned)((npix + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, keys, npix,
                       d0, subpix, invalid_disparity, ctx->disp, ctx->validity);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

// ---- all-NaN pixels (criteria.py:303-305) ---------------------------------------------------------
__global__ __launch_bounds__(kBlock) void nan_pixels_kernel(const float* __restrict__ cv, size_t npix, int D,
                                                            uint8_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const size_t nwaves = (size_t)gridDim.x * (kBlock / 64);
    for (size_t pix = wave; pix < npix; pix += nwaves) {
        const float* p = cv + pix * (size_t)D;
        int any = 0;
        for (int k = lane; k < D; k += 64) {
            float x = p[k];
            any |= (x == x);
        }
        unsigned long long b = __ballot(any);
        if (lane == 0) out[pix] = b ? 0 : 1;
    }
}

int pmx_launch_nan_pixels(pmx_ctx* ctx, const pmx_cv* cv, uint8_t* dev_out) {
    size_t npix = (size_t)cv->H * cv->W;
    size_t want = (npix + 3) / 4;
    int grid = (int)(want < 16384 ? want : 16384);
    hipLaunchKernelGGL(nan_pixels_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, cv->data, npix, cv->D, dev_out);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

// ---- the cost volume's validity mask put together on the device (criteria.py:66-158 line + :291-322 + :325-353) -----------
// validity(r, c) = base (one line for every row, or a full map), | RIGHT_NODATA_OR_DISPARITY_RANGE_MISSING where the volume was
// NaN for every disparity when pmx_cv_mark_missing looked, then the frame of `border` pixels = LEFT_NODATA_OR_BORDER.
__global__ __launch_bounds__(kBlock) void compose_validity_kernel(const int64_t* base, int base_rows,  // (base may BE validity)
                                                                  const uint8_t* __restrict__ missing, int H, int W, int border,
                                                                  int64_t* validity) {
    const size_t npix = (size_t)H * W;
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= npix) return;
    const int r = (int)(i / W), c = (int)(i - (size_t)r * W);
    int64_t v = base_rows == 1 ? base[c] : base[i];
    if (missing && missing[i]) v |= 2;  // PANDORA_MSK_PIXEL_RIGHT_NODATA_OR_DISPARITY_RANGE_MISSING
    if (border > 0 && (r < border || r >= H - border || c < border || c >= W - border)) v = 1;  // ..._LEFT_NODATA_OR_BORDER
    validity[i] = v;
}

int pmx_launch_compose_validity_into(pmx_ctx* ctx, const int64_t* dev_base, int base_rows, const uint8_t* dev_missing, int border,
                                     int64_t* dev_out) {
    const size_t npix = (size_t)ctx->H * ctx->W;
    hipLaunchKernelGGL(compose_validity_kernel, dim3((unsigned)((npix + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, dev_base,
                       base_rows, dev_missing, ctx->H, ctx->W, border, dev_out);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

int pmx_launch_compose_validity(pmx_ctx* ctx, const int64_t* dev_base, int base_rows, const uint8_t* dev_missing, int border) {
    return pmx_launch_compose_validity_into(ctx, dev_base, base_rows, dev_missing, border, ctx->validity);
}

// ---- reverse_cost_volume (matching_cost.cpp:26-56): out(i, j, d) = in(i, j + d + min_disp, D-1-d) ---------------------
// An output tile of TC columns x D disparities reads a parallelogram of the input: from input column q it needs the
// run of (at most TC) consecutive disparities D-1-d, d = q - min_disp - c, c in the tile - contiguous in memory.  The runs go
// through an LDS tile laid out as the output (row pitch D+1 or D+2 floats, odd stride across the lanes of a run: no bank
// conflicts), which then leaves as ONE contiguous block of TC*D floats.  Both sides of the re-index are coalesced.
template <int TC>
__global__ __launch_bounds__(kBlock) void reverse_tiled_kernel(const float* __restrict__ in, int W, int D, int min_disp,
                                                               float* __restrict__ out) {
    extern __shared__ float tile[];  // [TC][pitch]
    const int pitch = D + 1 + (D & 1);  // even, so that pitch - 1 is odd
    const int r = blockIdx.y, c0 = blockIdx.x * TC;
    const size_t base = (size_t)r * W * D;
    const int t = threadIdx.x % TC, slot = threadIdx.x / TC;
    constexpr int QPI = kBlock / TC;    // input columns handled per iteration
    const int q0 = c0 + min_disp;       // first input column of the band (d = 0 for the tile's first column)
    const int nq = TC + D - 1;
    const int c = c0 + t;
    for (int iq = slot; iq < nq; iq += QPI) {
        const int q = q0 + iq;
        const int d = q - min_disp - c;  // = iq - t
        if (d >= 0 && d < D) {
            float v = d_nan();
            if (q >= 0 && q < W && c < W) v = in[base + (size_t)q * D + (D - 1 - d)];
            tile[t * pitch + d] = v;
        }
    }
    __syncthreads();
    const int ncol = min(TC, W - c0);
    const int n = ncol * D;
    float* dst = out + base + (size_t)c0 * D;
    for (int i = threadIdx.x; i < n; i += kBlock) {
        const int tc = i / D, d = i - tc * D;
        dst[i] = tile[tc * pitch + d];
    }
}

// plain gather for disparity ranges too wide for the LDS tile
__global__ __launch_bounds__(kBlock) void reverse_kernel(const float* __restrict__ in, int W, int D, int min_disp,
                                                         float* __restrict__ out) {
    const int r = blockIdx.y;
    int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= W * D) return;
    int c = j / D, d = j - c * D;
    int col = c + d + min_disp;
    size_t base = (size_t)r * W * D;
    out[base + j] = (col < 0 || col >= W) ? d_nan() : in[base + (size_t)col * D + (D - 1 - d)];
}

int pmx_launch_reverse(pmx_ctx* ctx, const pmx_cv* in, int min_disp, pmx_cv* out) {
    pmx_stage_scope t(ctx, PMX_STAGE_REVERSE);
    const int pitch = in->D + 1 + (in->D & 1);
    if (in->D <= 256) {
        dim3 grid((in->W + 31) / 32, in->H);
        hipLaunchKernelGGL(reverse_tiled_kernel<32>, grid, dim3(kBlock), 32 * pitch * sizeof(float), ctx->stream, in->data, in->W, in->D,
                           min_disp, out->data);
    } else if (in->D > 900) {
        dim3 grid((in->W * in->D + kBlock - 1) / kBlock, in->H);
        hipLaunchKernelGGL(reverse_kernel, grid, dim3(kBlock), 0, ctx->stream, in->data, in->W, in->D, min_disp, out->data);
    } else {
        dim3 grid((in->W + 15) / 16, in->H);
        hipLaunchKernelGGL(reverse_tiled_kernel<16>, grid, dim3(kBlock), 16 * pitch * sizeof(float), ctx->stream, in->data, in->W, in->D,
                           min_disp, out->data);
    }
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}
