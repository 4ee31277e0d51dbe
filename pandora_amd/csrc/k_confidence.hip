// k_confidence.hip - SURVEY 8f N4: the ambiguity integral of cost_volume_confidence (a reduction over D on the resident
// float32 volume).  gfx950.
#include <cstring>

#include "pmx_internal.h"

static constexpr int kBlock = 256;
static constexpr int kMaxEtas = 1024;

__device__ __forceinline__ float c_inf() { return __int_as_float(0x7f800000); }

// float <-> unsigned key that orders like the float (for atomicMin / atomicMax on the volume's extrema)
__device__ __forceinline__ uint32_t f2ord(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ __forceinline__ float ord2f(uint32_t k) {
    const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}

// cost_volume_confidence_tools.cpp:41-88 min_max_cost, global part: extrema of the non-NaN costs (sign-flipped when the
// measure is a similarity, ambiguity.py:117-119)
__global__ __launch_bounds__(kBlock) void volume_minmax_kernel(const float* __restrict__ cv, size_t n, float sign, uint32_t* __restrict__ mm) {
    uint32_t lo = 0xffffffffu, hi = 0u;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        const float v = cv[i] * sign;
        if (v == v) {
            const uint32_t k = f2ord(v);
            lo = k < lo ? k : lo;
            hi = k > hi ? k : hi;
        }
    }
    // wave reduce then one atomic per wave
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t l2 = __shfl_down(lo, off), h2 = __shfl_down(hi, off);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&mm[0], lo);
        atomicMax(&mm[1], hi);
    }
}

struct amb_args {
    const float* cv;
    const float* etas;         // [nbr_etas] device
    const int64_t* grid_min;   // [H][W] device
    const int64_t* grid_max;
    const uint32_t* mm;        // ordered keys of the global min / max
    float* amb;
    int H, W, D, d0, subpix, nbr_etas;
    float sign;
};

// ambiguity.cpp:28-142.  Four pixels per wavefront (one per 16-lane row), lane `sub` strides over the disparities.
// The inner double loop "for eta, for d: nc[d] <= ne + eta" is evaluated per d as E - (first eta index that satisfies
// it): the thresholds fl32(ne + eta_i) are non-decreasing in i, so the count is exact.
__global__ __launch_bounds__(kBlock) void ambiguity_kernel(amb_args a) {
    __shared__ float etas_s[kMaxEtas];
    for (int i = threadIdx.x; i < a.nbr_etas; i += kBlock) etas_s[i] = a.etas[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4;
    const size_t npix = (size_t)a.H * a.W;
    const size_t wave = (size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const size_t nwaves = (size_t)gridDim.x * (kBlock / 64);
    const float min_cost = ord2f(a.mm[0]), max_cost = ord2f(a.mm[1]);
    const float diff = max_cost - min_cost;
    const int E = a.nbr_etas, D = a.D;
    for (size_t quad = wave; quad * 4 < npix; quad += nwaves) {
        const size_t pix = min(quad * 4 + grp, npix - 1);
        const float* row = a.cv + pix * (size_t)D;
        // per-pixel minimum over the non-NaN costs
        float lo = c_inf();
        bool any = false;
        for (int k = sub; k < D; k += 16) {
            const float v = row[k] * a.sign;
            if (v == v) { any = true; lo = fminf(lo, v); }
        }
#pragma unroll
        for (int m = 8; m > 0; m >>= 1) {
            lo = fminf(lo, __shfl_xor(lo, m));
            any = any | (__shfl_xor((int)any, m) != 0);
        }
        float result;
        const float ne = any ? (lo - min_cost) / diff : __int_as_float(0x7fc00000);
        if (ne != ne) {
            result = (float)(E * D);
        } else {
            // searchsorted(disparity_range, grid) with right = D - 1 (cost_volume_confidence_tools.cpp:22-39)
            auto range_at = [&](int k) { return (float)((double)a.d0 + (double)k / (double)a.subpix); };
            auto search = [&](float value) {
                int left = 0, right = D - 1;
                while (left < right) {
                    const int mid = left + (right - left) / 2;
                    if (range_at(mid) < value) left = mid + 1; else right = mid;
                }
                return left;
            };
            const int i0 = a.grid_min ? search((float)a.grid_min[pix]) : 0, i1 = a.grid_max ? search((float)a.grid_max[pix]) + 1 : D;  // no grids: the whole range
            int count = 0;
            for (int k = sub; k < D; k += 16) {
                const float v = row[k] * a.sign;
                float nc;
                if (v != v) nc = (k >= i0 && k < i1) ? -c_inf() : c_inf();
                else nc = (v - min_cost) / diff;
                // first i with nc <= ne + eta_i
                int left = 0, right = E;
                while (left < right) {
                    const int mid = (left + right) >> 1;
                    if (nc <= ne + etas_s[mid]) right = mid; else left = mid + 1;
                }
                count += E - left;
            }
#pragma unroll
            for (int m = 8; m > 0; m >>= 1) count += __shfl_xor(count, m);
            result = (float)count;
        }
        if (sub == 0 && quad * 4 + grp < npix) a.amb[pix] = result;
    }
}

int pmx_launch_ambiguity(pmx_ctx* ctx, pmx_cv* cv, const float* d_etas, int nbr_etas, const int64_t* d_gmin, const int64_t* d_gmax,
                         int negate, uint32_t* d_mm, float* d_amb) {
    const float sign = negate ? -1.f : 1.f;
    const size_t n = cv->cells();
    const uint32_t init[2] = {0xffffffffu, 0u};
    PMX_HIP(hipMemcpyAsync(d_mm, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(volume_minmax_kernel, dim3(4096), dim3(kBlock), 0, ctx->stream, cv->data, n, sign, d_mm);
    amb_args a;
    a.cv = cv->data; a.etas = d_etas; a.grid_min = d_gmin; a.grid_max = d_gmax; a.mm = d_mm; a.amb = d_amb;
    a.H = cv->H; a.W = cv->W; a.D = cv->D; a.d0 = cv->d0; a.subpix = cv->subpix; a.nbr_etas = nbr_etas; a.sign = sign;
    const size_t want = ((size_t)cv->H * cv->W + 15) / 16;
    hipLaunchKernelGGL(ambiguity_kernel, dim3((unsigned)(want < 65536 ? want : 65536)), dim3(kBlock), 0, ctx->stream, a);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}


// ---- use_confidence of the SGM step (docs/source/userguide/plugins/plugin_libsgm.rst:38-47): E(D) takes C(p, d) * Confidence(p).
// One weight per pixel; NaN weights (pixels without a confidence) count as 1, NaN costs stay NaN.
__global__ __launch_bounds__(kBlock) void scale_pixels_kernel(float* __restrict__ cv, const float* __restrict__ w, int W, int D) {
    const int r = blockIdx.y;
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= W * D) return;
    const float wp = w[(size_t)r * W + j / D];
    if (wp == wp) cv[(size_t)r * W * D + j] *= wp;
}

int pmx_launch_scale_pixels(pmx_ctx* ctx, pmx_cv* cv, const float* d_weights) {
    dim3 grid((cv->W * cv->D + kBlock - 1) / kBlock, cv->H);
    hipLaunchKernelGGL(scale_pixels_kernel, grid, dim3(kBlock), 0, ctx->stream, cv->data, d_weights, cv->W, cv->D);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

// ---- risk (cost_volume_confidence/cpp/src/risk.cpp:28-197 driven as risk.py:144-166) ------------------------------------
// For every eta the reference scans the D normalised costs of a pixel for the first and the last disparity index within
// eta of the pixel's minimum (double comparison: float cost > float extremum + double eta) and averages, over the etas,
// the span, 1 + span - sampled ambiguity (float comparison with float32 etas, ambiguity.cpp:120-128) and the disparities
// at the two ends.  Here every disparity k gets r(k) = the first eta index that admits it (the thresholds grow with
// the eta index); k is the FIRST admitted index exactly for the etas in [r(k), min r(k' < k)) and the LAST one for the
// etas in [r(k), min r(k' > k)), so the sums over etas are sums over k of k times an interval length: one ascending and
// one descending sweep in chunks of 16 disparities with a running prefix / suffix minimum.  Every term is an integer
// (or a multiple of 1/subpix), far below 2^24: the reference's float32 running sums are exact, and so is this.
struct risk_args {
    const float* cv;
    const double* etas;        // [nbr_etas] device, ascending, etas[0] >= 0
    const int64_t* grid_min;
    const int64_t* grid_max;
    const uint32_t* mm;
    float* risk_max;
    float* risk_min;
    float* disp_sup;
    float* disp_inf;
    int H, W, D, d0, subpix, nbr_etas;
    float sign;
};

__global__ __launch_bounds__(kBlock) void risk_kernel(risk_args a) {
    __shared__ double etas_d[kMaxEtas];
    __shared__ float etas_f[kMaxEtas];
    for (int i = threadIdx.x; i < a.nbr_etas; i += kBlock) { etas_d[i] = a.etas[i]; etas_f[i] = (float)a.etas[i]; }
    __syncthreads();
    const int lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4;
    const size_t npix = (size_t)a.H * a.W;
    const size_t wave = (size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const size_t nwaves = (size_t)gridDim.x * (kBlock / 64);
    const float min_cost = ord2f(a.mm[0]), max_cost = ord2f(a.mm[1]);
    const float diff = max_cost - min_cost;
    const int E = a.nbr_etas, D = a.D;
    const float qnan = __int_as_float(0x7fc00000);
    for (size_t quad = wave; quad * 4 < npix; quad += nwaves) {
        const size_t pix = min(quad * 4 + grp, npix - 1);
        const float* row = a.cv + pix * (size_t)D;
        float lo = c_inf();
        bool any = false;
        for (int k = sub; k < D; k += 16) {
            const float v = row[k] * a.sign;
            if (v == v) { any = true; lo = fminf(lo, v); }
        }
#pragma unroll
        for (int m = 8; m > 0; m >>= 1) {
            lo = fminf(lo, __shfl_xor(lo, m));
            any = any | (__shfl_xor((int)any, m) != 0);
        }
        const float ne = any ? (lo - min_cost) / diff : qnan;
        const bool dead = ne != ne;  // uniform over the 16 lanes of the pixel
        auto range_at = [&](int k) { return (float)((double)a.d0 + (double)k / (double)a.subpix); };
        auto search = [&](float value) {
            int left = 0, right = D - 1;
            while (left < right) {
                const int mid = left + (right - left) / 2;
                if (range_at(mid) < value) left = mid + 1; else right = mid;
            }
            return left;
        };
        const int i0 = a.grid_min ? search((float)a.grid_min[pix]) : 0, i1 = a.grid_max ? search((float)a.grid_max[pix]) + 1 : D;  // no grids: the whole range
        const double ne_d = (double)ne;
        // normalised cost of disparity k (+inf past the end of the volume: never admitted)
        auto norm_at = [&](int k) {
            if (k >= D) return c_inf();
            const float v = row[k] * a.sign;
            if (v != v) return (k >= i0 && k < i1) ? -c_inf() : c_inf();
            return (v - min_cost) / diff;
        };
        auto first_eta_d = [&](float nc) {  // first i with !(nc > ne + eta_i), in double (risk.cpp:140)
            int left = 0, right = E;
            while (left < right) {
                const int mid = (left + right) >> 1;
                if (!((double)nc > ne_d + etas_d[mid])) right = mid; else left = mid + 1;
            }
            return left;
        };
        long long s_lo = 0, s_hi = 0;
        int amb = 0;
        // ascending sweep: first admitted index per eta, and the ambiguity integral on the way
        int cur = E;
        for (int base = 0; base < D; base += 16) {
            const int k = base + sub;
            const float nc = norm_at(k);
            const int r = dead ? E : first_eta_d(nc);
            {
                int left = 0, right = E;  // first i with nc <= ne + eta_i, in float (ambiguity.cpp:123)
                while (left < right) {
                    const int mid = (left + right) >> 1;
                    if (nc <= ne + etas_f[mid]) right = mid; else left = mid + 1;
                }
                amb += E - left;
            }
            int pm = r;  // inclusive prefix minimum over the 16 lanes of the pixel
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) {
                const int o = __shfl_up(pm, m, 16);
                if (sub >= m) pm = min(pm, o);
            }
            int excl = __shfl_up(pm, 1, 16);
            excl = min(sub == 0 ? E : excl, cur);
            if (r < excl) s_lo += (long long)k * (excl - r);
            cur = min(cur, __shfl(pm, 15, 16));
        }
        // descending sweep: last admitted index per eta
        cur = E;
        for (int base = ((D - 1) / 16) * 16; base >= 0; base -= 16) {
            const int k = base + sub;
            const int r = dead ? E : first_eta_d(norm_at(k));
            int sm = r;  // inclusive suffix minimum
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) {
                const int o = __shfl_down(sm, m, 16);
                if (sub + m < 16) sm = min(sm, o);
            }
            int excl = __shfl_down(sm, 1, 16);
            excl = min(sub == 15 ? E : excl, cur);
            if (r < excl) s_hi += (long long)k * (excl - r);
            cur = min(cur, __shfl(sm, 0, 16));
        }
#pragma unroll
        for (int m = 8; m > 0; m >>= 1) {
            s_lo += __shfl_xor(s_lo, m);
            s_hi += __shfl_xor(s_hi, m);
            amb += __shfl_xor(amb, m);
        }
        if (sub == 0 && quad * 4 + grp < npix) {
            if (dead) {
                a.risk_max[pix] = qnan; a.risk_min[pix] = qnan; a.disp_sup[pix] = qnan; a.disp_inf[pix] = qnan;
            } else {
                const float fe = (float)E;
                const float span = (float)(s_hi - s_lo);
                a.risk_max[pix] = span / fe;
                a.risk_min[pix] = (float)((long long)E + (s_hi - s_lo) - (long long)amb) / fe;
                a.disp_sup[pix] = (float)((double)a.d0 * E + (double)s_hi / a.subpix) / fe;
                a.disp_inf[pix] = (float)((double)a.d0 * E + (double)s_lo / a.subpix) / fe;
            }
        }
    }
}

int pmx_launch_risk(pmx_ctx* ctx, pmx_cv* cv, const double* d_etas, int nbr_etas, const int64_t* d_gmin, const int64_t* d_gmax,
                    int negate, uint32_t* d_mm, float* d_out4) {
    const float sign = negate ? -1.f : 1.f;
    const size_t n = cv->cells(), npix = (size_t)cv->H * cv->W;
    const uint32_t init[2] = {0xffffffffu, 0u};
    PMX_HIP(hipMemcpyAsync(d_mm, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(volume_minmax_kernel, dim3(4096), dim3(kBlock), 0, ctx->stream, cv->data, n, sign, d_mm);
    risk_args a;
    a.cv = cv->data; a.etas = d_etas; a.grid_min = d_gmin; a.grid_max = d_gmax; a.mm = d_mm;
    a.risk_max = d_out4; a.risk_min = d_out4 + npix; a.disp_sup = d_out4 + 2 * npix; a.disp_inf = d_out4 + 3 * npix;
    a.H = cv->H; a.W = cv->W; a.D = cv->D; a.d0 = cv->d0; a.subpix = cv->subpix; a.nbr_etas = nbr_etas; a.sign = sign;
    const size_t want = (npix + 15) / 16;
    hipLaunchKernelGGL(risk_kernel, dim3((unsigned)(want < 65536 ? want : 65536)), dim3(kBlock), 0, ctx->stream, a);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

// ---- interval bounds (cost_volume_confidence/cpp/src/interval_bounds.cpp:28-161) ---------------------------------------
// Inside the pixel's [grid_min, grid_max]: possibility = type_factor * norm + 1 - max(type_factor * norm); the bounds are
// the first and last disparity whose possibility reaches the threshold, moved out by one sample where the end's
// possibility truncates to 1.  The float operations keep the reference's order (no contraction: *_rn intrinsics).
struct ivb_args {
    const float* cv;
    const int64_t* grid_min;
    const int64_t* grid_max;
    const uint32_t* mm;
    float* inf;
    float* sup;
    int H, W, D, d0, subpix;
    float threshold, type_factor;
};

__global__ __launch_bounds__(kBlock) void interval_bounds_kernel(ivb_args a) {
    const int lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4;
    const size_t npix = (size_t)a.H * a.W;
    const size_t wave = (size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const size_t nwaves = (size_t)gridDim.x * (kBlock / 64);
    const float min_cost = ord2f(a.mm[0]), max_cost = ord2f(a.mm[1]);
    const float diff = max_cost - min_cost;
    const int D = a.D;
    const float qnan = __int_as_float(0x7fc00000);
    for (size_t quad = wave; quad * 4 < npix; quad += nwaves) {
        const size_t pix = min(quad * 4 + grp, npix - 1);
        const float* row = a.cv + pix * (size_t)D;
        auto range_at = [&](int k) { return (float)((double)a.d0 + (double)k / (double)a.subpix); };
        auto search = [&](float value) {
            int left = 0, right = D - 1;
            while (left < right) {
                const int mid = left + (right - left) / 2;
                if (range_at(mid) < value) left = mid + 1; else right = mid;
            }
            return left;
        };
        const int i0 = a.grid_min ? search((float)a.grid_min[pix]) : 0, i1 = a.grid_max ? search((float)a.grid_max[pix]) + 1 : D;  // no grids: the whole range
        auto norm_at = [&](int k) { return __fdiv_rn(__fsub_rn(row[k], min_cost), diff); };
        float mx = -c_inf();
        for (int k = i0 + sub; k < i1; k += 16) {
            const float nrm = norm_at(k);
            if (nrm == nrm) mx = fmaxf(mx, __fmul_rn(a.type_factor, nrm));
        }
#pragma unroll
        for (int m = 8; m > 0; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m));
        auto poss_at = [&](int k) { return __fsub_rn(__fadd_rn(__fmul_rn(a.type_factor, norm_at(k)), 1.f), mx); };  // NaN stays NaN
        int lo_k = 0x7fffffff, hi_k = -1;
        if (mx > -c_inf() && mx < c_inf()) {
            for (int k = i0 + sub; k < i1; k += 16)
                if (poss_at(k) >= a.threshold) { lo_k = min(lo_k, k); hi_k = max(hi_k, k); }
        }
#pragma unroll
        for (int m = 8; m > 0; m >>= 1) {
            lo_k = min(lo_k, __shfl_xor(lo_k, m));
            hi_k = max(hi_k, __shfl_xor(hi_k, m));
        }
        if (sub == 0 && quad * 4 + grp < npix) {
            float lo = qnan, hi = qnan;
            if (hi_k >= 0) {
                if (lo_k > 0 && (int)poss_at(lo_k) == 1) --lo_k;
                if (hi_k < D - 1 && (int)poss_at(hi_k) == 1) ++hi_k;
                lo = range_at(lo_k);
                hi = range_at(hi_k);
            }
            a.inf[pix] = lo;
            a.sup[pix] = hi;
        }
    }
}

int pmx_launch_interval_bounds(pmx_ctx* ctx, pmx_cv* cv, float threshold, float type_factor, const int64_t* d_gmin,
                               const int64_t* d_gmax, uint32_t* d_mm, float* d_out2) {
    const size_t n = cv->cells(), npix = (size_t)cv->H * cv->W;
    const uint32_t init[2] = {0xffffffffu, 0u};
    PMX_HIP(hipMemcpyAsync(d_mm, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(volume_minmax_kernel, dim3(4096), dim3(kBlock), 0, ctx->stream, cv->data, n, 1.f, d_mm);
    ivb_args a;
    a.cv = cv->data; a.grid_min = d_gmin; a.grid_max = d_gmax; a.mm = d_mm; a.inf = d_out2; a.sup = d_out2 + npix;
    a.H = cv->H; a.W = cv->W; a.D = cv->D; a.d0 = cv->d0; a.subpix = cv->subpix; a.threshold = threshold; a.type_factor = type_factor;
    const size_t want = (npix + 15) / 16;
    hipLaunchKernelGGL(interval_bounds_kernel, dim3((unsigned)(want < 65536 ? want : 65536)), dim3(kBlock), 0, ctx->stream, a);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}


// ---- order statistics of a float32 map (np.percentile's partition, ambiguity.py:168-184) ---------------------------------------
// The k-th smallest of n values by radix selection on order-preserving 32-bit keys: three histogram passes (11 + 11 + 10 bits),
// each over the values whose higher bits match what the earlier passes chose.  Exact (it returns an element of the input), no
// sort.  NaNs sort last, as in numpy; their count comes back with the first pass.
__device__ __forceinline__ uint32_t os_key(float v) {
    const uint32_t u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0xffffffffu;  // NaN of either sign: last
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(256) void os_hist_kernel(const float* __restrict__ v, size_t n, uint32_t prefix, int prefix_bits, int shift,
                                                      int bits, uint32_t* __restrict__ hist) {
    __shared__ uint32_t h[2048];
    const int nb = 1 << bits;
    for (int i = threadIdx.x; i < nb; i += 256) h[i] = 0;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint32_t k = os_key(v[i]);
        if (prefix_bits == 0 || (k >> (32 - prefix_bits)) == prefix) atomicAdd(&h[(k >> shift) & (uint32_t)(nb - 1)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += 256)
        if (h[i]) atomicAdd(&hist[i], h[i]);
}

int pmx_launch_order_statistic(pmx_ctx* ctx, const float* dev_values, size_t n, size_t rank, uint32_t* dev_hist, uint32_t* host_hist,
                               float* out) {
    const int widths[3] = {11, 11, 10};
    uint32_t prefix = 0;
    int prefix_bits = 0;
    size_t k = rank;  // rank among the values that share the prefix
    for (int pass = 0; pass < 3; ++pass) {
        const int bits = widths[pass], shift = 32 - prefix_bits - bits;
        PMX_HIP(hipMemsetAsync(dev_hist, 0, 2048 * sizeof(uint32_t), ctx->stream));
        hipLaunchKernelGGL(os_hist_kernel, dim3(1024), dim3(256), 0, ctx->stream, dev_values, n, prefix, prefix_bits, shift, bits, dev_hist);
        PMX_HIP(hipMemcpyAsync(host_hist, dev_hist, ((size_t)1 << bits) * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        PMX_HIP(hipStreamSynchronize(ctx->stream));
        uint32_t b = 0;
        for (; b < (1u << bits); ++b) {
            if (k < host_hist[b]) break;
            k -= host_hist[b];
        }
        PMX_CHECK(b < (1u << bits), PMX_ERR_STATE, "pmx_order_statistics: rank beyond the values (internal)");
        prefix = (prefix << bits) | b;
        prefix_bits += bits;
    }
    uint32_t u = (prefix & 0x80000000u) ? (prefix & 0x7fffffffu) : ~prefix;  // the key back to the float's bits
    if (prefix == 0xffffffffu) u = 0x7fc00000u;  // (NaN)
    memcpy(out, &u, sizeof(float));
    return PMX_OK;
}
