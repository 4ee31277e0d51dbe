// k_matching.hip - matching-cost kernels: census transform + Hamming cost, SAD/SSD, ZNCC,
// the cv_masked NaN predicate, mask dilation, sub-pixel right-image shift.  gfx950.
//
// HBM roofline: every kernel here WRITES the float32 volume once (4 B/cell algorithmic); the two
// images and their census codes are O(H*W) and live in L2 / Infinity Cache.
#include <type_traits>

#include "pmx_internal.h"

static constexpr int kBlock = 256;

__device__ __forceinline__ float qnan() { return __int_as_float(0x7fc00000); }

// ---- sub-pixel shift (img_tools.py:713-752) ---------------------------------------------------
__global__ void shift_right_kernel(const float* __restrict__ R, int H, int W, double f, float* __restrict__ out) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    int r = blockIdx.y;
    if (c >= W - 1) return;
    double a = R[(size_t)r * W + c], b = R[(size_t)r * W + c + 1];
    out[(size_t)r * (W - 1) + c] = (float)((1.0 - f) * a + f * b);
}

int pmx_launch_shift_right(pmx_ctx* ctx, const float* R, int H, int W, int subpix, int k, float* out) {
    dim3 grid((W - 1 + kBlock - 1) / kBlock, H);
    hipLaunchKernelGGL(shift_right_kernel, grid, dim3(kBlock), 0, ctx->stream, R, H, W, (double)k / (double)subpix, out);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

// ---- NaN fill -----------------------------------------------------------------------------------
__global__ void fill_nan_kernel(float* __restrict__ p, size_t n) {
    size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    float4 v = make_float4(qnan(), qnan(), qnan(), qnan());
    for (; i + 3 < n; i += stride) *reinterpret_cast<float4*>(p + i) = v;
    if (i < n)
        for (size_t j = i; j < n; ++j) p[j] = qnan();
}

int pmx_launch_fill_nan(pmx_ctx* ctx, float* p, size_t n) {
    size_t want = (n / 4 + kBlock - 1) / kBlock + 1;
    int grid = (int)(want < 8192 ? want : 8192);
    hipLaunchKernelGGL(fill_nan_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, p, n);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

// ---- mask dilation (matching_cost.py:484-602) ---------------------------------------------------
__global__ void mask_dilate_kernel(const int16_t* __restrict__ msk, int H, int W, int o, int valid, int nodata,
                                   uint8_t* __restrict__ bad) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    int r = blockIdx.y;
    if (c >= W) return;
    int16_t m = msk[(size_t)r * W + c];
    bool b = (m != valid) && (m != nodata);
    for (int i = -o; i <= o && !b; ++i) {
        int rr = r + i;
        if (rr < 0 || rr >= H) continue;
        for (int j = -o; j <= o; ++j) {
            int cc = c + j;
            if (cc < 0 || cc >= W) continue;
            if (msk[(size_t)rr * W + cc] == nodata) { b = true; break; }
        }
    }
    bad[(size_t)r * W + c] = b ? 1 : 0;
}

int pmx_launch_mask_dilate(pmx_ctx* ctx, const int16_t* msk, int H, int W, int win, int valid, int nodata, uint8_t* bad) {
    dim3 grid((W + kBlock - 1) / kBlock, H);
    hipLaunchKernelGGL(mask_dilate_kernel, grid, dim3(kBlock), 0, ctx->stream, msk, H, W, win / 2, valid, nodata, bad);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

// ---- shared cell geometry -----------------------------------------------------------------------
// A cell (r,c,k) has a cost iff the w x w window around (r,c) is inside the left image and the
// window around (r, c + floor(d)) is inside shifted right image (k % subpix), whose width is W
// (phase 0) or W-1.  Same rule for census (census.cpp:132-152), sad/ssd (sad_ssd.py:180-204) and
// zncc (zncc.py:183-230) - the reference tests run one NaN-pattern fixture over all four.
struct cell_geom {
    int ph;  // sub-pixel phase = index of the shifted right image
    int q;   // right column of the window centre
    bool ok;
};

__device__ __forceinline__ cell_geom cell_geometry(int H, int W, int d0, int subpix, int o, int r, int c, int k) {
    cell_geom g;
    int kk = k / subpix;
    g.ph = k - kk * subpix;
    g.q = c + d0 + kk;
    int wk = g.ph == 0 ? W : W - 1;
    g.ok = (r >= o) && (r < H - o) && (c >= o) && (c < W - o) && (g.q >= o) && (g.q < wk - o);
    return g;
}

// cv_masked predicate (matching_cost.py:815-860): true = the cell must be NaN
__device__ __forceinline__ bool cell_masked(const pmx_mc_params& p, int r, int c, int k, int ph, int q) {
    bool bad = false;
    if (p.bad_left || p.bad_right) {
        int qmax = ph == 0 ? p.W - 1 : p.W - 2;
        if (q >= 0 && q <= qmax) {
            if (p.bad_left && p.bad_left[(size_t)r * p.W + c]) bad = true;
            if (p.bad_right) {
                if (p.bad_right[(size_t)r * p.W + q]) bad = true;
                if (ph != 0 && p.bad_right[(size_t)r * p.W + q + 1]) bad = true;
            }
        }
    }
    if (p.grid_min) {
        double d = (double)p.d0 + (double)k / (double)p.subpix;
        if (d < p.grid_min[(size_t)r * p.W + c] || d > p.grid_max[(size_t)r * p.W + c]) bad = true;
    }
    return bad;
}

static pmx_mc_params make_params(pmx_ctx* ctx, const pmx_cv* cv, int win) {
    pmx_mc_params p;
    p.H = cv->H; p.W = cv->W; p.D = cv->D; p.d0 = cv->d0; p.subpix = cv->subpix; p.win = win;
    p.left = ctx->left;
    for (int k = 0; k < PMX_MAX_SUBPIX; ++k) p.right[k] = ctx->right[k];
    p.bad_left = ctx->bad_left;
    p.bad_right = ctx->bad_right;
    p.grid_min = ctx->grid_min;
    p.grid_max = ctx->grid_max;
    p.apply_mask = 0;
    return p;
}

// ---- census transform ---------------------------------------------------------------------------
// census.cpp:45-95: bit = (window pixel > centre), strict.  Only the Hamming distance of two codes
// is ever used, so any fixed bit order is equivalent to the reference's MSB-first bytes; codes are
// packed little-end first into NW = ceil(w*w/32) uint32 words.  Border pixels get code 0.
// A workgroup takes a 64 x 16 tile (+ the window's border) through LDS, a thread four rows of one column: a window column is read
// once for the four windows it is part of (WIN x (WIN + 3) LDS reads for four codes instead of 4 x WIN x WIN).
static constexpr int kCtRows = 4;                              // output rows per thread
static constexpr int kCtTileY = (kBlock / 64) * kCtRows;       // 16
template <int WIN>
__global__ __launch_bounds__(kBlock) void census_transform_kernel(const float* __restrict__ img, int H, int Wd,
                                                                  uint32_t* __restrict__ codes) {
    constexpr int O = WIN / 2;
    constexpr int NW = (WIN * WIN + 31) / 32;
    constexpr int TX = 64, TY = kCtTileY;
    constexpr int LW = TX + 2 * O, LH = TY + 2 * O;
    __shared__ float tile[LH][LW + 1];
    const int c0 = blockIdx.x * TX, r0 = blockIdx.y * TY;
    for (int i = threadIdx.x; i < LH * LW; i += kBlock) {
        const int ty = i / LW, tx = i - ty * LW;
        const int r = r0 + ty - O, c = c0 + tx - O;
        tile[ty][tx] = (r >= 0 && r < H && c >= 0 && c < Wd) ? img[(size_t)r * Wd + c] : 0.f;
    }
    __syncthreads();
    const int tx = threadIdx.x % TX, ty = (threadIdx.x / TX) * kCtRows;
    const int c = c0 + tx;
    if (c >= Wd) return;
    uint32_t w[kCtRows][NW];
    float ctr[kCtRows];
#pragma unroll
    for (int y = 0; y < kCtRows; ++y) {
        ctr[y] = tile[ty + y + O][tx + O];
#pragma unroll
        for (int i = 0; i < NW; ++i) w[y][i] = 0u;
    }
#pragma unroll
    for (int j = 0; j < WIN; ++j) {
        float col[kCtRows + WIN - 1];
#pragma unroll
        for (int i = 0; i < kCtRows + WIN - 1; ++i) col[i] = tile[ty + i][tx + j];
#pragma unroll
        for (int y = 0; y < kCtRows; ++y)
#pragma unroll
            for (int i = 0; i < WIN; ++i) {
                const int b = i * WIN + j;
                if (col[y + i] > ctr[y]) w[y][b >> 5] |= (1u << (b & 31));
            }
    }
    const bool col_in = c >= O && c < Wd - O;
#pragma unroll
    for (int y = 0; y < kCtRows; ++y) {
        const int r = r0 + ty + y;
        if (r >= H) break;
        const bool in = col_in && r >= O && r < H - O;
#pragma unroll
        for (int i = 0; i < NW; ++i) codes[((size_t)r * Wd + c) * NW + i] = in ? w[y][i] : 0u;
    }
}

// ---- census Hamming cost (census.cpp:97-180) ----------------------------------------------------
// One thread per 4 consecutive cells of a row (coalesced 16-B stores, disparity innermost).
// Reads: left code once per pixel (broadcast), right codes of consecutive columns (L1-resident).
template <int NW>
struct code_ptrs {
    const uint32_t* left;
    const uint32_t* right[PMX_MAX_SUBPIX];
};

template <int NW>
__global__ __launch_bounds__(kBlock) void census_cost_kernel(pmx_mc_params p, code_ptrs<NW> cp, float* __restrict__ cv) {
    const int r = blockIdx.y;
    const int o = p.win / 2;
    const size_t row_base = (size_t)r * p.W * p.D;
    const int row_cells = p.W * p.D;
    const int mis = (int)(row_base & 3);  // float4 alignment of this row inside the volume
    int j0 = (blockIdx.x * kBlock + threadIdx.x) * 4 - mis;
    if (j0 >= row_cells) return;
    float v[4];
    int c = 0, k = 0;
    bool started = false;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float val = qnan();
        int j = j0 + e;
        if (j >= 0 && j < row_cells) {
            if (!started) {
                c = j / p.D;
                k = j - c * p.D;
                started = true;
            }
            cell_geom g = cell_geometry(p.H, p.W, p.d0, p.subpix, o, r, c, k);
            if (g.ok) {
                int wk = g.ph == 0 ? p.W : p.W - 1;
                const uint32_t* lc = cp.left + ((size_t)r * p.W + c) * NW;
                const uint32_t* rc = cp.right[g.ph] + ((size_t)r * wk + g.q) * NW;
                int w = 0;
#pragma unroll
                for (int i = 0; i < NW; ++i) w += __popc(lc[i] ^ rc[i]);
                val = (float)w;
                if (p.apply_mask && cell_masked(p, r, c, k, g.ph, g.q)) val = qnan();
            }
            if (++k == p.D) { k = 0; ++c; }
        }
        v[e] = val;
    }
    float* out = cv + row_base;
    if (j0 >= 0 && j0 + 3 < row_cells) {
        *reinterpret_cast<float4*>(out + j0) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (j0 + e >= 0 && j0 + e < row_cells) out[j0 + e] = v[e];
    }
}

// subpix == 1 fast variant: four pixels per wavefront (one per 16-lane row).  Lane `sub` owns, for each
// block q < NB, the four disparities 64q + 4*sub + {0..3}: every 16-byte load of right codes and every
// 16-byte store of costs is lane-contiguous (256 B per row per instruction), and there are few of them -
// the texture addresser charges per instruction.  Codes carry kCodePad guard dwords on both sides.
template <int NW, int NB>
__global__ __launch_bounds__(kBlock) void census_cost4_kernel(pmx_mc_params p, const uint32_t* __restrict__ codeL,
                                                               const uint32_t* __restrict__ codeR, float* __restrict__ cv) {
    const int lane = threadIdx.x & 63;
    const int sub = lane & 15, grp = lane >> 4;
    const size_t npix = (size_t)p.H * p.W;
    const size_t wave = (size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const size_t nwaves = (size_t)gridDim.x * (kBlock / 64);
    const int o = p.win / 2;
    const uint32_t wvalid = (uint32_t)(p.W - 2 * o);
    for (size_t quad = wave; quad * 4 < npix; quad += nwaves) {
        const size_t pix = min(quad * 4 + grp, npix - 1);  // surplus rows repeat the last pixel (same values)
        const int r = (int)(pix / p.W), c = (int)(pix - (size_t)r * p.W);
        uint32_t lc[NW];
        __builtin_memcpy(lc, codeL + pix * NW, sizeof(uint32_t) * NW);
        const bool pix_ok = (r >= o) && (r < p.H - o) && (c >= o) && (c < p.W - o);
        float* const dst = cv + pix * (size_t)p.D;
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int d_first = 64 * q + 4 * sub;
            if (d_first < p.D) {  // (wave-uniform for all blocks but the last)
                uint32_t rc[4 * NW];
                __builtin_memcpy(rc, codeR + ((ptrdiff_t)pix + p.d0 + d_first) * NW, sizeof(uint32_t) * 4 * NW);
                const uint32_t u = (uint32_t)(c + p.d0 + d_first - o);  // cell e valid iff u + e < wvalid (unsigned)
                float out[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int w = 0;
#pragma unroll
                    for (int i = 0; i < NW; ++i) w += __popc(lc[i] ^ rc[e * NW + i]);
                    float val = (pix_ok && (u + (uint32_t)e < wvalid)) ? (float)w : qnan();
                    if (p.apply_mask && val == val && cell_masked(p, r, c, d_first + e, 0, c + p.d0 + d_first + e)) val = qnan();
                    out[e] = val;
                }
                if (d_first + 4 <= p.D) {
                    __builtin_memcpy(dst + d_first, out, sizeof(float) * 4);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (d_first + e < p.D) dst[d_first + e] = out[e];
                }
            }
        }
    }
}

static constexpr size_t kCodePad = 1024;  // dwords readable before/after each code image, at least (fast path over-reads)

// The kernels that take a pixel's right words through raw pointers (census_cost4_kernel here; census_cost_u8_kernel and the path
// kernel of k_fused.hip on the integer path) read NW words per disparity at (pixel + d0 + d) NW for every d of the lane maps' padded
// range, whether the cell exists or not: the guards have to cover the whole disparity range on both sides, in words of THIS census
// window.  (Until round 6 the guard was 1024 dwords whatever the range: a 13 x 13 window - six words - with d = [0, 256] read up
// to 2.5 KB behind the allocation on the image's last rows; the loaded words belong to cells that are not numbers, so nothing was
// ever wrong, but the addresses were not ours.)
static size_t code_pad(const pmx_cv* cv, int nw) {
    const size_t need = ((size_t)abs(cv->d0) + (size_t)cv->D + 64) * (size_t)nw;
    return need <= kCodePad ? kCodePad : (need + kCodePad - 1) / kCodePad * kCodePad;
}

// census codes of the resident pair into buffers owned by the volume handle
template <int WIN>
static int census_codes(pmx_ctx* ctx, pmx_cv* cv) {
    constexpr int NW = (WIN * WIN + 31) / 32;
    const int H = cv->H, W = cv->W;
    const size_t per_img = (size_t)H * W * NW;
    const size_t pad = code_pad(cv, NW);
    const size_t total = (pad + per_img) * (1 + (size_t)cv->subpix) + pad;
    if (cv->codes_bytes < total * sizeof(uint32_t)) {
        PMX_HIP(hipStreamSynchronize(ctx->stream));
        pmx_pool_free(ctx, cv->codes);
        cv->codes = nullptr;
        cv->codes_bytes = 0;
        PMX_HIP(pmx_pool_alloc(ctx, (void**)&cv->codes, total * sizeof(uint32_t)));
        cv->codes_bytes = total * sizeof(uint32_t);
        PMX_HIP(hipMemsetAsync(cv->codes, 0, total * sizeof(uint32_t), ctx->stream));
    }
    uint32_t* left = cv->codes + pad;
    cv->codeL = left;
    cv->codeR = left + per_img + pad;
    cv->win = WIN;
    pmx_stage_scope t(ctx, PMX_STAGE_CENSUS_TRANSFORM);
    dim3 grid((W + 63) / 64, (H + kCtTileY - 1) / kCtTileY);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(census_transform_kernel<WIN>), grid, dim3(kBlock), 0, ctx->stream, ctx->left, H, W, left);
    for (int k = 0; k < cv->subpix; ++k) {
        uint32_t* dst = left + (per_img + pad) * (size_t)(k + 1);
        int wk = pmx_shifted_width(W, k);
        dim3 g2((wk + 63) / 64, (H + kCtTileY - 1) / kCtTileY);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(census_transform_kernel<WIN>), g2, dim3(kBlock), 0, ctx->stream, ctx->right[k], H, wk, dst);
    }
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

template <int NW>
static int census_costs(pmx_ctx* ctx, pmx_cv* cv) {
    const int H = cv->H, W = cv->W;
    const size_t per_img = (size_t)H * W * NW;
    code_ptrs<NW> cp;
    cp.left = cv->codeL;
    const size_t pad = code_pad(cv, NW);
    for (int k = 0; k < PMX_MAX_SUBPIX; ++k) cp.right[k] = k < cv->subpix ? cv->codeL + (per_img + pad) * (size_t)(k + 1) : nullptr;
    pmx_mc_params p = make_params(ctx, cv, cv->win);
    pmx_stage_scope t(ctx, PMX_STAGE_CENSUS_COST);
    if (cv->subpix == 1 && cv->D <= 512 && abs(cv->d0) + cv->D <= (int)kCodePad / NW - 64) {
        size_t want = ((size_t)H * W + 15) / 16;
        int grid = (int)(want < 65536 ? want : 65536);
        const int nb = (cv->D + 63) / 64;
#define PMX_CC4(NBV) \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(census_cost4_kernel<NW, NBV>), dim3(grid), dim3(kBlock), 0, ctx->stream, p, cv->codeL, cv->codeR, cv->data)
        switch (nb) {
            case 1: PMX_CC4(1); break;
            case 2: PMX_CC4(2); break;
            case 3: PMX_CC4(3); break;
            case 4: PMX_CC4(4); break;
            case 5: PMX_CC4(5); break;
            case 6: PMX_CC4(6); break;
            case 7: PMX_CC4(7); break;
            default: PMX_CC4(8); break;
        }
#undef PMX_CC4
        PMX_HIP(hipGetLastError());
        cv->repr = PMX_REPR_FLOAT;
        return PMX_OK;
    }
    int threads_per_row = (W * cv->D + 3) / 4 + 1;
    dim3 grid((threads_per_row + kBlock - 1) / kBlock, H);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(census_cost_kernel<NW>), grid, dim3(kBlock), 0, ctx->stream, p, cp, cv->data);
    PMX_HIP(hipGetLastError());
    cv->repr = PMX_REPR_FLOAT;
    return PMX_OK;
}

int pmx_launch_census_costs(pmx_ctx* ctx, pmx_cv* cv) {
    switch ((cv->win * cv->win + 31) / 32) {
        case 1: return census_costs<1>(ctx, cv);
        case 2: return census_costs<2>(ctx, cv);
        case 3: return census_costs<3>(ctx, cv);
        case 4: return census_costs<4>(ctx, cv);
        case 6: return census_costs<6>(ctx, cv);
    }
    pmx_set_error("census: unsupported window %d", cv->win);
    return PMX_ERR_ARG;
}

int pmx_launch_census(pmx_ctx* ctx, pmx_cv* cv, int win, bool defer_costs) {
    int rc;
    switch (win) {
        case 3: rc = census_codes<3>(ctx, cv); break;
        case 5: rc = census_codes<5>(ctx, cv); break;
        case 7: rc = census_codes<7>(ctx, cv); break;
        case 9: rc = census_codes<9>(ctx, cv); break;
        case 11: rc = census_codes<11>(ctx, cv); break;
        case 13: rc = census_codes<13>(ctx, cv); break;
        default:
            pmx_set_error("pmx_census: unsupported window %d", win);
            return PMX_ERR_ARG;
    }
    if (rc) return rc;
    if (defer_costs) {
        cv->repr = PMX_REPR_CENSUS_DEFERRED;
        return PMX_OK;
    }
    return pmx_launch_census_costs(ctx, cv);
}

// ---- cv_masked as its own pass (only launched when masks or grids are resident) ----------------
__global__ __launch_bounds__(kBlock) void cv_masked_kernel(pmx_mc_params p, float* __restrict__ cv) {
    const int r = blockIdx.y;
    int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= p.W * p.D) return;
    int c = j / p.D, k = j - c * p.D;
    int kk = k / p.subpix;
    int ph = k - kk * p.subpix;
    int q = c + p.d0 + kk;
    if (cell_masked(p, r, c, k, ph, q)) cv[(size_t)r * p.W * p.D + j] = qnan();
}

int pmx_launch_cv_masked(pmx_ctx* ctx, pmx_cv* cv, int win) {
    if (!ctx->bad_left && !ctx->bad_right && !ctx->grid_min) return PMX_OK;  // NaN pattern already complete
    pmx_mc_params p = make_params(ctx, cv, win);
    pmx_stage_scope t(ctx, PMX_STAGE_MASK);
    dim3 grid((cv->W * cv->D + kBlock - 1) / kBlock, cv->H);
    hipLaunchKernelGGL(cv_masked_kernel, grid, dim3(kBlock), 0, ctx->stream, p, cv->data);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

// ---- SAD / SSD (sad_ssd.py:146-207, 226-368) ---------------------------------------------------
// float32 window sum in the reference's order: window columns outer, rows inner, sequential.
__global__ __launch_bounds__(kBlock) void sad_ssd_kernel(pmx_mc_params p, int squared, float* __restrict__ cv) {
    const int r = blockIdx.y;
    int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= p.W * p.D) return;
    const int o = p.win / 2;
    int c = j / p.D, k = j - c * p.D;
    cell_geom g = cell_geometry(p.H, p.W, p.d0, p.subpix, o, r, c, k);
    float val = qnan();
    if (g.ok) {
        int wk = g.ph == 0 ? p.W : p.W - 1;
        const float* R = p.right[g.ph];
        float s = 0.f;
        for (int jj = -o; jj <= o; ++jj)  // window columns outer, rows inner: numpy's order (sad_ssd.py:367)
            for (int i = -o; i <= o; ++i) {
                float d = p.left[(size_t)(r + i) * p.W + c + jj] - R[(size_t)(r + i) * wk + g.q + jj];
                s = s + (squared ? d * d : fabsf(d));
            }
        val = s;
    }
    cv[(size_t)r * p.W * p.D + j] = val;
}

// subpix == 1, window <= 7: four pixels per wavefront (one per 16-lane row), lane `sub` owns the four disparities
// 64q + 4*sub + {0..3} of block q.  The WIN x WIN left window and the WIN x (WIN+3) right window of a block live in
// registers (raw buffer loads: an offset before the first / past the last row returns 0, and such cells are NaN
// anyway), so a term costs two instructions; the float32 sum runs in the reference's order (window columns outer,
// rows inner, sad_ssd.py:367) and stays bit-exact.  Costs leave as lane-contiguous 16-byte stores.
typedef uint32_t su32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t su32x2 __attribute__((ext_vector_type(2)));
// N consecutive floats at byte offset `off` of a raw buffer, as few loads as possible (16 / 8 / 4 bytes)
template <int N>
__device__ __forceinline__ void buf_load_row(__amdgpu_buffer_rsrc_t rs, uint32_t off, float (&dst)[N]) {
    int k = 0;
#pragma unroll
    for (; k + 4 <= N; k += 4) {
        su32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 4 * k, 0, 0);
        dst[k] = __uint_as_float(t.x); dst[k + 1] = __uint_as_float(t.y); dst[k + 2] = __uint_as_float(t.z); dst[k + 3] = __uint_as_float(t.w);
    }
    if (k + 2 <= N) {
        su32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, off + 4 * k, 0, 0);
        dst[k] = __uint_as_float(t.x); dst[k + 1] = __uint_as_float(t.y);
        k += 2;
    }
    if (k < N) dst[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off + 4 * k, 0, 0));
}

template <int WIN, bool SQUARED>
__global__ __launch_bounds__(kBlock) void sad_ssd_window_kernel(pmx_mc_params p, uint32_t img_bytes, float* __restrict__ cv) {
    constexpr int O = WIN / 2, RW = WIN + 3;  // right columns a lane touches per row
    const int lane = threadIdx.x & 63;
    const int sub = lane & 15, grp = lane >> 4;
    const size_t npix = (size_t)p.H * p.W;
    const size_t wave = (size_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const size_t nwaves = (size_t)gridDim.x * (kBlock / 64);
    const uint32_t wvalid = (uint32_t)(p.W - 2 * O);
    const int nblk = (p.D + 63) / 64;
    // the resources start at the images' zeroed guards (pmx_api img_alloc): a wide load that runs a few elements past
    // the last row must not be range-checked away as a whole
    constexpr uint32_t G = (uint32_t)kImgGuardBytes;
    const __amdgpu_buffer_rsrc_t rsL = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.left - G), 0, img_bytes + 2 * G, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.right[0] - G), 0, img_bytes + 2 * G, 0x00020000);
    for (size_t quad = wave; quad * 4 < npix; quad += nwaves) {
        const size_t pix = min(quad * 4 + grp, npix - 1);  // surplus rows repeat the last pixel (same values)
        const int r = (int)(pix / p.W), c = (int)(pix - (size_t)r * p.W);
        const bool pix_ok = (r >= O) & (r < p.H - O) & (c >= O) & (c < p.W - O);
        float lw[WIN][WIN];
#pragma unroll
        for (int i = 0; i < WIN; ++i) buf_load_row<WIN>(rsL, (uint32_t)(((r + i - O) * p.W + c - O) * 4) + G, lw[i]);
        float* const dst = cv + pix * (size_t)p.D;
        for (int q = 0; q < nblk; ++q) {
            const int d_first = 64 * q + 4 * sub;
            if (d_first >= p.D) continue;  // (whole groups of lanes in the last block)
            float rw[WIN][RW];
#pragma unroll
            for (int i = 0; i < WIN; ++i) buf_load_row<RW>(rsR, (uint32_t)(((r + i - O) * p.W + c + p.d0 + d_first - O) * 4) + G, rw[i]);
            const uint32_t u = (uint32_t)(c + p.d0 + d_first - O);  // cell e valid iff u + e < wvalid (unsigned)
            float out[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < WIN; ++j)
#pragma unroll
                    for (int i = 0; i < WIN; ++i) {
                        const float d = lw[i][j] - rw[i][j + e];
                        s = s + (SQUARED ? d * d : fabsf(d));
                    }
                out[e] = (pix_ok && (u + (uint32_t)e < wvalid)) ? s : qnan();
            }
            if (d_first + 4 <= p.D) {
                __builtin_memcpy(dst + d_first, out, sizeof(float) * 4);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (d_first + e < p.D) dst[d_first + e] = out[e];
            }
        }
    }
}

int pmx_launch_sad_ssd(pmx_ctx* ctx, pmx_cv* cv, int win, int squared) {
    pmx_mc_params p = make_params(ctx, cv, win);
    pmx_stage_scope t(ctx, PMX_STAGE_SAD_SSD);
    const size_t img_bytes = (size_t)cv->H * cv->W * 4;
    if (cv->subpix == 1 && win <= 7 && img_bytes < (1ull << 31)) {
        size_t want = ((size_t)cv->H * cv->W + 15) / 16;  // 4 pixels per wave, 4 waves per block
        dim3 grid((unsigned)(want < 65536 ? want : 65536));
#define PMX_SAD_CASE(WN)                                                                                                        \
    case WN:                                                                                                                    \
        if (squared) hipLaunchKernelGGL(HIP_KERNEL_NAME(sad_ssd_window_kernel<WN, true>), grid, dim3(kBlock), 0, ctx->stream, p, \
                                        (uint32_t)img_bytes, cv->data);                                                        \
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(sad_ssd_window_kernel<WN, false>), grid, dim3(kBlock), 0, ctx->stream, p,       \
                                (uint32_t)img_bytes, cv->data);                                                                \
        break;
        switch (win) {
            PMX_SAD_CASE(1) PMX_SAD_CASE(3) PMX_SAD_CASE(5) PMX_SAD_CASE(7)
        }
#undef PMX_SAD_CASE
    } else {
        dim3 grid((cv->W * cv->D + kBlock - 1) / kBlock, cv->H);
        hipLaunchKernelGGL(sad_ssd_kernel, grid, dim3(kBlock), 0, ctx->stream, p, squared, cv->data);
    }
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

// ---- ZNCC (zncc.py:153-277 + img_tools.py:834-952) ---------------------------------------------
// Window statistics in float64 (the reference's integral images are float64): mean and std of
// every full window, std with the float32 squares and the 1e-15 clip of img_tools.py:941-951.
// A thread owns one column of the raster and a strip of output rows (32; 8 for small images, which need the threads more): the
// sums of the window's ROWS wait in a register ring while the window moves down the strip - win + win^2 / strip loads per pixel
// instead of win^2 (11 x 11 at 4096^2: 0.62 -> 0.19 ms per image).
template <int WIN_T>  // window known at compile time (0: any): the rows' sums wait in a register ring for the step at which they leave
__global__ __launch_bounds__(kBlock) void window_stats_kernel(const float* __restrict__ img, int H, int Wd, int win_rt, int strip,
                                                              double* __restrict__ mean, double* __restrict__ sd,
                                                              double* __restrict__ isd) {
    const int win = WIN_T ? WIN_T : win_rt;
    const int o = win / 2, Wo = Wd - 2 * o, Ho = H - 2 * o;
    const int c = blockIdx.x * kBlock + threadIdx.x;
    const int r0 = blockIdx.y * strip, r1 = min(r0 + strip, Ho);
    if (c >= Wo) return;
    auto row_sums = [&](int row, double& s, double& s2) {
        const float* p = img + (size_t)row * Wd + c;
        double a = 0, a2 = 0;
        if (WIN_T) {
            float x[WIN_T ? WIN_T : 1];
#pragma unroll
            for (int j = 0; j < WIN_T; ++j) x[j] = p[j];
#pragma unroll
            for (int j = 0; j < WIN_T; ++j) {
                const float x2 = x[j] * x[j];
                a += (double)x[j];
                a2 += (double)x2;
            }
        } else {
            for (int j = 0; j < win; ++j) {
                const float x = p[j];
                const float x2 = x * x;
                a += (double)x;
                a2 += (double)x2;
            }
        }
        s = a;
        s2 = a2;
    };
    const double n = (double)win * win;
    auto emit = [&](int r, double s, double s2) {
        double m = s / n, m2 = s2 / n;
        double var = m2 - m * m;
        if (var < 1e-15 * fabs(m2)) var = 0;
        const double s_ = sqrt(var);
        mean[(size_t)r * Wo + c] = m;
        sd[(size_t)r * Wo + c] = s_;
        isd[(size_t)r * Wo + c] = s_ > 0 ? 1.0 / s_ : 0.0;  // marching kernel: z = cov * isd_L * isd_R (0 when a std is 0)
    };
    if (WIN_T) {
        // no sliding difference: a sum that has held a bright row keeps that row's rounding (float32 squares of values four
        // decimal orders apart do not add exactly in float64), and the 1e-15 test for std = 0 would see it rows later.  Every
        // output sums the WIN_T row sums it consists of, as the direct form sums its pixels.
        double ring[WIN_T ? WIN_T : 1], ring2[WIN_T ? WIN_T : 1];  // sums of image rows r0 + k (mod WIN_T)
#pragma unroll
        for (int k = 0; k < WIN_T; ++k) row_sums(r0 + k, ring[k], ring2[k]);
        auto total = [&](int r) {
            double s = ring[0], s2 = ring2[0];
#pragma unroll
            for (int k = 1; k < WIN_T; ++k) {
                s += ring[k];
                s2 += ring2[k];
            }
            emit(r, s, s2);
        };
        total(r0);
        for (int rb = r0 + 1; rb < r1; rb += WIN_T) {
#pragma unroll
            for (int k = 0; k < WIN_T; ++k) {  // output row rb + k: image row rb + k + WIN_T - 1 takes the slot of row rb + k - 1
                const int r = rb + k;
                if (r < r1) {
                    row_sums(r + WIN_T - 1, ring[k], ring2[k]);
                    total(r);
                }
            }
        }
    } else {
        for (int r = r0; r < r1; ++r) {
            double s = 0, s2 = 0;
            for (int i = 0; i < win; ++i) {
                double a, a2;
                row_sums(r + i, a, a2);
                s += a;
                s2 += a2;
            }
            emit(r, s, s2);
        }
    }
}

static void launch_window_stats(pmx_ctx* ctx, const float* img, int H, int Wd, int win, double* mean, double* sd, double* isd) {
    const int o = win / 2;
    const int strip = (size_t)H * Wd >= ((size_t)4 << 20) ? 32 : 8;  // output rows per thread (small images need the threads more)
    dim3 grid((Wd - 2 * o + kBlock - 1) / kBlock, (H - 2 * o + strip - 1) / strip);
    switch (win) {
#define PMX_STATS_CASE(WN) case WN: hipLaunchKernelGGL(window_stats_kernel<WN>, grid, dim3(kBlock), 0, ctx->stream, img, H, Wd, win, strip, mean, sd, isd); break;
        PMX_STATS_CASE(3) PMX_STATS_CASE(5) PMX_STATS_CASE(7) PMX_STATS_CASE(9) PMX_STATS_CASE(11) PMX_STATS_CASE(13)
#undef PMX_STATS_CASE
        default: hipLaunchKernelGGL(window_stats_kernel<0>, grid, dim3(kBlock), 0, ctx->stream, img, H, Wd, win, strip, mean, sd, isd); break;
    }
}

struct zncc_stats {
    const double* lmean;
    const double* lsd;
    const double* rmean[PMX_MAX_SUBPIX];
    const double* rsd[PMX_MAX_SUBPIX];
};

__global__ __launch_bounds__(kBlock) void zncc_kernel(pmx_mc_params p, zncc_stats st, float* __restrict__ cv) {
    const int r = blockIdx.y;
    int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= p.W * p.D) return;
    const int o = p.win / 2;
    int c = j / p.D, k = j - c * p.D;
    cell_geom g = cell_geometry(p.H, p.W, p.d0, p.subpix, o, r, c, k);
    float val = qnan();
    if (g.ok) {
        int wk = g.ph == 0 ? p.W : p.W - 1;
        const float* R = p.right[g.ph];
        double s = 0;
        for (int i = -o; i <= o; ++i) {
            const float* lrow = p.left + (size_t)(r + i) * p.W + c;
            const float* rrow = R + (size_t)(r + i) * wk + g.q;
            for (int jj = -o; jj <= o; ++jj) {
                float prod = lrow[jj] * rrow[jj];  // float32 product (zncc.py:209-212)
                s += (double)prod;
            }
        }
        double z = s / ((double)p.win * p.win);
        size_t il = (size_t)(r - o) * (p.W - 2 * o) + (c - o);
        size_t ir = (size_t)(r - o) * (wk - 2 * o) + (g.q - o);
        z -= st.lmean[il] * st.rmean[g.ph][ir];
        double dv = st.lsd[il] * st.rsd[g.ph][ir];
        z = dv > 0 ? z / dv : 0.0;
        val = (float)z;
    }
    cv[(size_t)r * p.W * p.D + j] = val;
}

// ---- ZNCC, subpix == 1: marching kernel ---------------------------------------------------------
// The reference gets the window mean of L*R_d from float64 integral images (img_tools.py:834-879), so any
// float64 summation order is inside its own rounding noise; the contract is 1e-5 on the float32 result.
// That licenses a separable, sliding evaluation that costs O(win) per cell instead of O(win^2):
//   * a wavefront owns 64 adjacent columns (lane = column, 2o of them halo), kZnccND adjacent disparities and a
//     strip of rows; the four wavefronts of a workgroup take four adjacent disparity chunks of the same columns;
//   * every lane keeps colsum[e] = sum over the 2o+1 window ROWS of (double)(L*R) for its column in registers
//     and slides it down the strip: + the row entering the window, - the row leaving it (2 products per cell);
//   * the 2o+1 window COLUMNS are summed from the neighbour lanes' colsums through a wave-private LDS tile
//     (conflict-free b64 reads, no workgroup barrier);
//   * z = (box/n - mean_L*mean_R) * isd_L * isd_R, right-image statistics staged through LDS once per row;
//   * the four chunks' results meet in LDS and leave as 128-byte runs per pixel (lane = 16 bytes of a pixel's
//     32 disparities).  Lane-per-column stores of 32 bytes at a 4*D-byte stride cost 2x the whole kernel.
// Loads are raw buffer loads: an offset outside the image returns 0 (rows above/below the image contribute
// nothing), and offsets that run over a row end only ever feed cells that are NaN anyway.
constexpr int kZnccND = 8;                      // disparities per lane
constexpr int kZnccWaves = 4;                   // wavefronts (= disparity chunks) per workgroup
constexpr int kZnccDB = kZnccND * kZnccWaves;   // disparities per workgroup
constexpr int kZnccOutStride = kZnccDB + 4;     // floats per staged pixel row (16-byte aligned, skewed banks)

struct zncc_march_params {
    const float* left;
    const float* right[PMX_MAX_SUBPIX];                   // shifted right images (phase k > 0 is one column narrower)
    const double *lmean, *lisd;
    const double *rmean[PMX_MAX_SUBPIX], *risd[PMX_MAX_SUBPIX];
    float* cv;
    int H, W, D, d0, win;
    int ntile, ndblock, nstrip, strip_rows;
    uint32_t img_bytes, stat_bytes;                       // of the full-width images; phase > 0: H * (W-1) * 4 etc.
};

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t zu32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t zu32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double buf_load_f64(__amdgpu_buffer_rsrc_t rs, uint32_t byte_off) {
    zu32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, byte_off, 0, 0);
    return __hiloint2double((int)v.y, (int)v.x);
}

// SUBPIX 2 / 4 (img_tools.py:713-752 shifted right images, cost index k = shift * subpix + phase): the four wavefronts of a
// workgroup still produce 32 consecutive cost indices of the same columns, but as phases x integer shifts - subpix 4: wave =
// phase, 8 shifts; subpix 2: 2 phases x 2 chunks of 8 shifts - each wave marching over ITS phase's right image and statistics.
template <int WIN_T, int SUBPIX>  // window known at compile time (0: any)
__global__ __launch_bounds__(64 * kZnccWaves) void zncc_march_kernel(zncc_march_params q) {
    constexpr int ND = kZnccND;
    // The window's COLUMNS: 8 x 7 (or 8 x 8) lanes of the wavefront each take one disparity and 8 adjacent output columns, read
    // the 8 + WIN - 1 column sums they span (16 bytes at a time), SLIDE the window sum along them in registers (WIN - 1 additions
    // for the first, two for each of the other seven) and leave the 8 sums where the output lanes pick them up: 18 doubles read
    // and 24 additions for 8 windows of 11 instead of 88 and 80.  (kColStride: room for the last segment's reads, rows of 16-byte
    // multiples, disparities in different banks.)
    constexpr bool kJob = WIN_T >= 3 && WIN_T <= 13;
    constexpr int kColStride = kJob ? 70 : 64;
    __shared__ __attribute__((aligned(16))) double colbuf[kZnccWaves][ND][kColStride];
    __shared__ double rstat[kZnccWaves][2][64 + ND];  // [wave][mean|isd][column]
    __shared__ __attribute__((aligned(16))) float ostage[2][64][kZnccOutStride];

    // Workgroup -> (tile, strip, disparity block), XCD-aware: the disparity blocks of one (tile, strip) write the SAME cache lines of
    // the volume (a block's 128 bytes per pixel start at 4-byte alignment: every line of a pixel is shared by two blocks), and the
    // hardware deals workgroups to the eight XCDs round robin.  Workgroups 8 apart sit on one XCD: the ndblock blocks of a group take
    // indices x, x + 8, x + 16 .. so that the two halves of a line meet in ONE L2 instead of leaving two of them as partial lines.
    const uint32_t xcd = blockIdx.x & 7u, seq = blockIdx.x >> 3;
    const uint32_t group = (seq / (uint32_t)q.ndblock) * 8u + xcd;  // (tile, strip) pair
    if (group >= (uint32_t)(q.ntile * q.nstrip)) return;            // (the grid is rounded up to whole rounds of eight groups)
    const uint32_t logical = group * (uint32_t)q.ndblock + seq % (uint32_t)q.ndblock;
    const int dblock = logical % q.ndblock;
    const int tile = (logical / q.ndblock) % q.ntile;
    const int strip = logical / (q.ndblock * q.ntile);

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int win = WIN_T ? WIN_T : q.win;
    const int o = win / 2, W = q.W, H = q.H;
    const int wo = W - 2 * o;                      // width of the statistics rasters
    const int tile_c0 = tile * (64 - 2 * o);       // first output column of the tile
    const int c = tile_c0 - o + lane;              // image column of this lane (halo lanes included)
    const int ph = SUBPIX == 1 ? 0 : (SUBPIX == 2 ? (wv & 1) : wv);                                   // sub-pixel phase of this wave
    const int kk0 = SUBPIX == 1 ? dblock * kZnccDB + wv * ND : (SUBPIX == 2 ? dblock * 16 + (wv >> 1) * ND : dblock * ND);
    const bool chunk_live = kk0 * SUBPIX + ph < q.D;  // first cost index of this wavefront's chunk
    const int Wp = ph ? W - 1 : W;                 // width of this phase's right image
    const int wop = Wp - 2 * o;                    // ... and of its statistics rasters
    const int x0 = c + q.d0 + kk0;                 // right-image column matched at e = 0
    const int r0 = strip * q.strip_rows;
    const int r1 = min(r0 + q.strip_rows, H);

    const __amdgpu_buffer_rsrc_t rsL = __builtin_amdgcn_make_buffer_rsrc((void*)q.left, 0, q.img_bytes, 0x00020000);
    const uint32_t img_bytes_p = ph ? (uint32_t)((size_t)H * Wp * 4) : q.img_bytes;
    const uint32_t stat_bytes_p = ph ? (uint32_t)((size_t)(H - 2 * o) * wop * 8) : q.stat_bytes;
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)q.right[ph] - kImgGuardBytes), 0,
                                                                          img_bytes_p + 2 * (uint32_t)kImgGuardBytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsLM = __builtin_amdgcn_make_buffer_rsrc((void*)q.lmean, 0, q.stat_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsLI = __builtin_amdgcn_make_buffer_rsrc((void*)q.lisd, 0, q.stat_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsRM = __builtin_amdgcn_make_buffer_rsrc((void*)q.rmean[ph], 0, stat_bytes_p, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsRI = __builtin_amdgcn_make_buffer_rsrc((void*)q.risd[ph], 0, stat_bytes_p, 0x00020000);
    constexpr uint32_t kOob = 0xfffffff0u;

    // raw loads of one image row for this lane's ND cells (zeros outside the image rows)
    struct row_vals { float lv; float rv[ND]; };
    auto load_row = [&](int row, row_vals& v) {
        const bool in = chunk_live && (row >= 0) && (row < H);
        const uint32_t offL = in ? (uint32_t)((row * W + c) * 4) : kOob;
        const uint32_t offR = in ? (uint32_t)((row * Wp + x0) * 4 + (int)kImgGuardBytes) : kOob;  // (guarded: pmx_api img_alloc)
        v.lv = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsL, offL, 0, 0));
#pragma unroll
        for (int i = 0; i < ND / 4; ++i) {
            // (__uint_as_float, not __builtin_bit_cast(float, t.y): this clang reads element 0 for every
            //  bit_cast of a vector-element lvalue)
            zu32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rsR, offR + 16 * i, 0, 0);
            v.rv[4 * i + 0] = __uint_as_float(t.x);
            v.rv[4 * i + 1] = __uint_as_float(t.y);
            v.rv[4 * i + 2] = __uint_as_float(t.z);
            v.rv[4 * i + 3] = __uint_as_float(t.w);
        }
    };
    // statistics of output row r: left for the lane's own pixel, right for columns x0 (and x0+64 for lane < ND)
    struct stat_vals { double mL, iL, rm0, ri0, rm1, ri1; };
    const bool col_ok = (c >= o) && (c < W - o);
    auto load_stats = [&](int r, stat_vals& s) {
        const bool row_ok = chunk_live && (r >= o) && (r < H - o);
        const uint32_t srow = row_ok ? (uint32_t)((r - o) * wo) : 0u;
        const uint32_t offLs = (row_ok && col_ok) ? (srow + (uint32_t)(c - o)) * 8u : kOob;
        s.mL = buf_load_f64(rsLM, offLs);
        s.iL = buf_load_f64(rsLI, offLs);
        const int xs = x0 - o;  // statistics column of right pixel x0
        const uint32_t srowp = row_ok ? (uint32_t)((r - o) * wop) : 0u;
        const bool ok0 = row_ok && (xs >= 0) && (xs < wop);
        const uint32_t off0 = ok0 ? (srowp + (uint32_t)xs) * 8u : kOob;
        s.rm0 = buf_load_f64(rsRM, off0);
        s.ri0 = buf_load_f64(rsRI, off0);
        const int xe = xs + 64;  // the ND columns past the tile's last lane
        const bool ok1 = (lane < ND) && row_ok && (xe >= 0) && (xe < wop);
        const uint32_t off1 = ok1 ? (srowp + (uint32_t)xe) * 8u : kOob;
        s.rm1 = buf_load_f64(rsRM, off1);
        s.ri1 = buf_load_f64(rsRI, off1);
    };

    double colsum[ND];
#pragma unroll
    for (int e = 0; e < ND; ++e) colsum[e] = 0.0;
    for (int i = -o; i < o; ++i) {  // prime with the 2o rows above the first window's last row
        row_vals v;
        load_row(r0 + i, v);
#pragma unroll
        for (int e = 0; e < ND; ++e) colsum[e] += (double)(v.lv * v.rv[e]);  // float32 product (zncc.py:209-212)
    }

    const double inv_n = 1.0 / ((double)win * win);
    const bool out_lane = (lane >= o) && (lane < 64 - o);
    // cooperative store: thread -> (pixel of the tile, 16 bytes of its kZnccDB disparities), two passes
    constexpr int kParts = kZnccDB / 4;                    // float4 pieces per pixel
    constexpr int kPixPerPass = 64 * kZnccWaves / kParts;  // pixels stored per pass
    const int st_part = tid % kParts, st_pix0 = tid / kParts;
    const int st_k = dblock * kZnccDB + 4 * st_part;

    int par = 0;
    row_vals vin, vout;
    stat_vals sv;
    load_row(r0 + o, vin);
    load_row(-1, vout);
    load_stats(r0, sv);
    for (int r = r0; r < r1; ++r, par ^= 1) {
        // the next row's operands are requested before this row's arithmetic so their latency hides behind it
        row_vals nin, nout;
        stat_vals ns;
        load_row(r + 1 + o, nin);
        load_row(r - o, nout);
        load_stats(r + 1, ns);
        const bool row_ok = (r >= o) && (r < H - o);
        const double mL = sv.mL, iL = sv.iL;
        rstat[wv][0][lane] = sv.rm0;
        rstat[wv][1][lane] = sv.ri0;
        if (lane < ND) {
            rstat[wv][0][64 + lane] = sv.rm1;
            rstat[wv][1][64 + lane] = sv.ri1;
        }
#pragma unroll
        for (int e = 0; e < ND; ++e) {
            colsum[e] = (colsum[e] - (double)(vout.lv * vout.rv[e])) + (double)(vin.lv * vin.rv[e]);
            colbuf[wv][e][lane] = colsum[e];
        }
        vin = nin; vout = nout; sv = ns;
        // colbuf / rstat are private to the wavefront: LDS executes its instructions in order, a wave-level
        // fence keeps the compiler from moving the reads above the writes
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        constexpr int kShift = kJob ? 2 - ((WIN_T / 2) & 1) : 0;  // where a window sum waits: column + kShift (16-byte aligned runs of 8)
        if (kJob) {
            constexpr int OJ = WIN_T / 2, NOUT = 64 - 2 * OJ, NSEG = (NOUT + 7) / 8, NRD = 8 + WIN_T - 1, NRD2 = (NRD + 1) / 2;
            if (lane < NSEG * 8) {
                typedef double zd2 __attribute__((ext_vector_type(2)));
                const int je = lane & 7, js = lane >> 3;
                double* row = &colbuf[wv][je][js * 8];
                double v[2 * NRD2];
#pragma unroll
                for (int i = 0; i < NRD2; ++i) {
                    const zd2 t = *reinterpret_cast<const zd2*>(row + 2 * i);
                    v[2 * i] = t.x;
                    v[2 * i + 1] = t.y;
                }
                // the segment's first window as a tree, the other seven slide to the right (so that the columns past the tile's 64,
                // which nobody wrote, only ever reach windows that are not output)
                double w8[8];
                double t[WIN_T > 0 ? WIN_T : 1];
#pragma unroll
                for (int i = 0; i < WIN_T; ++i) t[i] = v[i];
#pragma unroll
                for (int n = WIN_T; n > 1; n = (n + 1) / 2)
#pragma unroll
                    for (int i = 0; i < n / 2; ++i) t[i] = t[i] + t[n - 1 - i];
                w8[0] = t[0];
#pragma unroll
                for (int i = 1; i < 8; ++i) w8[i] = (w8[i - 1] - v[i - 1]) + v[i + WIN_T - 1];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    zd2 pr;
                    pr.x = w8[2 * i];
                    pr.y = w8[2 * i + 1];
                    *reinterpret_cast<zd2*>(row + OJ + kShift + 2 * i) = pr;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        if (out_lane) {
            float out[ND];
            double box[ND];
            if (kJob) {
#pragma unroll
                for (int e = 0; e < ND; ++e) box[e] = colbuf[wv][e][lane + kShift];
            } else {
            // SUBPIX 1: volatile, typed as LDS = plain ds_read_b64 (2 LDS cycles each).  Left alone, the compiler pairs neighbouring
            // columns into ds_read2_b64, which the LDS serves at half the rate (8 cycles for the two: MI355X_MICROARCH.md, LDS
            // table) - and the window columns are most of what this kernel asks of the LDS (11 x 11 at 4096^2 x 257: 13.0 -> 11.0 ms;
            // tools/ubench/zncc_windows.py).  The sub-pixel variants (more registers, other bounds) measured 2-4 % slower with it.
            typedef __attribute__((address_space(3))) double lds_double;
            typedef typename std::conditional<SUBPIX == 1, const volatile lds_double*, const lds_double*>::type colbuf_ptr;
            colbuf_ptr cb = (colbuf_ptr)&colbuf[wv][0][lane - o];
#pragma unroll
            for (int e = 0; e < ND; ++e) box[e] = cb[e * 64];
            if (WIN_T) {
#pragma unroll
                for (int j = 1; j < (WIN_T ? WIN_T : 1); ++j)
#pragma unroll
                    for (int e = 0; e < ND; ++e) box[e] += cb[e * 64 + j];
            } else {
                for (int j = 1; j < win; ++j)
#pragma unroll
                    for (int e = 0; e < ND; ++e) box[e] += cb[e * 64 + j];
            }
            }
#pragma unroll
            for (int e = 0; e < ND; ++e) {
                double z = box[e] * inv_n - mL * rstat[wv][0][lane + e];
                z = z * iL * rstat[wv][1][lane + e];
                const int x = x0 + e;
                const bool ok = row_ok && col_ok && (x - o >= 0) && (x + o < Wp);
                out[e] = ok ? (float)z : qnan();
            }
            if (SUBPIX == 1) {
                __builtin_memcpy(&ostage[par][lane][wv * ND], out, sizeof(float) * ND);
            } else {  // cost index inside the workgroup's 32: (shift) * SUBPIX + phase
#pragma unroll
                for (int e = 0; e < ND; ++e) ostage[par][lane][((SUBPIX == 2 ? (wv >> 1) * ND : 0) + e) * SUBPIX + ph] = out[e];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __syncthreads();  // the only workgroup barrier per row: ostage[par] is complete
#pragma unroll
        for (int pass = 0; pass < 64 / kPixPerPass; ++pass) {
            const int pix = st_pix0 + pass * kPixPerPass;  // lane index inside the tile
            const int pc = tile_c0 - o + pix;
            if (pix >= o && pix < 64 - o && pc < W && st_k < q.D) {
                f32x4 v = *reinterpret_cast<const f32x4*>(&ostage[par][pix][4 * st_part]);
                float* dst = q.cv + ((size_t)r * W + pc) * q.D + st_k;
                const int left = q.D - st_k;  // the volume's last piece of a pixel: one store of what is left
                if (left >= 4) __builtin_memcpy(dst, &v, 16);
                else if (left == 3) __builtin_memcpy(dst, &v, 12);
                else if (left == 2) __builtin_memcpy(dst, &v, 8);
                else dst[0] = v.x;
            }
        }
    }
}

int pmx_launch_zncc(pmx_ctx* ctx, pmx_cv* cv, int win) {
    const int H = cv->H, W = cv->W, o = win / 2;
    // (an image exactly one window wide still has ONE column of valid cells at the integer disparities: only its half-pixel
    //  phases, whose shifted right images are one column narrower, have none - their statistics rasters are empty)
    if (H - 2 * o <= 0 || W - 2 * o <= 0) return pmx_launch_fill_nan(ctx, cv->data, cv->cells());
    size_t per = (size_t)(H - 2 * o) * (W - 2 * o) * sizeof(double);
    int rc = pmx_need_small(ctx, per * 3 * (1 + cv->subpix));
    if (rc) return rc;
    char* base = (char*)ctx->small;
    zncc_stats st;
    st.lmean = (double*)base;
    st.lsd = (double*)(base + per);
    double* lisd = (double*)(base + per * 2);
    double* risd[PMX_MAX_SUBPIX];
    pmx_mc_params p = make_params(ctx, cv, win);
    pmx_stage_scope t(ctx, PMX_STAGE_ZNCC);
    launch_window_stats(ctx, ctx->left, H, W, win, (double*)st.lmean, (double*)st.lsd, lisd);
    for (int k = 0; k < PMX_MAX_SUBPIX; ++k) { st.rmean[k] = nullptr; st.rsd[k] = nullptr; risd[k] = nullptr; }
    for (int k = 0; k < cv->subpix; ++k) {
        int wk = pmx_shifted_width(W, k);
        st.rmean[k] = (double*)(base + per * (3 + 3 * k));
        st.rsd[k] = (double*)(base + per * (4 + 3 * k));
        risd[k] = (double*)(base + per * (5 + 3 * k));
        if (wk - 2 * o <= 0) continue;  // nothing of this phase is valid
        launch_window_stats(ctx, ctx->right[k], H, wk, win, (double*)st.rmean[k], (double*)st.rsd[k], risd[k]);
    }
    const bool march = (cv->subpix == 1 || cv->subpix == 2 || cv->subpix == 4) && 2 * o < 32 && (size_t)H * W * 4 < (1ull << 31) &&
                       (size_t)H * W * 8 < (1ull << 32);
    if (march) {
        zncc_march_params q;
        q.left = ctx->left;
        q.lmean = st.lmean; q.lisd = lisd;
        for (int k = 0; k < PMX_MAX_SUBPIX; ++k) {
            const int kk = k < cv->subpix ? k : 0;
            q.right[k] = ctx->right[kk]; q.rmean[k] = st.rmean[kk]; q.risd[k] = risd[kk];
        }
        q.cv = cv->data;
        q.H = H; q.W = W; q.D = cv->D; q.d0 = cv->d0; q.win = win;
        q.strip_rows = 64;
        q.ntile = (W + (64 - 2 * o) - 1) / (64 - 2 * o);
        q.ndblock = (cv->D + kZnccDB - 1) / kZnccDB;
        q.nstrip = (H + q.strip_rows - 1) / q.strip_rows;
        q.img_bytes = (uint32_t)((size_t)H * W * 4);
        q.stat_bytes = (uint32_t)per;
        const uint32_t grid = (((uint32_t)q.ntile * q.nstrip + 7u) / 8u) * 8u * (uint32_t)q.ndblock;
        const dim3 block(64 * kZnccWaves);
#define PMX_ZNCC_LAUNCH(WN)                                                                                                  \
    if (cv->subpix == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(zncc_march_kernel<WN, 1>), dim3(grid), block, 0, ctx->stream, q);     \
    else if (cv->subpix == 2) hipLaunchKernelGGL(HIP_KERNEL_NAME(zncc_march_kernel<WN, 2>), dim3(grid), block, 0, ctx->stream, q); \
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(zncc_march_kernel<WN, 4>), dim3(grid), block, 0, ctx->stream, q)
        switch (win) {
#define PMX_ZNCC_CASE(WN) case WN: PMX_ZNCC_LAUNCH(WN); break;
            PMX_ZNCC_CASE(1) PMX_ZNCC_CASE(3) PMX_ZNCC_CASE(5) PMX_ZNCC_CASE(7) PMX_ZNCC_CASE(9) PMX_ZNCC_CASE(11) PMX_ZNCC_CASE(13)
#undef PMX_ZNCC_CASE
            default: PMX_ZNCC_LAUNCH(0); break;
        }
#undef PMX_ZNCC_LAUNCH
    } else {
        dim3 grid((W * cv->D + kBlock - 1) / kBlock, H);
        hipLaunchKernelGGL(zncc_kernel, grid, dim3(kBlock), 0, ctx->stream, p, st, cv->data);
    }
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}
