// k_filter.hip - SURVEY 8f N2: disparity-map filters on the device (2-D work; host maps in / out like the
// validation step).  gfx950.
#include "pmx_internal.h"

static constexpr int kBlock = 256;
#define FMSK_INVALID 0x3C3LL

__device__ __forceinline__ float f_nan() { return __int_as_float(0x7fc00000); }

// filter/median.py:94-179 MedianFilter.filter_disparity: invalid pixels count as NaN; np.nanmedian over SIZE x SIZE
// for the finite pixels whose window fits in the image (the frame keeps its values); even counts average the two
// middle values in float32.  Thread per pixel; the window is sorted in registers (SIZE known at compile time).
template <int SIZE>
__global__ __launch_bounds__(kBlock) void median_disparity_kernel(const float* __restrict__ in, const int64_t* __restrict__ validity,
                                                                  int H, int W, float* __restrict__ out) {
    constexpr int RAD = SIZE / 2, N = SIZE * SIZE;
    const int c = blockIdx.x * kBlock + threadIdx.x, r = blockIdx.y;
    if (c >= W) return;
    const size_t i = (size_t)r * W + c;
    const float centre = in[i];
    const bool centre_ok = (validity[i] & FMSK_INVALID) == 0 && isfinite(centre);
    float res = centre;
    if (centre_ok && r >= RAD && r < H - RAD && c >= RAD && c < W - RAD) {
        float v[N];
        int n = 0;
#pragma unroll
        for (int a = -RAD; a <= RAD; ++a)
#pragma unroll
            for (int b = -RAD; b <= RAD; ++b) {
                const size_t k = (size_t)(r + a) * W + c + b;
                const float x = in[k];
                const bool ok = (validity[k] & FMSK_INVALID) == 0 && x == x;
                v[a * SIZE + b + RAD * SIZE + RAD] = ok ? x : __int_as_float(0x7f800000);  // missing -> +inf: sorts last
                n += ok ? 1 : 0;
            }
        // full sorting pass (odd-even transposition: fixed indices, stays in registers)
#pragma unroll
        for (int pass = 0; pass < N; ++pass)
#pragma unroll
            for (int k = pass & 1; k + 1 < N; k += 2) {
                const float lo = fminf(v[k], v[k + 1]), hi = fmaxf(v[k], v[k + 1]);
                v[k] = lo;
                v[k + 1] = hi;
            }
        // a genuine +inf disparity sorts among the padding; it only matters if it is one of the middle values, where the
        // value is +inf either way
        float m0 = 0.f, m1 = 0.f;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            if (k == (n - 1) / 2) m0 = v[k];
            if (k == n / 2) m1 = v[k];
        }
        res = (n & 1) ? m1 : (m0 + m1) / 2.0f;
    }
    out[i] = res;
}

// any odd size: window gathered into a per-thread array (scratch memory), insertion sort
__global__ __launch_bounds__(kBlock) void median_disparity_generic_kernel(const float* __restrict__ in, const int64_t* __restrict__ validity,
                                                                          int H, int W, int size, float* __restrict__ out) {
    constexpr int kMax = 15 * 15;
    const int rad = size / 2;
    const int c = blockIdx.x * kBlock + threadIdx.x, r = blockIdx.y;
    if (c >= W) return;
    const size_t i = (size_t)r * W + c;
    const float centre = in[i];
    const bool centre_ok = (validity[i] & FMSK_INVALID) == 0 && isfinite(centre);
    float res = centre;
    if (centre_ok && r >= rad && r < H - rad && c >= rad && c < W - rad) {
        float v[kMax];
        int n = 0;
        for (int a = -rad; a <= rad; ++a)
            for (int b = -rad; b <= rad; ++b) {
                const size_t k = (size_t)(r + a) * W + c + b;
                const float x = in[k];
                if ((validity[k] & FMSK_INVALID) == 0 && x == x) {
                    int p = n++;
                    while (p > 0 && v[p - 1] > x) { v[p] = v[p - 1]; --p; }
                    v[p] = x;
                }
            }
        res = (n & 1) ? v[n / 2] : (v[n / 2 - 1] + v[n / 2]) / 2.0f;
    }
    out[i] = res;
}

int pmx_launch_median_disparity(pmx_ctx* ctx, const float* in, const int64_t* validity, int H, int W, int size, float* out) {
    dim3 grid((W + kBlock - 1) / kBlock, H);
    switch (size) {
        case 1: hipLaunchKernelGGL(median_disparity_kernel<1>, grid, dim3(kBlock), 0, ctx->stream, in, validity, H, W, out); break;
        case 3: hipLaunchKernelGGL(median_disparity_kernel<3>, grid, dim3(kBlock), 0, ctx->stream, in, validity, H, W, out); break;
        case 5: hipLaunchKernelGGL(median_disparity_kernel<5>, grid, dim3(kBlock), 0, ctx->stream, in, validity, H, W, out); break;
        default: hipLaunchKernelGGL(median_disparity_generic_kernel, grid, dim3(kBlock), 0, ctx->stream, in, validity, H, W, size, out); break;
    }
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}


// filter/bilateral.py:100-255 BilateralFilter.filter_disparity.  Thread per pixel; the spatial gaussian comes from a
// host-built float64 table, the colour gaussian is expf() of the float32 difference (as numpy evaluates it) scaled in
// float64; both sums run in float64 over the window's non-NaN elements.
__global__ __launch_bounds__(kBlock) void bilateral_disparity_kernel(const float* __restrict__ in, const int64_t* __restrict__ validity,
                                                                     int H, int W, int win, const double* __restrict__ gs,
                                                                     float sigma_color_f, double color_norm, float* __restrict__ out) {
    const int c = blockIdx.x * kBlock + threadIdx.x, r = blockIdx.y;
    if (c >= W) return;
    const int offset = win / 2;
    const size_t i0 = (size_t)r * W + c;
    const float centre = in[i0];
    const bool centre_ok = (validity[i0] & FMSK_INVALID) == 0 && isfinite(centre);
    float res = centre;
    if (centre_ok && r >= offset && r - offset + win <= H && c >= offset && c - offset + win <= W) {
        double num = 0.0, den = 0.0;
        for (int i = 0; i < win; ++i) {
            const size_t row = (size_t)(r - offset + i) * W + (c - offset);
            for (int j = 0; j < win; ++j) {
                const float w = in[row + j];
                if ((validity[row + j] & FMSK_INVALID) != 0 || w != w) continue;  // NaN window element: ignored (nansum)
                const float t = (w - centre) / sigma_color_f;
                const float g = expf(-(t * t) * 0.5f);
                const double weight = gs[i * win + j] * ((double)g / color_norm);
                if (weight != weight) continue;
                num += (double)w * weight;
                den += weight;
            }
        }
        res = (float)(num / den);
    }
    out[i0] = res;
}

int pmx_launch_bilateral_disparity(pmx_ctx* ctx, const float* in, const int64_t* validity, int H, int W, int win, const double* gs,
                                   double sigma_color, float* out) {
    dim3 grid((W + kBlock - 1) / kBlock, H);
    hipLaunchKernelGGL(bilateral_disparity_kernel, grid, dim3(kBlock), 0, ctx->stream, in, validity, H, W, win, gs, (float)sigma_color,
                       sigma_color * sqrt(2 * 3.14159265358979323846), out);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}


// filter/disparity_denoiser.py:223-313 DisparityDenoiser.filter_disparity after get_grad (the caller runs scipy's gaussian_filter
// and np.gradient: 2-D host work, the reference's own expressions).  Thread per pixel, two sweeps over the filter_size^2 window
// of the maps read through numpy's "reflect" padding (:151-166): the mean distance to the tangent plane first (:204-214), then
// the weights - euclidian gaussian from a host-built float64 table, colour gaussian in float32 as numpy evaluates it, planar
// gaussian in float64 (:290-295) - and sum(planar * w) / sum(w) (:228-232, the reference divides every weight first: 1e-16).
__device__ __forceinline__ int dn_reflect(int i, int n) {
    if (n == 1) return 0;
    const int p = 2 * (n - 1);
    int m = i % p;
    m = m < 0 ? m + p : m;
    return m < n ? m : p - m;
}

__global__ __launch_bounds__(kBlock) void denoise_disparity_kernel(const float* __restrict__ in, const int64_t* __restrict__ validity,
                                                                   const float* __restrict__ color, const float* __restrict__ grad_row,
                                                                   const float* __restrict__ grad_col, int H, int W, int ws,
                                                                   const double* __restrict__ ge, float sigma_color_f, double sigma_planar,
                                                                   float* __restrict__ out) {
    const int c = blockIdx.x * kBlock + threadIdx.x, r = blockIdx.y;
    if (c >= W) return;
    const int o = ws / 2;
    const size_t at = (size_t)r * W + c;
    const float dc = in[at];
    float res = dc;
    if ((validity[at] & FMSK_INVALID) == 0 && isfinite(dc)) {
        const double g0 = grad_row[at], g1 = grad_col[at];
        const float cc = color[at];
        double mean = 0.0;
        for (int i = -o; i <= o; ++i) {
            const size_t row = (size_t)dn_reflect(r + i, H) * W;
            for (int j = -o; j <= o; ++j) mean += (double)in[row + dn_reflect(c + j, W)] - ((double)i * g0 + (double)j * g1);
        }
        mean /= (double)(ws * ws);
        double sw = 0.0, sp = 0.0;
        for (int i = -o; i <= o; ++i) {
            const size_t row = (size_t)dn_reflect(r + i, H) * W;
            for (int j = -o; j <= o; ++j) {
                const size_t q = row + dn_reflect(c + j, W);
                const double d = (double)in[q] - ((double)i * g0 + (double)j * g1);
                const float tc = (color[q] - cc) / sigma_color_f;
                const double pc = (d - mean) / sigma_planar;
                const double w = ge[(i + o) * ws + (j + o)] * (double)expf(-(tc * tc) / 2.0f) * exp(-(pc * pc) / 2.0);
                sw += w;
                sp += (d - (double)dc) * w;
            }
        }
        res = (float)((double)dc + sp / sw);
    }
    out[at] = res;
}

int pmx_launch_denoise_disparity(pmx_ctx* ctx, const float* in, const int64_t* validity, const float* color, const float* grad_row,
                                 const float* grad_col, int H, int W, int ws, const double* ge, double sigma_color, double sigma_planar,
                                 float* out) {
    dim3 grid((W + kBlock - 1) / kBlock, H);
    hipLaunchKernelGGL(denoise_disparity_kernel, grid, dim3(kBlock), 0, ctx->stream, in, validity, color, grad_row, grad_col, H, W, ws, ge,
                       (float)sigma_color, sigma_planar, out);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}
