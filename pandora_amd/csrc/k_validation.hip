// k_validation.hip - SURVEY 8f N1: cross-checking of a left/right disparity pair and the right-side disparity
// ranges, on the device.  2-D work (H x W, with a loop over the disparity range only for the pixels the check
// rejects), so the maps come and go as host buffers; the heavy inputs of the step (the right cost volume, built
// with pmx_reverse_cost_volume or a second matching pass, and its WTA) never leave HBM.  gfx950.
#include "pmx_internal.h"

static constexpr int kBlock = 256;
#define VMSK_INVALID 0x3C3LL
#define VMSK_OCCLUSION (1LL << 8)
#define VMSK_MISMATCH (1LL << 9)

__device__ __forceinline__ float v_nan() { return __int_as_float(0x7fc00000); }
__device__ __forceinline__ float v_inf() { return __int_as_float(0x7f800000); }

// validation/validation.py:226-371 (CrossCheckingAccurate.disparity_checking; same class for cross_checking_fast).
// Thread per left pixel.  q = rint(col + d_left) in double (numpy promotes float32 + int64), half to even;
// q outside the row: untouched (the reference's `outside_right` test can never be true, :354-355);
// dist = |d_right[q] + d_left| in float32, NaN -> +inf; above the threshold: MISMATCH if some d of the disparity
// interval has rint(d_right[col + d]) == -d, else OCCLUSION.
__global__ __launch_bounds__(kBlock) void cross_check_kernel(const float* __restrict__ dl, int64_t* __restrict__ validity,
                                                             const float* __restrict__ dr, int H, int W, int dmin, int dmax,
                                                             double threshold, float* __restrict__ conf) {
    const int c = blockIdx.x * kBlock + threadIdx.x, r = blockIdx.y;
    if (c >= W) return;
    const size_t i = (size_t)r * W + c;
    float out = v_nan();
    const int64_t m = validity[i];
    const float d_left = dl[i];
    if ((m & VMSK_INVALID) == 0 && d_left == d_left) {
        const double qf = rint((double)c + (double)d_left);
        if (qf >= 0.0 && qf < (double)W) {
            float d_right = dr[(size_t)r * W + (int)qf];
            if (d_right != d_right) d_right = v_inf();
            const float dist = fabsf(d_right + d_left);
            out = dist;
            if ((double)dist > threshold) {
                bool mismatch = false;
                const int lo = max(dmin, -c), hi = min(dmax, W - 1 - c);  // col + d inside the row
                for (int d = lo; d <= hi && !mismatch; ++d) mismatch = rintf(dr[(size_t)r * W + c + d]) == (float)(-d);
                validity[i] = m + (mismatch ? VMSK_MISMATCH : VMSK_OCCLUSION);
            }
        }
    }
    conf[i] = out;
}

// matching_cost.cpp:59-132 reverse_disp_range as a gather (no atomics): right pixel rc collects -d over every left
// pixel c = rc - d of its row whose integer range [(int)min, (int)max] contains d; d scans the global range.
__global__ __launch_bounds__(kBlock) void reverse_disp_range_kernel(const float* __restrict__ lmin, const float* __restrict__ lmax,
                                                                    int H, int W, int gmin, int gmax,
                                                                    float* __restrict__ rmin, float* __restrict__ rmax) {
    const int rc = blockIdx.x * kBlock + threadIdx.x, r = blockIdx.y;
    if (rc >= W) return;
    float lo = v_inf(), hi = -v_inf();
    const int d0 = max(gmin, rc - (W - 1)), d1 = min(gmax, rc);  // 0 <= rc - d < W
    for (int d = d0; d <= d1; ++d) {
        const size_t k = (size_t)r * W + (rc - d);
        const float a = lmin[k], b = lmax[k];
        if (a != a || b != b) continue;
        if (d >= (int)a && d <= (int)b) {
            lo = fminf(lo, (float)(-d));
            hi = fmaxf(hi, (float)(-d));
        }
    }
    const bool none = lo == v_inf();
    rmin[(size_t)r * W + rc] = none ? v_nan() : lo;
    rmax[(size_t)r * W + rc] = none ? v_nan() : hi;
}

#define VMSK_FILLED_OCCLUSION (1LL << 4)
#define VMSK_FILLED_MISMATCH (1LL << 5)

// validation/cpp/src/interpolated_disparity.cpp:28-75 (find_valid_neighbors, (drow, dcol) order): the disparity of the
// first pixel without an INVALID bit along each of the 8 directions, NaN when the path leaves the map.
__device__ __forceinline__ void valid_neighbors8(const float* __restrict__ disp, const int64_t* __restrict__ valid, int H, int W, int r,
                                                 int c, float (&v)[8]) {
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        const int dr = (d >= 1 && d <= 3) ? -1 : (d >= 5 ? 1 : 0);
        const int dc = (d <= 1 || d == 7) ? 1 : ((d >= 3 && d <= 5) ? -1 : 0);
        int rr = r + dr, cc = c + dc;
        float x = v_nan();
        while (rr >= 0 && rr < H && cc >= 0 && cc < W) {
            const size_t j = (size_t)rr * W + cc;
            if ((valid[j] & VMSK_INVALID) == 0) {
                x = disp[j];
                break;
            }
            rr += dr;
            cc += dc;
        }
        v[d] = x;
    }
}

// compute_median (:141-163) over N values held in registers: NaN dropped, even counts average the two middle values.
template <int N>
__device__ __forceinline__ float median_nan_free(float (&v)[N]) {
    int n = 0;
#pragma unroll
    for (int a = 0; a < N; ++a) n += v[a] == v[a];
#pragma unroll
    for (int a = 1; a < N; ++a)
#pragma unroll
        for (int b = a; b > 0; --b) {
            const float lo = v[b - 1], hi = v[b];
            const bool swap = (lo != lo) || (hi == hi && hi < lo);  // NaN sinks to the end
            v[b - 1] = swap ? hi : lo;
            v[b] = swap ? lo : hi;
        }
    if (n == 0) return v_nan();
    float a = v[0], b = v[0];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if (k == n / 2) a = v[k];
        if (k == (n - 1) / 2) b = v[k];
    }
    return (n & 1) ? a : (b + a) / 2.f;
}

// AbstractInterpolation's four passes (validation/interpolated_disparity.py:200-233, :318-330 -> interpolated_disparity.cpp).
// Every pass gathers from the input maps and writes separate outputs, thread per pixel.
// PASS 0 interpolate_occlusion_mc_cnn (:232-296): nearest valid pixel of the row, to the left first, else to the right;
//      1 interpolate_mismatch_mc_cnn (:298-393): median of the first valid pixel along 16 directions whose steps are
//        (int)(dir * i), i = 0, 1, ... (half steps repeat a pixel), the border gives NaN;
//      2 interpolate_occlusion_sgm (:101-139): among the 8 neighbours' values the one of second smallest magnitude;
//      3 interpolate_mismatch_sgm (:166-230): mismatches touching an occlusion become occlusions, the others take the median.
template <int PASS>
__global__ __launch_bounds__(kBlock) void interpolate_disparity_kernel(const float* __restrict__ disp, const int64_t* __restrict__ valid,
                                                                       int H, int W, float* __restrict__ out_disp,
                                                                       int64_t* __restrict__ out_valid) {
    const int c = blockIdx.x * kBlock + threadIdx.x, r = blockIdx.y;
    if (c >= W) return;
    const size_t i = (size_t)r * W + c;
    const int64_t m = valid[i];
    float od = disp[i];
    int64_t om = m;
    if (PASS == 0 && (m & VMSK_OCCLUSION)) {
        int found = -1;
        for (int k = c - 1; k >= 0 && found < 0; --k)
            if ((valid[(size_t)r * W + k] & VMSK_INVALID) == 0) found = k;
        for (int k = c + 1; k < W && found < 0; ++k)
            if ((valid[(size_t)r * W + k] & VMSK_INVALID) == 0) found = k;
        if (found >= 0) {
            od = disp[(size_t)r * W + found];
            om = m - VMSK_OCCLUSION + VMSK_FILLED_OCCLUSION;
        }
    } else if (PASS == 1 && (m & VMSK_MISMATCH)) {
        float v[16];
        const int maxlen = max(H, W);
#pragma unroll
        for (int d = 0; d < 16; ++d) {
            // (dcol, drow) of :318-335, doubled: the ring (0,1) (-.5,1) (-1,1) (-1,.5) (-1,0) (-1,-.5) (-1,-1) (-.5,-1) (0,-1) ...
            constexpr int tx[16] = {0, -1, -2, -2, -2, -2, -2, -1, 0, 1, 2, 2, 2, 2, 2, 1};
            constexpr int ty[16] = {2, 2, 2, 1, 0, -1, -2, -2, -2, -2, -2, -1, 0, 1, 2, 2};
            const float fx = 0.5f * (float)tx[d], fy = 0.5f * (float)ty[d];
            float x = 0.f;
            for (int k = 0; k < maxlen; ++k) {
                const int cc = c + (int)(fx * (float)k), rr = r + (int)(fy * (float)k);
                if (rr < 0 || rr >= H || cc < 0 || cc >= W) {
                    x = v_nan();
                    break;
                }
                const size_t j = (size_t)rr * W + cc;
                if ((valid[j] & VMSK_INVALID) == 0) {
                    x = disp[j];
                    break;
                }
            }
            v[d] = x;
        }
        od = median_nan_free<16>(v);
        om = m + VMSK_FILLED_MISMATCH - VMSK_MISMATCH;
    } else if (PASS == 2 && (m & VMSK_OCCLUSION)) {
        float v[8];
        valid_neighbors8(disp, valid, H, W, r, c, v);
        float mn = v_inf(), mna = v_inf(), sm = v_inf(), sma = v_inf();  // get_second_min_val_abs :77-99
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            const float a = fabsf(v[d]);
            if (a < mna) {
                sma = mna;
                sm = mn;
                mna = a;
                mn = v[d];
            } else if (a < sma) {
                sma = a;
                sm = v[d];
            }
        }
        od = sm;
        om = m + VMSK_FILLED_OCCLUSION - VMSK_OCCLUSION;
    } else if (PASS == 3 && (m & VMSK_MISMATCH)) {
        bool near_occ = false;
        for (int rr = max(r - 1, 0); rr <= min(r + 1, H - 1); ++rr)
            for (int cc = max(c - 1, 0); cc <= min(c + 1, W - 1); ++cc) near_occ |= (valid[(size_t)rr * W + cc] & VMSK_OCCLUSION) != 0;
        if (near_occ) {
            om = m - VMSK_MISMATCH + VMSK_OCCLUSION;
        } else {
            float v[8];
            valid_neighbors8(disp, valid, H, W, r, c, v);
            od = median_nan_free<8>(v);
            om = m + VMSK_FILLED_MISMATCH - VMSK_MISMATCH;
        }
    }
    out_disp[i] = od;
    out_valid[i] = om;
}

int pmx_launch_interpolate_disparity(pmx_ctx* ctx, int pass, const float* disp, const int64_t* valid, int H, int W, float* out_disp,
                                     int64_t* out_valid) {
    dim3 grid((W + kBlock - 1) / kBlock, H);
#define PMX_INTERP(P) \
    hipLaunchKernelGGL(interpolate_disparity_kernel<P>, grid, dim3(kBlock), 0, ctx->stream, disp, valid, H, W, out_disp, out_valid)
    switch (pass) {
        case 0: PMX_INTERP(0); break;
        case 1: PMX_INTERP(1); break;
        case 2: PMX_INTERP(2); break;
        case 3: PMX_INTERP(3); break;
        default: return PMX_ERR_ARG;
    }
#undef PMX_INTERP
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

int pmx_launch_cross_checking(pmx_ctx* ctx, const float* dl, int64_t* validity, const float* dr, int H, int W, int dmin, int dmax,
                              double threshold, float* conf) {
    dim3 grid((W + kBlock - 1) / kBlock, H);
    hipLaunchKernelGGL(cross_check_kernel, grid, dim3(kBlock), 0, ctx->stream, dl, validity, dr, H, W, dmin, dmax, threshold, conf);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

int pmx_launch_reverse_disp_range(pmx_ctx* ctx, const float* lmin, const float* lmax, int H, int W, int gmin, int gmax, float* rmin,
                                  float* rmax) {
    dim3 grid((W + kBlock - 1) / kBlock, H);
    hipLaunchKernelGGL(reverse_disp_range_kernel, grid, dim3(kBlock), 0, ctx->stream, lmin, lmax, H, W, gmin, gmax, rmin, rmax);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}


// ---- multiscale/fixed_zoom_pyramid.py:106-172 disparity_range (before the zoom) ----------------------
// Thread per pixel: min / max of the valid, non-NaN disparities of the window -/+ marge; the frame and the pixels that
// are invalid or NaN themselves keep the global range.
__global__ __launch_bounds__(kBlock) void disparity_range_kernel(const float* __restrict__ disp, const int64_t* __restrict__ validity,
                                                                 int H, int W, int win, float marge, float gmin, float gmax,
                                                                 float* __restrict__ out_min, float* __restrict__ out_max) {
    const int c = blockIdx.x * kBlock + threadIdx.x, r = blockIdx.y;
    if (c >= W) return;
    const int off = (win - 1) / 2;
    const size_t i0 = (size_t)r * W + c;
    float lo = gmin, hi = gmax;
    const float centre = disp[i0];
    if ((validity[i0] & VMSK_INVALID) == 0 && centre == centre && r >= off && r - off + win <= H && c >= off && c - off + win <= W) {
        float mn = v_inf(), mx = -v_inf();
        for (int i = 0; i < win; ++i) {
            const size_t row = (size_t)(r - off + i) * W + (c - off);
            for (int j = 0; j < win; ++j) {
                const float v = disp[row + j];
                if ((validity[row + j] & VMSK_INVALID) != 0 || v != v) continue;
                mn = fminf(mn, v);
                mx = fmaxf(mx, v);
            }
        }
        lo = mn - marge;
        hi = mx + marge;
    }
    out_min[i0] = lo;
    out_max[i0] = hi;
}

int pmx_launch_disparity_range(pmx_ctx* ctx, const float* disp, const int64_t* validity, int H, int W, int win, int marge, int gmin,
                               int gmax, float* out_min, float* out_max) {
    dim3 grid((W + kBlock - 1) / kBlock, H);
    hipLaunchKernelGGL(disparity_range_kernel, grid, dim3(kBlock), 0, ctx->stream, disp, validity, H, W, win, (float)marge, (float)gmin,
                       (float)gmax, out_min, out_max);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}
