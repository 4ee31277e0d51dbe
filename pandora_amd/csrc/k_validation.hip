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
