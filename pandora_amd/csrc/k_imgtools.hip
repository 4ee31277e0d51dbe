// k_imgtools.hip - SURVEY 8f N3: no-data interpolation of a masked image before the multiscale pyramid is built.
// 2-D work (H x W), host buffers in and out.  gfx950.
#include "pmx_internal.h"

static constexpr int kBlock = 256;

// cpp/src/img_tools.cpp:27-155 (interpolate_nodata_sgm with find_valid_neighbors and compute_median).  Thread per
// pixel; only masked pixels walk: along each of the 8 directions to the first pixel whose mask has no bit of
// `invalid_bits`, a direction that reaches the border gives nothing, NaN values are dropped; the pixel becomes the
// median of what was found (even count: float32 mean of the two middle values; nothing: NaN) and its mask `filled`.
__global__ __launch_bounds__(kBlock) void interpolate_nodata_kernel(const float* __restrict__ img, const int* __restrict__ msk, int H,
                                                                    int W, int invalid_bits, int filled, float* __restrict__ out_img,
                                                                    int* __restrict__ out_msk) {
    const int c = blockIdx.x * kBlock + threadIdx.x, r = blockIdx.y;
    if (c >= W) return;
    const size_t i = (size_t)r * W + c;
    const int m = msk[i];
    if (!(m & invalid_bits)) {
        out_img[i] = img[i];
        out_msk[i] = m;
        return;
    }
    float v[8];
    int n = 0;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        const int dc = (d >= 1 && d <= 3) ? -1 : (d >= 5 ? 1 : 0);
        const int dr = (d <= 1 || d == 7) ? 1 : ((d >= 3 && d <= 5) ? -1 : 0);
        int rr = r + dr, cc = c + dc;
        float x = __int_as_float(0x7fc00000);
        while (rr >= 0 && rr < H && cc >= 0 && cc < W) {
            const size_t j = (size_t)rr * W + cc;
            if (!(msk[j] & invalid_bits)) {
                x = img[j];
                break;
            }
            rr += dr;
            cc += dc;
        }
        v[d] = x;
        n += x == x;
    }
    // sorting network-free: 8 values, NaN ordered last by the comparison below; insertion over registers
#pragma unroll
    for (int a = 1; a < 8; ++a)
#pragma unroll
        for (int b = a; b > 0; --b) {
            const float lo = v[b - 1], hi = v[b];
            const bool swap = (lo != lo) || (hi == hi && hi < lo);  // NaN sinks to the end
            v[b - 1] = swap ? hi : lo;
            v[b] = swap ? lo : hi;
        }
    float med = __int_as_float(0x7fc00000);
    if (n > 0) {
        float a = v[0], b = v[0];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k == n / 2) a = v[k];
            if (k == (n - 1) / 2) b = v[k];
        }
        med = (n & 1) ? a : (b + a) / 2.f;
    }
    out_img[i] = med;
    out_msk[i] = filled;
}

int pmx_launch_interpolate_nodata(pmx_ctx* ctx, const float* img, const int* msk, int H, int W, int invalid_bits, int filled,
                                  float* out_img, int* out_msk) {
    dim3 grid((W + kBlock - 1) / kBlock, H);
    hipLaunchKernelGGL(interpolate_nodata_kernel, grid, dim3(kBlock), 0, ctx->stream, img, msk, H, W, invalid_bits, filled, out_img,
                       out_msk);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}
