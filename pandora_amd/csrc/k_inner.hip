// k_inner.hip - the reference's INNER native functions as C entry points (SURVEY 8b "inner" boundary): the granularity of
// aggregation_cpp.cross_support / aggregation_cpp.cbca, which the plugin-level entry points (pmx_cbca: median + arms + every
// disparity in two fused scans) deliberately do not have.  They exist so that the pybind11 face (csrc/inner_face.cpp ->
// pandora_amd.inner_cpp) can offer the reference's own call signatures over libpandora_amd.so - a maintainer swaps one import in
// aggregation/cbca.py and keeps the python loop over disparities - and so that each call can be diffed against the compiled
// reference (oracle/_ref) argument for argument.  Host arrays in, host arrays out: exactly what the pybind functions they replace
// take and return.  gfx950.
#include "pmx_internal.h"

namespace {

// aggregation.cpp:28-121 (steps 1 and 2) on one disparity slice: thread = image row, float32 running sums in the reference's order
__global__ __launch_bounds__(64) void cbca_slice_h_kernel(const float* __restrict__ in, const int16_t* __restrict__ cl,
                                                          const int16_t* __restrict__ cr, const int* __restrict__ colmap, int H, int W,
                                                          int Wr, float* __restrict__ s1, float* __restrict__ e2, float* __restrict__ n2) {
    const int r = blockIdx.x * 64 + threadIdx.x;
    if (r >= H) return;
    float acc = 0.f;
    for (int c = 0; c < W; ++c) {
        const float v = in[(size_t)r * W + c];
        if (v == v) acc = acc + v;  // NaN costs are skipped, the sum is carried
        s1[(size_t)r * (W + 1) + c] = acc;
    }
    s1[(size_t)r * (W + 1) + W] = 0.f;
    for (int c = 0; c < W; ++c) {
        const int q = colmap[c];
        float e = 0.f, n = 0.f;
        if (q >= 0) {
            const int16_t* al = cl + ((size_t)r * W + c) * 4;
            const int16_t* ar = cr + ((size_t)r * Wr + q) * 4;
            const int left = al[0] < ar[0] ? al[0] : ar[0], right = al[1] < ar[1] ? al[1] : ar[1];
            const int lo = c - left - 1;
            e = s1[(size_t)r * (W + 1) + c + right] - (lo < 0 ? 0.f : s1[(size_t)r * (W + 1) + lo]);  // index -1 reads as 0
            n = (float)(left + right);
        }
        e2[(size_t)r * W + c] = e;
        n2[(size_t)r * W + c] = n;
    }
}

// aggregation.cpp:123-221 (steps 3 and 4): thread = image column
__global__ __launch_bounds__(64) void cbca_slice_v_kernel(const int16_t* __restrict__ cl, const int16_t* __restrict__ cr,
                                                          const int* __restrict__ colmap, int H, int W, int Wr, const float* __restrict__ e2,
                                                          const float* __restrict__ n2, float* __restrict__ s3, float* __restrict__ out_e,
                                                          float* __restrict__ out_n) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= W) return;
    float acc = 0.f;
    for (int r = 0; r < H; ++r) {
        acc = r == 0 ? e2[c] : acc + e2[(size_t)r * W + c];
        s3[(size_t)r * W + c] = acc;
    }
    s3[(size_t)H * W + c] = 0.f;
    const int q = colmap[c];
    for (int r = 0; r < H; ++r) {
        float step4 = 0.f, sum4 = n2[(size_t)r * W + c];
        if (q >= 0) {
            const int16_t* al = cl + ((size_t)r * W + c) * 4;
            const int16_t* ar = cr + ((size_t)r * Wr + q) * 4;
            const int top = al[2] < ar[2] ? al[2] : ar[2], bot = al[3] < ar[3] ? al[3] : ar[3];
            int sr = r - top - 1;
            if (sr < 0) sr += H + 1;  // wraps to the zero row
            step4 = s3[(size_t)(r + bot) * W + c] - s3[(size_t)sr * W + c];
            sum4 += (float)(top + bot);
            if (top > 0) {
                float s = 0.f;
                for (int i = 1; i <= top; ++i) s += n2[(size_t)(r - i) * W + c];
                sum4 += s;
            }
            if (bot > 0) {
                float s = 0.f;
                for (int i = 1; i <= bot; ++i) s += n2[(size_t)(r + i) * W + c];
                sum4 += s;
            }
        }
        out_e[(size_t)r * W + c] = step4;
        out_n[(size_t)r * W + c] = sum4;
    }
}

__global__ __launch_bounds__(64) void arms_kernel(const float* __restrict__ img, int H, int W, int len_arms, float intensity,
                                                  int16_t* __restrict__ cross) {
    // aggregation.cpp:224-321 on a given image (the caller has filtered it and marked invalid pixels +inf, cbca.py:217-295)
    const int col = blockIdx.x * 64 + threadIdx.x, row = blockIdx.y;
    if (col >= W) return;
    auto at = [&](int rr, int cc) { return img[(size_t)rr * W + cc]; };
    const float cur = at(row, col);
    int l = 0, rt = 0, up = 0, dn = 0;
    if (isfinite(cur)) {
        int lo = max(col - len_arms, -1);
        for (int x = col - 1; x > lo; --x) { if (fabsf(cur - at(row, x)) >= intensity) break; l++; }
        l = max(l, (int)(col >= 1 && isfinite(at(row, col - 1))));
        int hi = min(col + len_arms, W);
        for (int x = col + 1; x < hi; ++x) { if (fabsf(cur - at(row, x)) >= intensity) break; rt++; }
        rt = max(rt, (int)(col < W - 1 && isfinite(at(row, col + 1))));
        lo = max(row - len_arms, -1);
        for (int y = row - 1; y > lo; --y) { if (fabsf(cur - at(y, col)) >= intensity) break; up++; }
        up = max(up, (int)(row >= 1 && isfinite(at(row - 1, col))));
        hi = min(row + len_arms, H);
        for (int y = row + 1; y < hi; ++y) { if (fabsf(cur - at(y, col)) >= intensity) break; dn++; }
        dn = max(dn, (int)(row < H - 1 && isfinite(at(row + 1, col))));
    }
    *reinterpret_cast<short4*>(cross + ((size_t)row * W + col) * 4) = make_short4((short)l, (short)rt, (short)up, (short)dn);
}

struct dev_block {  // a pool block that goes back on every way out
    pmx_ctx* ctx;
    void* p = nullptr;
    ~dev_block() { pmx_pool_free(ctx, p); }
};

}  // namespace

extern "C" int pmx_cross_support_image(pmx_ctx* ctx, const float* image, int H, int W, int len_arms, float intensity, int16_t* host_out) {
    PMX_CHECK(ctx && image && host_out && H > 0 && W > 0 && len_arms >= 0, PMX_ERR_ARG, "pmx_cross_support_image: bad argument");
    PMX_HIP(hipSetDevice(ctx->device));
    const size_t nimg = (size_t)H * W * sizeof(float), narms = (size_t)H * W * 4 * sizeof(int16_t);
    dev_block b{ctx};
    PMX_HIP(pmx_pool_alloc(ctx, &b.p, nimg + narms));
    float* dimg = (float*)b.p;
    int16_t* darms = (int16_t*)((char*)b.p + nimg);
    PMX_HIP(hipMemcpyAsync(dimg, image, nimg, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(arms_kernel, dim3((W + 63) / 64, H), dim3(64), 0, ctx->stream, dimg, H, W, len_arms, intensity, darms);
    PMX_HIP(hipGetLastError());
    PMX_HIP(hipMemcpyAsync(host_out, darms, narms, hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    return PMX_OK;
}

extern "C" int pmx_cbca_slice(pmx_ctx* ctx, const float* input, const int16_t* cross_left, const int16_t* cross_right, int H, int W,
                              int Wr, const int64_t* range_col, const int64_t* range_col_right, int nvalid, float* out_e, float* out_n) {
    PMX_CHECK(ctx && input && cross_left && cross_right && out_e && out_n && H > 0 && W > 0 && Wr > 0 && nvalid >= 0 &&
                  (nvalid == 0 || (range_col && range_col_right)),
              PMX_ERR_ARG, "pmx_cbca_slice: bad argument");
    std::vector<int> colmap((size_t)W, -1);
    for (int i = 0; i < nvalid; ++i) {
        const int64_t c = range_col[i], q = range_col_right[i];
        PMX_CHECK(c >= 0 && c < W && q >= 0 && q < Wr, PMX_ERR_ARG, "pmx_cbca_slice: column pair (%lld, %lld) outside the images", (long long)c,
                  (long long)q);
        colmap[(size_t)c] = (int)q;
    }
    // The kernels index their prefix sums with the arms (aggregation.cpp:99-117, :181-213 do the same, unchecked): arms that do
    // not fit the image would read foreign device memory, so they are refused here, on the host, before anything is launched.
    for (int r = 0; r < H; ++r) {
        for (int c = 0; c < W; ++c) {
            const int q = colmap[(size_t)c];
            if (q < 0) continue;
            const int16_t* al = cross_left + ((size_t)r * W + c) * 4;
            const int16_t* ar = cross_right + ((size_t)r * Wr + q) * 4;
            const int left = al[0] < ar[0] ? al[0] : ar[0], right = al[1] < ar[1] ? al[1] : ar[1];
            const int top = al[2] < ar[2] ? al[2] : ar[2], bot = al[3] < ar[3] ? al[3] : ar[3];
            PMX_CHECK(left >= 0 && left <= c && right >= 0 && c + right <= W && top >= 0 && top <= r && bot >= 0 && r + bot <= H - 1, PMX_ERR_ARG,
                      "pmx_cbca_slice: the arms (%d, %d, %d, %d) of pixel (%d, %d) do not fit the %d x %d image", left, right, top, bot, r, c, H, W);
        }
    }
    PMX_HIP(hipSetDevice(ctx->device));
    const size_t nf = (size_t)H * W * sizeof(float), nl = (size_t)H * W * 4 * sizeof(int16_t), nr = (size_t)H * Wr * 4 * sizeof(int16_t);
    const size_t ns1 = (size_t)H * (W + 1) * sizeof(float), ns3 = (size_t)(H + 1) * W * sizeof(float), ncm = (size_t)W * sizeof(int);
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    dev_block b{ctx};
    PMX_HIP(pmx_pool_alloc(ctx, &b.p, up(nf) * 5 + up(nl) + up(nr) + up(ns1) + up(ns3) + up(ncm)));
    char* p = (char*)b.p;
    auto take = [&](size_t n) { char* q = p; p += up(n); return q; };
    float* din = (float*)take(nf); float* e2 = (float*)take(nf); float* n2 = (float*)take(nf);
    float* doe = (float*)take(nf); float* don = (float*)take(nf);
    int16_t* dcl = (int16_t*)take(nl); int16_t* dcr = (int16_t*)take(nr);
    float* s1 = (float*)take(ns1); float* s3 = (float*)take(ns3);
    int* dcm = (int*)take(ncm);
    PMX_HIP(hipMemcpyAsync(din, input, nf, hipMemcpyHostToDevice, ctx->stream));
    PMX_HIP(hipMemcpyAsync(dcl, cross_left, nl, hipMemcpyHostToDevice, ctx->stream));
    PMX_HIP(hipMemcpyAsync(dcr, cross_right, nr, hipMemcpyHostToDevice, ctx->stream));
    PMX_HIP(hipMemcpyAsync(dcm, colmap.data(), ncm, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(cbca_slice_h_kernel, dim3((H + 63) / 64), dim3(64), 0, ctx->stream, din, dcl, dcr, dcm, H, W, Wr, s1, e2, n2);
    hipLaunchKernelGGL(cbca_slice_v_kernel, dim3((W + 63) / 64), dim3(64), 0, ctx->stream, dcl, dcr, dcm, H, W, Wr, e2, n2, s3, doe, don);
    PMX_HIP(hipGetLastError());
    PMX_HIP(hipMemcpyAsync(out_e, doe, nf, hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipMemcpyAsync(out_n, don, nf, hipMemcpyDeviceToHost, ctx->stream));
    PMX_HIP(hipStreamSynchronize(ctx->stream));  // (colmap and the host arrays must outlive the copies)
    return PMX_OK;
}
