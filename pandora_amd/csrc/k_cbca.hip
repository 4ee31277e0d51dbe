// k_cbca.hip - Cross-Based Cost Aggregation (Zhang 2009) on the device-resident volume.  gfx950.
//
// Reference: aggregation/cbca.py:90-295 driving aggregation_cpp (aggregation/cpp/src/aggregation.cpp):
// 3x3 nan-median (filter/median.py:134-179) -> cross_support (:224-321) -> per disparity
// cbca_step_1..4 (:28-221) -> normalisation (cbca.py:166-171).
//
// The reference differences SEQUENTIAL float32 running sums (rows, then columns).  To stay
// bit-identical for float costs each (row, d) / (column, d) scan is one sequential lane here too:
// pass H: thread = (row, d), marches along columns; pass V: thread = (column, d), marches down
// rows.  H*D and W*D independent scans, disparity innermost -> coalesced.  The prefix values a
// segment sum needs (at most cbca_distance-1 ahead / behind) live in a per-thread LDS ring.
// HBM: pass H reads cv, writes E_h; pass V reads E_h + cv (NaN test), writes cv.
#include "pmx_internal.h"

static constexpr int kBlock = 256;

__device__ __forceinline__ float c_inf() { return __int_as_float(0x7f800000); }
__device__ __forceinline__ float c_nan() { return __int_as_float(0x7fc00000); }

// ---- image preparation: mask -> NaN, 3x3 nanmedian, NaN -> +inf (cbca.py:217-282) ---------------
__global__ __launch_bounds__(kBlock) void mask_image_kernel(const float* __restrict__ img, const int16_t* __restrict__ msk,
                                                            int H, int Wd, int Wfull, int shifted, int valid,
                                                            float* __restrict__ out) {
    int c = blockIdx.x * kBlock + threadIdx.x;
    int r = blockIdx.y;
    if (c >= Wd) return;
    float v = img[(size_t)r * Wd + c];
    if (msk) {
        bool bad = msk[(size_t)r * Wfull + c] != valid;
        if (shifted) bad = bad || (msk[(size_t)r * Wfull + c + 1] != valid);  // cbca.py:246-262
        if (bad) v = c_nan();
    }
    out[(size_t)r * Wd + c] = v;
}

__global__ __launch_bounds__(kBlock) void median3_inf_kernel(const float* __restrict__ in, int H, int Wd, float* __restrict__ out) {
    int c = blockIdx.x * kBlock + threadIdx.x;
    int r = blockIdx.y;
    if (c >= Wd) return;
    float ctr = in[(size_t)r * Wd + c];
    float res = ctr;
    if (r >= 1 && r < H - 1 && c >= 1 && c < Wd - 1 && ctr == ctr) {
        float v[9];
        int n = 0;
#pragma unroll
        for (int i = -1; i <= 1; ++i)
#pragma unroll
            for (int j = -1; j <= 1; ++j) {
                float x = in[(size_t)(r + i) * Wd + c + j];
                if (x == x) v[n++] = x;
            }
        for (int a = 1; a < n; ++a) {
            float x = v[a];
            int b = a - 1;
            while (b >= 0 && v[b] > x) { v[b + 1] = v[b]; --b; }
            v[b + 1] = x;
        }
        res = (n & 1) ? v[n / 2] : (v[n / 2 - 1] + v[n / 2]) / 2.0f;
    }
    // np.nan_to_num(nan=inf): NaN -> +inf, +inf -> FLT_MAX, -inf -> -FLT_MAX
    if (res != res) res = c_inf();
    else if (res == c_inf()) res = 3.402823466e+38f;
    else if (res == -c_inf()) res = -3.402823466e+38f;
    out[(size_t)r * Wd + c] = res;
}

// aggregation.cpp:224-321 on the image cropped by `o` on every side
__global__ __launch_bounds__(kBlock) void cross_support_kernel(const float* __restrict__ img, int Wd, int o, int Hc, int Wc,
                                                               int len_arms, float intensity, int16_t* __restrict__ cross) {
    int col = blockIdx.x * kBlock + threadIdx.x;
    int row = blockIdx.y;
    if (col >= Wc) return;
    auto at = [&](int rr, int cc) { return img[(size_t)(rr + o) * Wd + cc + o]; };
    float cur = at(row, col);
    int16_t l = 0, rt = 0, up = 0, dn = 0;
    if (isfinite(cur)) {
        int lo = max(col - len_arms, -1);
        for (int x = col - 1; x > lo; --x) { if (fabsf(cur - at(row, x)) >= intensity) break; l++; }
        l = max((int)l, (int)(col >= 1 && isfinite(at(row, col - 1))));
        int hi = min(col + len_arms, Wc);
        for (int x = col + 1; x < hi; ++x) { if (fabsf(cur - at(row, x)) >= intensity) break; rt++; }
        rt = max((int)rt, (int)(col < Wc - 1 && isfinite(at(row, col + 1))));
        lo = max(row - len_arms, -1);
        for (int y = row - 1; y > lo; --y) { if (fabsf(cur - at(y, col)) >= intensity) break; up++; }
        up = max((int)up, (int)(row >= 1 && isfinite(at(row - 1, col))));
        hi = min(row + len_arms, Hc);
        for (int y = row + 1; y < hi; ++y) { if (fabsf(cur - at(y, col)) >= intensity) break; dn++; }
        dn = max((int)dn, (int)(row < Hc - 1 && isfinite(at(row + 1, col))));
    }
    short4 v = make_short4(l, rt, up, dn);
    *reinterpret_cast<short4*>(cross + ((size_t)row * Wc + col) * 4) = v;
}

// builds the arms of image `side` (0 = left, k+1 = k-th shifted right) into dev_out; tmp = 2 images
static int build_arms(pmx_ctx* ctx, int side, int offset, float intensity, int distance, float* tmp, int16_t* dev_out) {
    const int H = ctx->H, W = ctx->W;
    const float* img = side == 0 ? ctx->left : ctx->right[side - 1];
    const int16_t* msk = side == 0 ? ctx->msk_left : ctx->msk_right;
    const int shifted = side >= 2;
    const int Wd = shifted ? W - 1 : W;
    float* masked = tmp;
    float* med = tmp + (size_t)H * W;
    dim3 grid((Wd + kBlock - 1) / kBlock, H);
    hipLaunchKernelGGL(mask_image_kernel, grid, dim3(kBlock), 0, ctx->stream, img, msk, H, Wd, W, shifted, ctx->valid_value, masked);
    hipLaunchKernelGGL(median3_inf_kernel, grid, dim3(kBlock), 0, ctx->stream, masked, H, Wd, med);
    int Hc = H - 2 * offset, Wc = Wd - 2 * offset;
    if (Hc <= 0 || Wc <= 0) return PMX_OK;
    dim3 g2((Wc + kBlock - 1) / kBlock, Hc);
    hipLaunchKernelGGL(cross_support_kernel, g2, dim3(kBlock), 0, ctx->stream, med, Wd, offset, Hc, Wc, distance, intensity, dev_out);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

int pmx_launch_cross_support(pmx_ctx* ctx, int side, int offset, float intensity, int distance, int16_t* dev_out) {
    float* tmp = nullptr;
    PMX_HIP(hipMalloc((void**)&tmp, (size_t)ctx->H * ctx->W * 2 * sizeof(float)));
    int rc = build_arms(ctx, side, offset, intensity, distance, tmp, dev_out);
    hipStreamSynchronize(ctx->stream);
    hipFree(tmp);
    return rc;
}

// ---- the two scan passes -------------------------------------------------------------------------
struct cbca_args {
    float* cv;        // [H][W][D] in/out
    float* eh;        // [H][W][D] horizontal segment sums (scratch)
    const int16_t* armsL;                   // [Hc][Wc][4]
    const int16_t* armsR[PMX_MAX_SUBPIX];   // [Hc][Wr][4]
    int H, W, D, d0, subpix, o, Hc, Wc;
    int A;       // longest possible arm
    int ring;    // power of two >= 2A+2
};

// combined arm lengths of the support cross at (r, c, k); false when the right position is outside
__device__ __forceinline__ bool combined_arms(const cbca_args& a, int r, int c, int k, int& left, int& right, int& top, int& bot) {
    int kk = k / a.subpix;
    int ph = k - kk * a.subpix;
    int q = c + a.d0 + kk;
    int Wr = ph == 0 ? a.Wc : a.Wc - 1;
    if (q < 0 || q > Wr - 1) return false;  // cbca.py:156-158
    short4 al = *reinterpret_cast<const short4*>(a.armsL + ((size_t)r * a.Wc + c) * 4);
    short4 ar = *reinterpret_cast<const short4*>(a.armsR[ph] + ((size_t)r * Wr + q) * 4);
    left = min((int)al.x, (int)ar.x);
    right = min((int)al.y, (int)ar.y);
    top = min((int)al.z, (int)ar.z);
    bot = min((int)al.w, (int)ar.w);
    return true;
}

// pass H: steps 1-2 (aggregation.cpp:28-121).  thread = (row, k); E_h(c) is emitted A columns late.
__global__ __launch_bounds__(kBlock) void cbca_h_kernel(cbca_args a) {
    extern __shared__ float ring[];  // [ring][kBlock]
    const int t = blockIdx.x * kBlock + threadIdx.x;
    const int total = a.Hc * a.D;
    if (t >= total) return;
    const int r = t / a.D, k = t - r * a.D;
    const int mask = a.ring - 1;
    float* my = ring + threadIdx.x;
    const size_t row_off = ((size_t)(r + a.o) * a.W + a.o) * a.D + k;
    float acc = 0.f;
    for (int c = 0; c < a.Wc + a.A; ++c) {
        if (c < a.Wc) {
            float v = a.cv[row_off + (size_t)c * a.D];
            if (v == v) acc = acc + v;  // NaN is skipped, the running sum carries on
            my[(c & mask) * kBlock] = acc;
        }
        int ce = c - a.A;  // column whose segment sum can now be emitted
        if (ce >= 0) {
            int left, right, top, bot;
            float e = 0.f;
            if (combined_arms(a, r, ce, k, left, right, top, bot)) {
                int lo = ce - left - 1;
                float hi_v = my[((ce + right) & mask) * kBlock];
                float lo_v = lo < 0 ? 0.f : my[(lo & mask) * kBlock];
                e = hi_v - lo_v;
            }
            a.eh[row_off + (size_t)ce * a.D] = e;
        }
    }
}

// pass V: steps 3-4 + normalisation (aggregation.cpp:123-221, cbca.py:166-171).  thread = (col, k).
__global__ __launch_bounds__(kBlock) void cbca_v_kernel(cbca_args a) {
    extern __shared__ float ring[];  // [2][ring][kBlock]: column prefix sums, n_h
    const int t = blockIdx.x * kBlock + threadIdx.x;
    const int total = a.Wc * a.D;
    if (t >= total) return;
    const int c = t / a.D, k = t - c * a.D;
    const int mask = a.ring - 1;
    float* s3 = ring + threadIdx.x;
    float* nh = ring + (size_t)a.ring * kBlock + threadIdx.x;
    const size_t col_off = ((size_t)a.o * a.W + (c + a.o)) * a.D + k;
    const size_t row_stride = (size_t)a.W * a.D;
    float acc = 0.f;
    for (int r = 0; r < a.Hc + a.A; ++r) {
        if (r < a.Hc) {
            float e = a.eh[col_off + (size_t)r * row_stride];
            acc = (r == 0) ? e : acc + e;
            s3[(r & mask) * kBlock] = acc;
            int left, right, top, bot;
            float n = 0.f;
            if (combined_arms(a, r, c, k, left, right, top, bot)) n = (float)(left + right);
            nh[(r & mask) * kBlock] = n;
        }
        int re = r - a.A;
        if (re >= 0) {
            int left, right, top, bot;
            float step4 = 0.f;
            float sum4 = nh[(re & mask) * kBlock];
            if (combined_arms(a, re, c, k, left, right, top, bot)) {
                int lo = re - top - 1;
                float hi_v = s3[((re + bot) & mask) * kBlock];
                float lo_v = lo < 0 ? 0.f : s3[(lo & mask) * kBlock];
                step4 = hi_v - lo_v;
                sum4 += (float)(top + bot);
                if (top > 0) { float s = 0.f; for (int i = 1; i <= top; ++i) s += nh[((re - i) & mask) * kBlock]; sum4 += s; }
                if (bot > 0) { float s = 0.f; for (int i = 1; i <= bot; ++i) s += nh[((re + i) & mask) * kBlock]; sum4 += s; }
            }
            sum4 += 1.f;
            size_t id = col_off + (size_t)re * row_stride;
            float in = a.cv[id];
            a.cv[id] = (in * 0.f + step4) / sum4;  // NaN stays NaN (cbca.py:145-146,168-171)
        }
    }
}

int pmx_launch_cbca(pmx_ctx* ctx, pmx_cv* cv, int offset, float intensity, int distance) {
    const int H = cv->H, W = cv->W, o = offset;
    const int Hc = H - 2 * o, Wc = W - 2 * o;
    if (Hc <= 0 || Wc <= 1) return PMX_OK;
    // small scratch: 2 float images + arms of left and of every shifted right image
    size_t img_bytes = (size_t)H * W * sizeof(float);
    size_t arm_bytes = (size_t)Hc * Wc * 4 * sizeof(int16_t);
    int rc = pmx_need_small(ctx, 2 * img_bytes + arm_bytes * (1 + cv->subpix));
    if (rc) return rc;
    rc = pmx_need_scratch(ctx, cv->cells() * sizeof(float) + 64);
    if (rc) return rc;
    char* base = (char*)ctx->small;
    float* tmp = (float*)base;
    cbca_args a;
    a.cv = cv->data;
    a.eh = ctx->scratch;
    a.armsL = (int16_t*)(base + 2 * img_bytes);
    for (int k = 0; k < PMX_MAX_SUBPIX; ++k) a.armsR[k] = nullptr;
    {
        pmx_stage_scope t(ctx, PMX_STAGE_CBCA_ARMS);
        rc = build_arms(ctx, 0, o, intensity, distance, tmp, (int16_t*)a.armsL);
        if (rc) return rc;
        for (int k = 0; k < cv->subpix; ++k) {
            int16_t* dst = (int16_t*)(base + 2 * img_bytes + arm_bytes * (1 + k));
            a.armsR[k] = dst;
            rc = build_arms(ctx, k + 1, o, intensity, distance, tmp, dst);
            if (rc) return rc;
        }
    }
    a.H = H; a.W = W; a.D = cv->D; a.d0 = cv->d0; a.subpix = cv->subpix; a.o = o; a.Hc = Hc; a.Wc = Wc;
    a.A = distance - 1 > 1 ? distance - 1 : 1;
    int ring = 4;
    while (ring < 2 * a.A + 2) ring <<= 1;
    a.ring = ring;
    {
        pmx_stage_scope t(ctx, PMX_STAGE_CBCA_H);
        int total = Hc * cv->D;
        hipLaunchKernelGGL(cbca_h_kernel, dim3((total + kBlock - 1) / kBlock), dim3(kBlock), (size_t)ring * kBlock * sizeof(float),
                           ctx->stream, a);
    }
    {
        pmx_stage_scope t(ctx, PMX_STAGE_CBCA_V);
        int total = Wc * cv->D;
        hipLaunchKernelGGL(cbca_v_kernel, dim3((total + kBlock - 1) / kBlock), dim3(kBlock),
                           (size_t)2 * ring * kBlock * sizeof(float), ctx->stream, a);
    }
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}
