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
//
// Kernels (pmx_launch_cbca picks; DESIGN.md section 3 has the measurements behind each choice):
//   census source, arms of at most 4 (cbca_distance <= 5), 11 <= D <= 256:
//           cbca_census_march_kernel   costs, horizontal sums, vertical scan and the division in ONE marching launch on exact integer
//                                sums (every census sum is an integer below 2^24: float32 order does not matter); no E_h volume
//   everything else (float costs, longer arms), two passes over the volume:
//   pass H  cbca_h_rows_kernel   a workgroup owns whole image rows and stores E_h through an LDS stage (16 bytes per lane);
//                                SRC 1-3: the census Hamming costs are computed in the kernel, the cost volume never exists
//           cbca_h_fast_kernel   phase-split scan, 4 bytes per lane (short scans, PMX_CBCA_FAST=2)
//           cbca_h_kernel        generic (images a few arms wide; 64-thread workgroups for arms of 21 columns and more)
//           cbca_h4_kernel       four disparities per thread, in place (no second volume available)
//   pass V  cbca_v_fast_kernel   phase-split scan with per-lane pointers (volumes up to ~6000 wavefronts)
//           cbca_v_buf_kernel    the same through buffer instructions, 62 registers, 256 / 512 threads (large volumes)
//           cbca_v_kernel, cbca_v4_kernel   as for pass H
// HBM: pass H reads the costs (or the census codes) and writes E_h; pass V reads E_h and writes the volume; which input costs
// were NaN travels in the sign bit of E_h when the costs are known to be >= 0 (else pass V reads the input volume again).
#include <cstdlib>

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

// nanmedian of the 3x3 neighbourhood, all in registers: NaNs become +inf and sink to the top of a 25-exchange sorting network
// (a valid +inf sorts among them with the same value, so whichever of them the median index meets reads the same), the median
// of the n numbers is sorted[n/2] or the mean of the two middle ones.  (An insertion sort over a local array indexes it
// dynamically, lives in scratch memory and took 3.1 ms per 10000^2 image against 0.3 for this.)
__device__ __forceinline__ void cb_cswap(float& x, float& y) {
    const float lo = fminf(x, y), hi = fmaxf(x, y);
    x = lo;
    y = hi;
}
__device__ __forceinline__ float cb_pick9(const float (&v)[9], int i) {
    float r = v[0];
#pragma unroll
    for (int j = 1; j < 9; ++j) r = (i == j) ? v[j] : r;
    return r;
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
                const float x = in[(size_t)(r + i) * Wd + c + j];
                const bool num = x == x;
                n += num ? 1 : 0;
                v[(i + 1) * 3 + j + 1] = num ? x : c_inf();
            }
        if (n == 9) {
            // nine numbers (no NaN among the operands): sort the three columns (min3 / med3 / max3); the median of all nine is the
            // median of (largest of the minima, median of the medians, smallest of the maxima) - 13 instructions
            const float lo = fmaxf(fmaxf(fminf(fminf(v[0], v[3]), v[6]), fminf(fminf(v[1], v[4]), v[7])), fminf(fminf(v[2], v[5]), v[8]));
            const float hi = fminf(fminf(fmaxf(fmaxf(v[0], v[3]), v[6]), fmaxf(fmaxf(v[1], v[4]), v[7])), fmaxf(fmaxf(v[2], v[5]), v[8]));
            const float mid = __builtin_amdgcn_fmed3f(__builtin_amdgcn_fmed3f(v[0], v[3], v[6]), __builtin_amdgcn_fmed3f(v[1], v[4], v[7]),
                                                      __builtin_amdgcn_fmed3f(v[2], v[5], v[8]));
            res = __builtin_amdgcn_fmed3f(lo, mid, hi);
        } else {
            // 9-input sorting network (25 compare-exchanges)
            cb_cswap(v[0], v[1]); cb_cswap(v[3], v[4]); cb_cswap(v[6], v[7]);
            cb_cswap(v[1], v[2]); cb_cswap(v[4], v[5]); cb_cswap(v[7], v[8]);
            cb_cswap(v[0], v[1]); cb_cswap(v[3], v[4]); cb_cswap(v[6], v[7]);
            cb_cswap(v[0], v[3]); cb_cswap(v[3], v[6]); cb_cswap(v[0], v[3]);
            cb_cswap(v[1], v[4]); cb_cswap(v[4], v[7]); cb_cswap(v[1], v[4]);
            cb_cswap(v[2], v[5]); cb_cswap(v[5], v[8]); cb_cswap(v[2], v[5]);
            cb_cswap(v[1], v[3]); cb_cswap(v[5], v[7]);
            cb_cswap(v[2], v[6]); cb_cswap(v[4], v[6]); cb_cswap(v[2], v[4]);
            cb_cswap(v[2], v[3]); cb_cswap(v[5], v[6]);
            const float hi = cb_pick9(v, n / 2);
            res = (n & 1) ? hi : (cb_pick9(v, n / 2 - 1) + hi) / 2.0f;
        }
    }
    // np.nan_to_num(nan=inf): NaN -> +inf, +inf -> FLT_MAX, -inf -> -FLT_MAX
    if (res != res) res = c_inf();
    else if (res == c_inf()) res = 3.402823466e+38f;
    else if (res == -c_inf()) res = -3.402823466e+38f;
    out[(size_t)r * Wd + c] = res;
}

// aggregation.cpp:224-321 on the image cropped by `o` on every side
// (pitch, xoff): pixels per output row and where column 0 sits in it (dense: Wc, 0; the four-disparity kernels read the right
// image's arms 4 pixels at a time from rows padded by 4 zeroed pixels on either side)
__global__ __launch_bounds__(kBlock) void cross_support_kernel(const float* __restrict__ img, int Wd, int o, int Hc, int Wc,
                                                               int len_arms, float intensity, int16_t* __restrict__ cross, int pitch,
                                                               int xoff) {
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
    *reinterpret_cast<short4*>(cross + ((size_t)row * pitch + col + xoff) * 4) = v;
}

// The same arms without a data-dependent loop, for arms shorter than LM (the default cbca_distance is 5): the 4 x (len - 1)
// neighbours are loaded at once (clamped addresses, one memory latency instead of one per step), every comparison leaves a bit,
// and an arm is the number of consecutive set bits from the pixel outwards.  PACKED: the arms leave as one word per pixel
// (left | right << 8 | top << 16 | bottom << 24) in the padded rows the whole-row and marching kernels read, else as short4.
template <int LM, bool PACKED>
__global__ __launch_bounds__(kBlock) void cross_support_flat_kernel(const float* __restrict__ img, int Wd, int o, int Hc, int Wc,
                                                                    int len_arms, float intensity, void* __restrict__ out, int pitch,
                                                                    int xoff) {
    // (PACKED: the grid spans the padded row, the pads are zero arms)
    const int col = blockIdx.x * kBlock + threadIdx.x - (PACKED ? xoff : 0);
    const int row = blockIdx.y;
    if (PACKED) {
        if (col + xoff >= pitch) return;
        if (col < 0 || col >= Wc) {
            reinterpret_cast<uint32_t*>(out)[(size_t)row * pitch + col + xoff] = 0u;
            return;
        }
    } else if (col >= Wc) return;
    const float* base = img + (size_t)o * Wd + o;
    const float* prow = base + (size_t)row * Wd;
    const float cur = prow[col];
    unsigned ml = 0, mr = 0, mu = 0, md = 0;
    bool nl = false, nr = false, nu = false, nd = false;  // the neighbour exists and is finite: an arm of at least 1
#pragma unroll
    for (int k = 1; k < LM; ++k) {
        const float vl = prow[max(col - k, 0)];
        const float vr = prow[min(col + k, Wc - 1)];
        const float vu = base[(size_t)max(row - k, 0) * Wd + col];
        const float vd = base[(size_t)min(row + k, Hc - 1) * Wd + col];
        const bool use = k < len_arms;
        const bool il = col - k >= 0, ir = col + k < Wc, iu = row - k >= 0, id = row + k < Hc;
        ml |= (use && il && !(fabsf(cur - vl) >= intensity)) ? 1u << (k - 1) : 0u;
        mr |= (use && ir && !(fabsf(cur - vr) >= intensity)) ? 1u << (k - 1) : 0u;
        mu |= (use && iu && !(fabsf(cur - vu) >= intensity)) ? 1u << (k - 1) : 0u;
        md |= (use && id && !(fabsf(cur - vd) >= intensity)) ? 1u << (k - 1) : 0u;
        if (k == 1) {
            nl = il && isfinite(vl);
            nr = ir && isfinite(vr);
            nu = iu && isfinite(vu);
            nd = id && isfinite(vd);
        }
    }
    int l = 0, rt = 0, up = 0, dn = 0;
    if (isfinite(cur)) {
        l = max(__builtin_ctz(~ml), (int)nl);
        rt = max(__builtin_ctz(~mr), (int)nr);
        up = max(__builtin_ctz(~mu), (int)nu);
        dn = max(__builtin_ctz(~md), (int)nd);
    }
    if (PACKED)
        reinterpret_cast<uint32_t*>(out)[(size_t)row * pitch + col + xoff] = (uint32_t)l | ((uint32_t)rt << 8) | ((uint32_t)up << 16) | ((uint32_t)dn << 24);
    else
        *reinterpret_cast<short4*>(reinterpret_cast<int16_t*>(out) + ((size_t)row * pitch + col + xoff) * 4) = make_short4((short)l, (short)rt, (short)up, (short)dn);
}

__global__ void pack_arms_kernel(const int16_t* __restrict__ arms, int Hc, int Wsrc, uint32_t* __restrict__ rows, int pitch, int xoff);

static bool arms_flat(const pmx_ctx* ctx, int distance) {
    const char* ef = pmx_opt(ctx, "CBCA_ARMS_FLAT");  // =0: the loop form (A/B and test hook)
    return !(ef && ef[0] == '0') && distance >= 1 && distance <= 18;
}

// builds the arms of image `side` (0 = left, k+1 = k-th shifted right) into dev_out; tmp = 2 images
// (packed != nullptr: also one word per pixel into rows of `ppitch` words, pixel 0 at word `pxoff`; !want16: only those)
static int build_arms(pmx_ctx* ctx, int side, int offset, float intensity, int distance, float* tmp, int16_t* dev_out, int pad = 0,
                      uint32_t* packed = nullptr, int ppitch = 0, int pxoff = 0, bool want16 = true) {
    const int H = ctx->H, W = ctx->W;
    const float* img = side == 0 ? ctx->left : ctx->right[side - 1];
    const int16_t* msk = side == 0 ? ctx->msk_left : ctx->msk_right;
    const int shifted = side >= 2;
    const int Wd = shifted ? W - 1 : W;
    float* masked = tmp;
    float* med = tmp + (size_t)H * W;
    dim3 grid((Wd + kBlock - 1) / kBlock, H);
    if (msk) hipLaunchKernelGGL(mask_image_kernel, grid, dim3(kBlock), 0, ctx->stream, img, msk, H, Wd, W, shifted, ctx->valid_value, masked);
    hipLaunchKernelGGL(median3_inf_kernel, grid, dim3(kBlock), 0, ctx->stream, msk ? masked : img, H, Wd, med);  // (no mask: nothing to copy)
    int Hc = H - 2 * offset, Wc = Wd - 2 * offset;
    if (Hc <= 0 || Wc <= 0) return PMX_OK;
    dim3 g2((Wc + kBlock - 1) / kBlock, Hc);
    const bool flat = arms_flat(ctx, distance);
#define PMX_FLAT(LMV)                                                                                                                  \
    do {                                                                                                                               \
        if (packed && !want16)                                                                                                         \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(cross_support_flat_kernel<LMV, true>), dim3((ppitch + kBlock - 1) / kBlock, Hc), dim3(kBlock), 0, ctx->stream, med, Wd, offset, Hc, \
                               Wc, distance, intensity, (void*)packed, ppitch, pxoff);                                                 \
        else                                                                                                                           \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(cross_support_flat_kernel<LMV, false>), g2, dim3(kBlock), 0, ctx->stream, med, Wd, offset, Hc, \
                               Wc, distance, intensity, (void*)dev_out, Wc + 2 * pad, pad);                                            \
    } while (0)
    if (flat && distance <= 6) PMX_FLAT(6);
    else if (flat && distance <= 10) PMX_FLAT(10);
    else if (flat) PMX_FLAT(18);
    else
        hipLaunchKernelGGL(cross_support_kernel, g2, dim3(kBlock), 0, ctx->stream, med, Wd, offset, Hc, Wc, distance, intensity, dev_out,
                           Wc + 2 * pad, pad);
    if (packed && (want16 || !flat))
        hipLaunchKernelGGL(pack_arms_kernel, dim3((Wc + 255) / 256, Hc), dim3(256), 0, ctx->stream, dev_out, Hc, Wc, packed, ppitch, pxoff);
#undef PMX_FLAT
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
    // four-disparity kernels (subpix 1): right arms in rows of Wc + 8 pixels (pixel q at index q + 4, pads zero), NaN flags of
    // the input costs as 4 bits per (row, column, group of 4 disparities): uint32 [Hc][ceil(Wc/8)][G], column c in bits 4*(c%8)..
    const int16_t* armsRpad;
    uint32_t* nanbits;
    int G;       // groups of 4 disparities per pixel
    // whole-row pass H (cbca_h_rows_kernel): arms as 4 bytes per pixel (left | right << 8 | top << 16 | bottom << 24), row-major
    // [Hc][pitchL] / per phase [Hc][pitchR] with pixel q of the right image at index padR + q and zero pads, so that the arms of
    // FOUR consecutive steps are one 16-byte load and no index needs clamping
    const uint32_t* armsL8;
    const uint32_t* armsR8;      // phase ph at armsR8 + ph * phase_words
    int pitchL, pitchR, padR;
    unsigned bytesL8, bytesR8, phase_words;
    int R;                       // image rows per workgroup
    // census costs computed in place of a volume read (cbca_h_rows_kernel<., 1>): the handle's codes (one word per pixel), the
    // per-pixel valid interval of cv_masked (or nullptr: census geometry), the census border
    const uint32_t* codes;       // the whole allocation (buffer descriptor)
    unsigned codes_bytes, offCL, offCR;  // word offsets of the left / right code images in it
    const uint32_t* range;
    int cb;                      // census window / 2
};

// combined arm lengths of the support cross at (r, c, k) packed as left | right<<8 | top<<16 | bot<<24;
// 0xffffffff when the right position is outside the image (cbca.py:156-158)
struct arm_pair {
    short4 l, r;
    bool inside;
};

__device__ __forceinline__ arm_pair load_arms(const cbca_args& a, int r, int c, int ph, int q) {
    arm_pair p;
    const int Wr = ph == 0 ? a.Wc : a.Wc - 1;
    p.inside = (q >= 0) && (q <= Wr - 1);
    const int qq = p.inside ? q : 0;
    p.l = *reinterpret_cast<const short4*>(a.armsL + ((size_t)r * a.Wc + c) * 4);
    p.r = *reinterpret_cast<const short4*>(a.armsR[ph] + ((size_t)r * Wr + qq) * 4);
    return p;
}

__device__ __forceinline__ uint32_t combine(const arm_pair& p) {
    if (!p.inside) return 0xffffffffu;
    const uint32_t left = min((int)p.l.x, (int)p.r.x), right = min((int)p.l.y, (int)p.r.y);
    const uint32_t top = min((int)p.l.z, (int)p.r.z), bot = min((int)p.l.w, (int)p.r.w);
    return left | (right << 8) | (top << 16) | (bot << 24);
}

static constexpr int kCbcaPF = 8;  // rows / columns of read-ahead (register ring)

// pass H: steps 1-2 (aggregation.cpp:28-121).  thread = (row, k) marches along the columns; the
// segment sum E_h(c) needs the running sums up to A columns ahead, so it is emitted A columns late
// from an LDS ring.  The cost stream and the arms are read kCbcaPF columns ahead into registers:
// the scan itself is a serial fp32 dependency (the reference's rounding), the loads are not.
template <int BS>  // threads per workgroup: 256, or 64 when the ring of a long arm would not fit the LDS otherwise
__global__ __launch_bounds__(BS) void cbca_h_kernel(cbca_args a) {
    extern __shared__ float ring[];  // [ring][BS]
    const int t = blockIdx.x * BS + threadIdx.x;
    const int total = a.Hc * a.D;
    const bool live = t < total;
    const int tt = live ? t : total - 1;
    const int r = tt / a.D, k = tt - r * a.D;
    const int kk = k / a.subpix, ph = k - kk * a.subpix, dq = a.d0 + kk;
    const int mask = a.ring - 1;
    float* my = ring + threadIdx.x;
    const size_t row_off = ((size_t)(r + a.o) * a.W + a.o) * a.D + k;
    const int last = a.Wc - 1;
    float vbuf[kCbcaPF];
    arm_pair abuf[kCbcaPF];
#pragma unroll
    for (int j = 0; j < kCbcaPF; ++j) {
        vbuf[j] = a.cv[row_off + (size_t)min(j, last) * a.D];
        const int ce = min(max(j - a.A, 0), last);
        abuf[j] = load_arms(a, r, ce, ph, ce + dq);
    }
    float acc = 0.f;
    const int nsteps = a.Wc + a.A;
    for (int c0 = 0; c0 < nsteps; c0 += kCbcaPF) {
#pragma unroll
        for (int j = 0; j < kCbcaPF; ++j) {
            const int c = c0 + j;
            const float v = vbuf[j];
            const uint32_t arms = combine(abuf[j]);
            {   // refill the slot with column c + kCbcaPF (clamped: surplus loads re-read the last column)
                const int cn = c + kCbcaPF;
                vbuf[j] = a.cv[row_off + (size_t)min(cn, last) * a.D];
                const int ce = min(max(cn - a.A, 0), last);
                abuf[j] = load_arms(a, r, ce, ph, ce + dq);
            }
            if (c < a.Wc) {
                if (v == v) acc = acc + v;  // NaN is skipped, the running sum carries on
                my[(c & mask) * BS] = acc;
            }
            const int ce = c - a.A;  // column whose segment sum can now be emitted
            if (ce >= 0 && ce < a.Wc && c < nsteps) {
                float e = 0.f;
                if (arms != 0xffffffffu) {
                    const int left = arms & 0xff, right = (arms >> 8) & 0xff;
                    const int lo = ce - left - 1;
                    const float hi_v = my[((ce + right) & mask) * BS];
                    const float lo_v = lo < 0 ? 0.f : my[(lo & mask) * BS];
                    e = hi_v - lo_v;
                }
                if (live) a.eh[row_off + (size_t)ce * a.D] = e;
            }
        }
    }
}

// pass V: steps 3-4 + normalisation (aggregation.cpp:123-221, cbca.py:166-171).  thread = (col, k) marches
// down the rows; the LDS ring holds the column prefix sums S3 and, packed in one word, n_h | top | bot of
// every row still needed (so the emit stage never re-reads the arms).
template <int BS>
__global__ __launch_bounds__(BS) void cbca_v_kernel(cbca_args a) {
    extern __shared__ float ring[];  // [2][ring][BS]: column prefix sums; packed (n_h, top, bot)
    const int t = blockIdx.x * BS + threadIdx.x;
    const int total = a.Wc * a.D;
    const bool live = t < total;
    const int tt = live ? t : total - 1;
    const int c = tt / a.D, k = tt - c * a.D;
    const int kk = k / a.subpix, ph = k - kk * a.subpix, q = c + a.d0 + kk;
    const int mask = a.ring - 1;
    float* s3 = ring + threadIdx.x;
    uint32_t* info = reinterpret_cast<uint32_t*>(ring) + (size_t)a.ring * BS + threadIdx.x;
    const size_t col_off = ((size_t)a.o * a.W + (c + a.o)) * a.D + k;
    const size_t row_stride = (size_t)a.W * a.D;
    const int last = a.Hc - 1;
    float ebuf[kCbcaPF], cbuf[kCbcaPF];
    arm_pair abuf[kCbcaPF];
#pragma unroll
    for (int j = 0; j < kCbcaPF; ++j) {
        const int rr = min(j, last);
        ebuf[j] = a.eh[col_off + (size_t)rr * row_stride];
        abuf[j] = load_arms(a, rr, c, ph, q);
        cbuf[j] = a.cv[col_off + (size_t)min(max(j - a.A, 0), last) * row_stride];
    }
    float acc = 0.f;
    uint32_t nacc = 0;
    const int nsteps = a.Hc + a.A;
    for (int r0 = 0; r0 < nsteps; r0 += kCbcaPF) {
#pragma unroll
        for (int j = 0; j < kCbcaPF; ++j) {
            const int r = r0 + j;
            const float e = ebuf[j];
            const float in = cbuf[j];
            const uint32_t arms = combine(abuf[j]);
            {   // refill with row r + kCbcaPF (clamped)
                const int rn = min(r + kCbcaPF, last);
                ebuf[j] = a.eh[col_off + (size_t)rn * row_stride];
                abuf[j] = load_arms(a, rn, c, ph, q);
                cbuf[j] = a.cv[col_off + (size_t)min(max(r + kCbcaPF - a.A, 0), last) * row_stride];
            }
            if (r < a.Hc) {
                acc = (r == 0) ? e : acc + e;
                s3[(r & mask) * BS] = acc;
                // ring word: running count N(r) = sum_{i<=r} n_h(i) in bits 0..19 (n_h = left+right, 0 outside the
                // right image), top in bits 20..25, bot in bits 26..31 (63,63 = this cell is outside).
                // The support size of step 4 (aggregation.cpp:199-215) then is two ring reads instead of loops:
                //   n_h(r) + sum_{i=1..top} n_h(r-i) + sum_{i=1..bot} n_h(r+i) = N(r+bot) - N(r-top-1)
                uint32_t tb = (63u << 20) | (63u << 26);
                if (arms != 0xffffffffu) {
                    nacc += (arms & 0xff) + ((arms >> 8) & 0xff);
                    tb = (((arms >> 16) & 0xff) << 20) | ((arms >> 24) << 26);
                }
                info[(r & mask) * BS] = nacc | tb;
            }
            const int re = r - a.A;
            if (re >= 0 && re < a.Hc && r < nsteps) {
                const uint32_t w = info[(re & mask) * BS];
                float step4 = 0.f, sum4 = 0.f;
                const int top = (w >> 20) & 63, bot = w >> 26;
                if (top != 63) {
                    const int lo = re - top - 1;
                    const float hi_v = s3[((re + bot) & mask) * BS];
                    const float lo_v = lo < 0 ? 0.f : s3[(lo & mask) * BS];
                    step4 = hi_v - lo_v;
                    const uint32_t n_hi = info[((re + bot) & mask) * BS] & 0xfffffu;
                    const uint32_t n_lo = lo < 0 ? 0u : (info[(lo & mask) * BS] & 0xfffffu);
                    sum4 = (float)(n_hi - n_lo + (uint32_t)(top + bot));  // small exact integers: any order
                }
                sum4 += 1.f;
                if (live) a.cv[col_off + (size_t)re * row_stride] = (in * 0.f + step4) / sum4;  // NaN stays NaN (cbca.py:145-146,168-171)
            }
        }
    }
}

// ---- phase-split variants of the two passes ----------------------------------------------------------------------
// The kernels above spend about a third of their instructions on per-step range tests and clamped 64-bit address
// arithmetic and are instruction-issue bound (tools/prof_cbca.sh).  When the scanned dimension is long enough the scan is
// cut into warm-up (prefix only), steady state (prefix + emit, no tests, incrementing pointers, four steps of loads in
// flight) and drain (emit only); arms are combined with two packed 16-bit minima.  Same arithmetic, same order.
typedef short cb_s2 __attribute__((ext_vector_type(2)));
typedef unsigned short cb_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cb_pk_min(uint32_t x, uint32_t y) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(cb_s2, x), __builtin_bit_cast(cb_s2, y)));
}
struct cb_arms { uint32_t lr, tb; };  // (left | right << 16), (top | bot << 16) as stored: int16 x 4
__device__ __forceinline__ cb_arms cb_load(const int16_t* base, size_t pixel) {
    const uint2 v = *reinterpret_cast<const uint2*>(base + pixel * 4);
    return {v.x, v.y};
}

__global__ __launch_bounds__(kBlock) void cbca_h_fast_kernel(cbca_args a) {
    extern __shared__ float ring[];  // [ring][kBlock]
    const int t = blockIdx.x * kBlock + threadIdx.x;
    const int total = a.Hc * a.D;
    const bool live = t < total;
    const int tt = live ? t : total - 1;
    const int r = tt / a.D, k = tt - r * a.D;
    const int kk = k / a.subpix, ph = k - kk * a.subpix, dq = a.d0 + kk;
    const int mask = a.ring - 1, A = a.A, Wc = a.Wc, D = a.D;
    const int Wr = ph == 0 ? Wc : Wc - 1;
    float* my = ring + threadIdx.x;
    for (int s = 0; s < a.ring; ++s) my[s * kBlock] = 0.f;  // S1 of columns < 0 is 0 (aggregation.cpp:113-114)
    const size_t row_off = ((size_t)(r + a.o) * a.W + a.o) * D + k;
    const float* pv = a.cv + row_off;   // cost of column c
    float* pe = a.eh + row_off;         // segment sum of column ce = c - A
    const int16_t* aL = a.armsL + (size_t)r * Wc * 4;
    const int16_t* aR = a.armsR[ph] + (size_t)r * Wr * 4;
    float acc = 0.f;
    auto prefix = [&](float v, int c) {
        acc = (v == v) ? acc + v : acc;  // NaN is skipped, the running sum carries on
        my[(c & mask) * kBlock] = acc;
    };
    auto emit = [&](cb_arms l, cb_arms rr, int ce) {
        const int q = ce + dq;
        const bool inside = (q >= 0) & (q <= Wr - 1);
        const uint32_t lr = cb_pk_min(l.lr, rr.lr);
        const int left = (int)(lr & 0xffffu), right = (int)(lr >> 16);
        const float hi_v = my[((ce + right) & mask) * kBlock];
        const float lo_v = my[((ce - left - 1) & mask) * kBlock];
        const float e = inside ? hi_v - lo_v : 0.f;
        if (live) *pe = e;
        pe += D;
    };
    auto right_px = [&](int ce) { return (size_t)min(max(ce + dq, 0), Wr - 1); };
    int c = 0;
    for (; c < A; ++c) {  // warm-up: columns whose segment cannot be closed yet
        prefix(*pv, c);
        pv += D;
    }
    // steady state, four columns per trip, the next four in flight
    float vb[4];
    cb_arms lb[4], rb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        vb[j] = pv[(size_t)min(j, Wc - 1 - c) * D];
        lb[j] = cb_load(aL, (size_t)(c - A + j));
        rb[j] = cb_load(aR, right_px(c - A + j));
    }
    for (; c + 4 <= Wc; c += 4) {
        float vn[4];
        cb_arms ln[4], rn[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cn = min(c + 4 + j, Wc - 1);  // (the last trips re-read the last column; its value is not used)
            vn[j] = pv[(size_t)(cn - c) * D];
            ln[j] = cb_load(aL, (size_t)min(c + 4 + j - A, Wc - 1));
            rn[j] = cb_load(aR, right_px(min(c + 4 + j - A, Wc - 1)));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            prefix(vb[j], c + j);
            emit(lb[j], rb[j], c + j - A);
        }
        pv += (size_t)4 * D;
#pragma unroll
        for (int j = 0; j < 4; ++j) { vb[j] = vn[j]; lb[j] = ln[j]; rb[j] = rn[j]; }
    }
    for (; c < Wc; ++c) {  // up to three leftover columns
        prefix(*pv, c);
        pv += D;
        emit(cb_load(aL, (size_t)(c - A)), cb_load(aR, right_px(c - A)), c - A);
    }
    for (; c < Wc + A; ++c)  // drain
        emit(cb_load(aL, (size_t)(c - A)), cb_load(aR, right_px(c - A)), c - A);
}

// SIGN: pass H marked the cells whose input cost was NaN in the sign bit of E_h (costs >= +0 only); the input volume is not read
template <bool SIGN>
__global__ __launch_bounds__(kBlock) void cbca_v_fast_kernel(cbca_args a) {
    extern __shared__ float ring[];  // [2][ring][kBlock]: column prefix sums; packed (N, top, bot)
    const int t = blockIdx.x * kBlock + threadIdx.x;
    const int total = a.Wc * a.D;
    const bool live = t < total;
    const int tt = live ? t : total - 1;
    const int c = tt / a.D, k = tt - c * a.D;
    const int kk = k / a.subpix, ph = k - kk * a.subpix, q = c + a.d0 + kk;
    const int mask = a.ring - 1, A = a.A, Hc = a.Hc, Wc = a.Wc;
    const int Wr = ph == 0 ? Wc : Wc - 1;
    const bool inside = (q >= 0) & (q <= Wr - 1);  // the same right column for every row
    const int qq = inside ? q : 0;
    float* s3 = ring + threadIdx.x;
    uint32_t* info = reinterpret_cast<uint32_t*>(ring) + (size_t)a.ring * kBlock + threadIdx.x;
    for (int s = 0; s < a.ring; ++s) { s3[s * kBlock] = 0.f; info[s * kBlock] = 0u; }  // row -1: zero sums, zero counts
    const size_t col_off = ((size_t)a.o * a.W + (c + a.o)) * a.D + k;
    const size_t row_stride = (size_t)a.W * a.D;
    const float* pe = a.eh + col_off;   // E_h of row r
    const float* pin = a.cv + col_off;  // input cost of row re = r - A (only its NaN-ness matters)
    float* pout = a.cv + col_off;
    const int16_t* aL = a.armsL + (size_t)c * 4;
    const int16_t* aR = a.armsR[ph] + (size_t)qq * 4;
    const size_t strideL = (size_t)Wc, strideR = (size_t)Wr;  // pixels per arms row
    float acc = 0.f;
    uint32_t nacc = 0;
    uint32_t nh = 0;  // SIGN: bit i = the input cost of row (current - i) was NaN
    auto prefix = [&](float e, cb_arms l, cb_arms rr, int r) {
        if (SIGN) {
            nh = (nh << 1) | (__float_as_uint(e) >> 31);
            e = fabsf(e);
        }
        acc = (r == 0) ? e : acc + e;
        s3[(r & mask) * kBlock] = acc;
        const uint32_t lr = cb_pk_min(l.lr, rr.lr), tb = cb_pk_min(l.tb, rr.tb);
        // ring word: running count N(r) of n_h = left + right in bits 0..19, top in bits 20..25, bot in bits 26..31
        // (63, 63 = this cell is outside the right image, n_h = 0)
        nacc += inside ? (lr & 0xffffu) + (lr >> 16) : 0u;
        const uint32_t word = inside ? (((tb & 0xffffu) << 20) | ((tb >> 16) << 26)) : ((63u << 20) | (63u << 26));
        info[(r & mask) * kBlock] = nacc | word;
    };
    auto emit = [&](float in, int re) {
        const uint32_t w = info[(re & mask) * kBlock];
        const int top = (w >> 20) & 63, bot = w >> 26;
        const bool cell = top != 63;
        const int hi_i = cell ? re + bot : re, lo_i = cell ? re - top - 1 : re;
        const float step = s3[(hi_i & mask) * kBlock] - s3[(lo_i & mask) * kBlock];
        const uint32_t n = (info[(hi_i & mask) * kBlock] & 0xfffffu) - (info[(lo_i & mask) * kBlock] & 0xfffffu) + (uint32_t)(top + bot);
        const float step4 = cell ? step : 0.f;
        const float sum4 = (cell ? (float)n : 0.f) + 1.f;  // small exact integers: any order
        float res;
        if (SIGN) res = ((nh >> A) & 1u) ? c_nan() : step4 / sum4;
        else res = (in * 0.f + step4) / sum4;  // NaN stays NaN (cbca.py:145-146,168-171)
        if (live) *pout = res;
        pout += row_stride;
    };
    // every stream advances by one row per step through its own pointer; the loads of the next four rows use the
    // wave-uniform offsets j * stride (no per-load multiplications), and the steady loop is unrolled over two four-row
    // buffers so that no register copies are needed between trips
    const int16_t* pl = aL;
    const int16_t* pr = aR;
    const size_t strideL4 = strideL * 4, strideR4 = strideR * 4;  // int16 elements per arms row
    int r = 0;
    for (; r < A; ++r) {  // warm-up
        prefix(*pe, cb_load(pl, 0), cb_load(pr, 0), r);
        pe += row_stride;
        pl += strideL4;
        pr += strideR4;
    }
    struct quad { float e[4], in[4]; cb_arms l[4], rr[4]; };
    auto load_quad = [&](quad& qd, int ahead) {  // rows r+ahead .. r+ahead+3 of the prefix streams, re+ahead.. of the input
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            qd.e[j] = pe[(size_t)(ahead + j) * row_stride];
            qd.in[j] = SIGN ? 0.f : pin[(size_t)(ahead + j) * row_stride];
            qd.l[j] = cb_load(pl, (size_t)(ahead + j) * strideL);
            qd.rr[j] = cb_load(pr, (size_t)(ahead + j) * strideR);
        }
    };
    auto run_quad = [&](const quad& qd) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            prefix(qd.e[j], qd.l[j], qd.rr[j], r + j);
            emit(qd.in[j], r + j - A);
        }
        r += 4;
        pe += 4 * row_stride;
        pin += 4 * row_stride;
        pl += 4 * strideL4;
        pr += 4 * strideR4;
    };
    if (r + 4 <= Hc) {
        quad qa, qb;
        load_quad(qa, 0);
        for (;;) {  // invariant: the buffer about to run holds rows r .. r+3, all inside the image
            if (r + 8 > Hc) { run_quad(qa); break; }
            load_quad(qb, 4);
            run_quad(qa);
            if (r + 8 > Hc) { run_quad(qb); break; }
            load_quad(qa, 4);
            run_quad(qb);
        }
    }
    for (; r < Hc; ++r) {
        prefix(*pe, cb_load(pl, 0), cb_load(pr, 0), r);
        pe += row_stride;
        pl += strideL4;
        pr += strideR4;
        emit(SIGN ? 0.f : *pin, r - A);
        pin += row_stride;
    }
    for (; r < Hc + A; ++r) {
        if (SIGN) nh <<= 1;
        emit(SIGN ? 0.f : *pin, r - A);
        pin += row_stride;
    }
}

// ---- pass H on whole rows ------------------------------------------------------------------------------------------------
// What bounds the phase-split pass H is its memory pattern, not its arithmetic (tools/ubench/cbca_pattern.hip, and the kernel
// with its stores removed runs in 0.66 instead of 1.5 ms at 2048^2 x 129): with D = 129 a wavefront's 64 cells are 256 bytes
// that straddle three cache lines, written 516 bytes further every step, each line finished by another wavefront much later.
// Here a workgroup owns R WHOLE image rows (thread = (row, disparity), R*D threads rounded up to whole wavefronts), so the
// segment sums of four consecutive columns are 4*D*4 contiguous bytes per row: they are collected in LDS and leave as 16 bytes
// per lane.  Same scan, same order, same results.
// SRC = 1: the matching costs are the Hamming distances of the handle's census codes (one word per pixel: windows up to 5x5),
// computed here instead of being read from a float32 volume that then never exists: census + cbca without the 4 B/cell write
// and the 4 B/cell read of the cost volume.  A cell's cost is a number iff the cost kernel would have written one: inside the
// pixel's valid interval when cv_masked left one (k_fused.hip: build_range_kernel), else where both census windows fit.
// SIGN as in the phase-split pass V above.
#include "pmx_buf.h"

static constexpr int kRowsT = 576;  // most threads per workgroup (R rows x D disparities, whole wavefronts)
static constexpr int kChunk = 8;    // columns per flush of the output stage
static constexpr int kStage = 2 * kChunk;

__device__ __forceinline__ uint32_t cb_lr16(uint32_t x) { return __builtin_amdgcn_perm(0u, x, 0x0c010c00u); }  // (left, right) as 16-bit halves

__global__ __launch_bounds__(256) void pack_arms_kernel(const int16_t* __restrict__ arms, int Hc, int Wsrc, uint32_t* __restrict__ rows,
                                                        int pitch, int xoff) {
    const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
    if (c >= Wsrc) return;
    const short4 s = *reinterpret_cast<const short4*>(arms + ((size_t)r * Wsrc + c) * 4);
    rows[(size_t)r * pitch + xoff + c] = (uint32_t)s.x | ((uint32_t)s.y << 8) | ((uint32_t)s.z << 16) | ((uint32_t)s.w << 24);
}

// census source: the volume first exists as pass V's output, which covers the image without its border of `o` pixels; the
// border cells are the NaN the census cost kernel would have left there
__global__ __launch_bounds__(256) void cbca_border_nan_kernel(float* __restrict__ cv, int H, int W, int D, int o) {
    // the border as a list of pixels: o full rows on top, o at the bottom, 2o pixels of every row between
    const int top = o * W, sides = (H - 2 * o) * 2 * o;
    const size_t cells = (size_t)(2 * top + sides) * D;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < cells; i += (size_t)gridDim.x * 256) {
        const int b = (int)(i / D), k = (int)(i - (size_t)b * D);
        int r, c;
        if (b < top) { r = b / W; c = b - r * W; }
        else if (b < top + sides) { const int j = b - top; r = o + j / (2 * o); const int e = j - (r - o) * 2 * o; c = e < o ? e : W - 2 * o + e; }
        else { const int j = b - top - sides; r = H - o + j / W; c = j - (r - (H - o)) * W; }
        cv[((size_t)r * W + c) * D + k] = c_nan();
    }
}

template <bool SIGN, int SRC>
__global__ __launch_bounds__(kRowsT) void cbca_h_rows_kernel(cbca_args a) {
    extern __shared__ float lds[];
    const int T = blockDim.x, R = a.R, D = a.D, A = a.A, Wc = a.Wc;
    const int mask = a.ring - 1;
    float* stage = lds + (size_t)a.ring * T;  // [R][2 * kChunk columns][D]
    const int tid = threadIdx.x;
    const int lrow = min(tid / D, R - 1), k = tid - (tid / D) * D;
    const int r0 = blockIdx.x * R;
    const bool owner = tid < R * D && r0 + lrow < a.Hc;
    const int r = min(r0 + lrow, a.Hc - 1);
    const int kk = k / a.subpix, ph = k - kk * a.subpix, dq = a.d0 + kk;
    const int Wr = ph == 0 ? Wc : Wc - 1;
    float* my = lds + tid;
    for (int s = 0; s < a.ring; ++s) my[s * T] = 0.f;  // S1 of columns < 0 is 0 (aggregation.cpp:113-114)
    const uint32_t T4 = (uint32_t)T * 4u;  // bytes from one ring slot to the next
    auto slot = [&](int i) -> float& { return *reinterpret_cast<float*>(reinterpret_cast<char*>(my) + __umul24((uint32_t)(i & mask), T4)); };
    float* st = stage + (size_t)lrow * kStage * D + k;  // + (ce & (kStage - 1)) * D
    const size_t row_off = ((size_t)(r + a.o) * a.W + a.o) * D + k;
    const float* pv = a.cv + row_off;  // SRC 0: cost of column c
    const __amdgpu_buffer_rsrc_t rsL = __builtin_amdgcn_make_buffer_rsrc((void*)a.armsL8, 0, a.bytesL8, kRsrcWord3);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)a.armsR8, 0, a.bytesR8, kRsrcWord3);
    const unsigned offL = (unsigned)r * (unsigned)a.pitchL * 4u;  // + 4 * ce
    const unsigned offR = ((unsigned)ph * a.phase_words + (unsigned)r * (unsigned)a.pitchR + (unsigned)(a.padR + dq)) * 4u;  // pads: every index is valid
    // SRC 1: codes and valid intervals of the row, full-image coordinates
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void*)a.codes, 0, a.codes_bytes, kRsrcWord3);
    const unsigned pix0 = (unsigned)(r + a.o) * (unsigned)a.W + (unsigned)a.o;          // pixel of column c = 0
    const unsigned offCL = (a.offCL + pix0) * 4u, offCR = (a.offCR + pix0 + (unsigned)dq) * 4u;  // + 4 * c (guard words around the images)
    const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc((void*)a.range, 0, (unsigned)a.H * (unsigned)a.W * 4u, kRsrcWord3);
    constexpr bool has_range = SRC == 2;  // SRC 2: census codes + the valid intervals cv_masked left (1 and 3: census geometry)
    const bool row_ok = (r + a.o >= a.cb) & (r + a.o < a.H - a.cb);
    const unsigned wvalid = (unsigned)(a.W - 2 * a.cb);
    auto census = [&](uint32_t cl, uint32_t cr, uint32_t rg, int c) {
        bool ok;
        if (has_range) ok = (k >= (int)(rg & 0xffffu)) & (k < (int)(rg >> 16));
        else ok = row_ok & ((unsigned)(c + a.o - a.cb) < wvalid) & ((unsigned)(c + a.o + dq - a.cb) < wvalid);
        return ok ? (float)__popc(cl ^ cr) : c_nan();
    };
    // SRC 3: census geometry alone and a crop equal to the census border (the pipeline's case): a cell's cost is a number iff
    // its right pixel c + dq lies in the cropped image - the very test of the segment sum below - so no NaN is ever made or
    // tested and no history of them kept
    constexpr bool kGeo = SRC == 3;
    float acc = 0.f;
    uint32_t hist = 0;  // SIGN: bit i = the cost of column (newest - i) was NaN
    auto prefix = [&](float v, int c) {
        acc = (v == v) ? acc + v : acc;  // NaN is skipped, the running sum carries on
        slot(c) = acc;
        if (SIGN) hist = (hist << 1) | (v == v ? 0u : 1u);
    };
    auto prefix_geo = [&](uint32_t cl, uint32_t cr, int c) {
        const float v = (float)__popc(cl ^ cr);
        acc = ((unsigned)(c + dq) < (unsigned)Wc) ? acc + v : acc;
        slot(c) = acc;
    };
    // segment sum of column ce; `age` = how many columns newer than ce + A the newest prefix is (SIGN)
    auto segment = [&](uint32_t l8, uint32_t r8, int ce, int age) {
        const int q = ce + dq;
        const bool inside = (q >= 0) & (q <= Wr - 1);
        const uint32_t lr = cb_pk_min(cb_lr16(l8), cb_lr16(r8));
        const int left = (int)(lr & 0xffffu), right = (int)(lr >> 16);
        const float hi_v = slot(ce + right);   // (v_mul_u32_u24: full rate, v_mul_lo_u32 is a quarter)
        const float lo_v = slot(ce - left - 1);
        if (kGeo) return inside ? hi_v - lo_v : -0.f;  // outside: E_h = 0 and "the input was NaN"
        float e = inside ? hi_v - lo_v : 0.f;
        if (SIGN) e = __uint_as_float(__float_as_uint(e) | (((hist >> (A + age)) & 1u) << 31));
        return e;
    };
    auto put = [&](float e, int ce) {  // ... into the output stage
        if (owner) st[(ce & (kStage - 1)) * D] = e;
    };
    // kChunk finished columns m0 .. of the block's rows leave LDS as 16 bytes per lane; the barrier in front also separates
    // this chunk's emits from the writes that reuse its half of the stage two chunks later
    const unsigned rows_left = (unsigned)min(R, a.Hc - r0);
    const unsigned row_floats = (unsigned)Wc * (unsigned)D;
    const __amdgpu_buffer_rsrc_t rsE = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.eh + ((size_t)(r0 + a.o) * a.W + a.o) * D), 0, 0x7ffffff0u, kRsrcWord3);
    const bool aligned = (A & 3) == 0;  // then a chunk is flushed right after the quad that emits its last column
    auto flush = [&](int m0) {
        __syncthreads();
        const unsigned nv = (unsigned)(kChunk / 4) * (unsigned)D;  // 16-byte vectors per row chunk (kChunk columns x D floats)
        for (unsigned i = tid; i < rows_left * nv; i += T) {
            const unsigned rr = i / nv, v = i - rr * nv;
            const u32x4 x = *reinterpret_cast<const u32x4*>(stage + ((size_t)rr * kStage + (m0 & (kStage - 1))) * D + 4 * v);
            const unsigned f = (unsigned)m0 * (unsigned)D + 4u * v;  // first float of the vector in its (cropped) row
            const unsigned off = (rr * (unsigned)a.W * (unsigned)D + f) * 4u;
            if (f + 4 <= row_floats) {
                __builtin_amdgcn_raw_buffer_store_b128(x, rsE, off, 0, 0);
            } else {  // the row ends inside this vector (its last chunk, Wc not a multiple of 4)
                __builtin_amdgcn_raw_buffer_store_b32(x.x, rsE, f < row_floats ? off : kOob, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(x.y, rsE, f + 1 < row_floats ? off + 4 : kOob, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(x.z, rsE, f + 2 < row_floats ? off + 8 : kOob, 0, 0);
            }
        }
        if (!aligned) __syncthreads();  // up to 3 newer columns are pending: the next emits reach into this chunk's half
    };
    auto arms1 = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned off, int ce) {
        return __builtin_amdgcn_raw_buffer_load_b32(rs, off + 4u * (unsigned)ce, 0, 0);
    };
    auto cost1 = [&](int c) {  // the cost of one column (warm-up, leftovers)
        if (SRC == 0) return pv[(size_t)c * D];
        const uint32_t cl = __builtin_amdgcn_raw_buffer_load_b32(rsC, offCL + 4u * (unsigned)c, 0, 0);
        const uint32_t cr = __builtin_amdgcn_raw_buffer_load_b32(rsC, offCR + 4u * (unsigned)c, 0, 0);
        const uint32_t rg = has_range ? __builtin_amdgcn_raw_buffer_load_b32(rsG, (pix0 + (unsigned)c) * 4u, 0, 0) : 0u;
        return census(cl, cr, rg, c);
    };
    auto step1 = [&](int c) {  // the prefix update of one column (warm-up, leftovers)
        if (kGeo) {
            prefix_geo(__builtin_amdgcn_raw_buffer_load_b32(rsC, offCL + 4u * (unsigned)c, 0, 0),
                       __builtin_amdgcn_raw_buffer_load_b32(rsC, offCR + 4u * (unsigned)c, 0, 0), c);
        } else {
            prefix(cost1(c), c);
        }
    };
    int c = 0, flushed = 0;
    for (; c < A; ++c) step1(c);  // warm-up: columns whose segment cannot be closed yet
    // steady state: quads of columns (four prefixes, then four emits: one LDS round trip per quad), two quads in registers, the
    // loop unrolled over the pair so that no register copies (which would wait for the loads) separate the trips
    struct quad { float v[4]; u32x4 cl, cr, rg, l, rr; };
    auto load_quad = [&](quad& g, int c0) {  // columns c0 .. c0+3 (all < Wc) and the arms of c0-A ..
        if (SRC == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) g.v[j] = pv[(size_t)(c0 + j) * D];
        } else {
            g.cl = __builtin_amdgcn_raw_buffer_load_b128(rsC, offCL + 4u * (unsigned)c0, 0, 0);
            g.cr = __builtin_amdgcn_raw_buffer_load_b128(rsC, offCR + 4u * (unsigned)c0, 0, 0);
            if (has_range) g.rg = __builtin_amdgcn_raw_buffer_load_b128(rsG, (pix0 + (unsigned)c0) * 4u, 0, 0);
            else g.rg = u32x4{0u, 0u, 0u, 0u};
        }
        g.l = __builtin_amdgcn_raw_buffer_load_b128(rsL, offL + 4u * (unsigned)(c0 - A), 0, 0);
        g.rr = __builtin_amdgcn_raw_buffer_load_b128(rsR, offR + 4u * (unsigned)(c0 - A), 0, 0);
    };
    auto run_quad = [&](const quad& g) {
        if (SRC == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) prefix(g.v[j], c + j);
        } else if (kGeo) {
            prefix_geo(g.cl.x, g.cr.x, c);
            prefix_geo(g.cl.y, g.cr.y, c + 1);
            prefix_geo(g.cl.z, g.cr.z, c + 2);
            prefix_geo(g.cl.w, g.cr.w, c + 3);
        } else {
            prefix(census(g.cl.x, g.cr.x, g.rg.x, c), c);
            prefix(census(g.cl.y, g.cr.y, g.rg.y, c + 1), c + 1);
            prefix(census(g.cl.z, g.cr.z, g.rg.z, c + 2), c + 2);
            prefix(census(g.cl.w, g.cr.w, g.rg.w, c + 3), c + 3);
        }
        // (the four segments first, their ring reads in one batch; then the four stage writes - both live in the one LDS array, so
        //  the compiler keeps reads and writes in program order)
        const float e0 = segment(g.l.x, g.rr.x, c - A, 3), e1 = segment(g.l.y, g.rr.y, c + 1 - A, 2);
        const float e2 = segment(g.l.z, g.rr.z, c + 2 - A, 1), e3 = segment(g.l.w, g.rr.w, c + 3 - A, 0);
        put(e0, c - A); put(e1, c + 1 - A); put(e2, c + 2 - A); put(e3, c + 3 - A);
        c += 4;
        while (flushed + kChunk <= c - A) {  // (uniform)
            flush(flushed);
            flushed += kChunk;
        }
    };
    if (c + 4 <= Wc) {
        quad ga, gb;
        load_quad(ga, c);
        for (;;) {  // invariant: the quad about to run holds columns c .. c+3, all inside the image
            if (c + 8 > Wc) { run_quad(ga); break; }
            load_quad(gb, c + 4);
            run_quad(ga);
            if (c + 8 > Wc) { run_quad(gb); break; }
            load_quad(ga, c + 4);
            run_quad(gb);
        }
    }
    for (; c < Wc + A; ++c) {  // leftover columns, then the drain (emit only)
        if (c < Wc) step1(c);
        else if (SIGN) hist <<= 1;
        put(segment(arms1(rsL, offL, c - A), arms1(rsR, offR, c - A), c - A, 0), c - A);
        if (flushed + kChunk <= c + 1 - A) {
            flush(flushed);
            flushed += kChunk;
        }
    }
    if (flushed < Wc) flush(flushed);  // the row's last, partial chunk (the stores past the row's end are dropped)
}

// ---- pass V through buffer instructions ---------------------------------------------------------------------------------
// The phase-split pass V carries five per-lane 64-bit pointers and two buffers of loads: 140 registers, 3 wavefronts per SIMD,
// which is what hides (or does not hide) the memory latency at sizes that have more wavefronts than that to offer.  Here the
// volume and the byte arms of the whole-row pass H move through buffer instructions whose descriptor is re-based every row
// (uniform SALU work) with a lane's cell at a fixed offset inside the row; the ring holds ONE 8-byte entry per row (column
// prefix sum S3 | packed word); the steps run in quads (four prefixes, then four emits whose ring reads are issued together).
// Same arithmetic and order as cbca_v_fast_kernel.  SIGN as there.
__device__ __forceinline__ uint32_t cb_tb16(uint32_t x) { return __builtin_amdgcn_perm(0u, x, 0x0c030c02u); }  // (top, bottom)

template <bool SIGN, int BS, bool ROWDESC>  // BS threads per workgroup: its cells are BS * 4 contiguous bytes of every row
__global__ __launch_bounds__(BS) void cbca_v_buf_kernel(cbca_args a) {
    extern __shared__ float ring[];  // [ring][BS] x (S3, word)
    const int t = blockIdx.x * BS + threadIdx.x;
    const int total = a.Wc * a.D;
    const bool live = t < total;
    const int tt = live ? t : total - 1;
    const int c = tt / a.D, k = tt - c * a.D;
    const int kk = k / a.subpix, ph = k - kk * a.subpix, q = c + a.d0 + kk;
    const int mask = a.ring - 1, A = a.A, Hc = a.Hc, Wc = a.Wc;
    const int Wr = ph == 0 ? Wc : Wc - 1;
    const bool inside = (q >= 0) & (q <= Wr - 1);  // the same right column for every row
    uint2* ent = reinterpret_cast<uint2*>(ring) + threadIdx.x;
    for (int s = 0; s < a.ring; ++s) ent[s * BS] = make_uint2(0u, 0u);  // row -1: zero sums, zero counts
    const unsigned row_bytes = (unsigned)a.W * (unsigned)a.D * 4u;
    const unsigned voff = ((unsigned)(c + a.o) * (unsigned)a.D + (unsigned)k) * 4u;
    const unsigned voff_st = live ? voff : kOob;
    const size_t row_stride = (size_t)a.W * a.D;
    // uniform row pointers, advanced as the scan goes
    const float* e_row = a.eh + (size_t)a.o * row_stride;    // E_h, row r
    const float* in_row = a.cv + (size_t)a.o * row_stride;   // input cost, row r - A (only its NaN-ness matters; unused when SIGN)
    float* out_row = a.cv + (size_t)a.o * row_stride;        // output, row r - A
    const uint32_t* l_row = a.armsL8;                        // arms, row r
    const uint32_t* r_row = a.armsR8;
    // Descriptors are re-based once per QUAD of rows (a volume is larger than a descriptor's 4 GB of offsets); inside the quad the
    // row is chosen by the instruction's SCALAR offset (it IS part of the range check on gfx950: the descriptors span the quad's
    // four rows; a lane's own offset stays inside one row, a dead lane's out-of-range offset still drops its access).  (Re-basing every row was 6 scalar instructions per memory instruction -
    // the pass is issue-bound with as many scalar as vector instructions)  PMX_CBCA_DBG empties descriptors.
    const unsigned quad_bytes = 4u * row_bytes;
    const unsigned e_bytes = quad_bytes, st_bytes = quad_bytes;
    auto rs_at = [&](const void* row_ptr, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc((void*)row_ptr, 0, bytes, kRsrcWord3); };
    auto ld = [&](const float* row_ptr) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_at(row_ptr, row_bytes), voff, 0, 0)); };
    const unsigned offL = (unsigned)c * 4u;
    const unsigned offR = ((unsigned)ph * a.phase_words + (unsigned)(a.padR + q)) * 4u;  // (pads: every q of the lane's cells is readable)
    const unsigned armsL_bytes = 0x7ffffff0u, armsR_bytes = 0x7ffffff0u;
    const unsigned pitchL_bytes = (unsigned)a.pitchL * 4u, pitchR_bytes = (unsigned)a.pitchR * 4u;
    float acc = 0.f;
    uint32_t nacc = 0;
    uint32_t nh = 0;  // SIGN: bit i = the input cost of row (newest - i) was NaN
    auto prefix = [&](float e, uint32_t l8, uint32_t r8, int r) {
        if (SIGN) {
            nh = (nh << 1) | (__float_as_uint(e) >> 31);
            e = fabsf(e);
        }
        acc = (r == 0) ? e : acc + e;
        const uint32_t lr = cb_pk_min(cb_lr16(l8), cb_lr16(r8)), tb = cb_pk_min(cb_tb16(l8), cb_tb16(r8));
        // the row's packed word: N(r) in bits 0..19, top in bits 20..25, bot in bits 26..31 (63, 63 = outside the right image)
        nacc += inside ? (lr & 0xffffu) + (lr >> 16) : 0u;
        const uint32_t word = nacc | (inside ? (((tb & 0xffffu) << 20) | ((tb >> 16) << 26)) : ((63u << 20) | (63u << 26)));
        ent[(r & mask) * BS] = make_uint2(__float_as_uint(acc), word);
    };
    // aggregated cost of row re; `age`: how many rows newer than re + A the newest prefix is (SIGN)
    auto emit = [&](float in, int re, int age, __amdgpu_buffer_rsrc_t rsO, unsigned srow) {
        const uint32_t w = ent[(re & mask) * BS].y;
        const int top = (w >> 20) & 63, bot = w >> 26;
        const bool cell = top != 63;
        const int hi_i = cell ? re + bot : re, lo_i = cell ? re - top - 1 : re;
        const uint2 hi = ent[(hi_i & mask) * BS], lo = ent[(lo_i & mask) * BS];
        const float step = __uint_as_float(hi.x) - __uint_as_float(lo.x);
        const uint32_t n = (hi.y & 0xfffffu) - (lo.y & 0xfffffu) + (uint32_t)(top + bot);
        const float step4 = cell ? step : 0.f;
        const float sum4 = (cell ? (float)n : 0.f) + 1.f;  // small exact integers: any order
        float res;
        if (SIGN) res = ((nh >> (A + age)) & 1u) ? c_nan() : step4 / sum4;
        else res = (in * 0.f + step4) / sum4;  // NaN stays NaN (cbca.py:145-146,168-171)
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(res), rsO, voff_st, srow, 0);
    };
    struct row_in { float e, in; uint32_t l, rr; };
    struct row_rs { __amdgpu_buffer_rsrc_t e, in, l, r; };
    auto rs_rows = [&](int ahead) {  // descriptors of row r + ahead of the prefix streams, r + ahead - A of the input
        return row_rs{rs_at(e_row + (size_t)ahead * row_stride, e_bytes), rs_at(in_row + (size_t)ahead * row_stride, quad_bytes),
                      rs_at(l_row + (size_t)ahead * a.pitchL, armsL_bytes), rs_at(r_row + (size_t)ahead * a.pitchR, armsR_bytes)};
    };
    auto load_row = [&](const row_rs& rs, int j) {  // row j below the descriptors' row
        row_in x;
        x.e = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs.e, voff, (unsigned)j * row_bytes, 0));
        x.in = SIGN ? 0.f : __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs.in, voff, (unsigned)j * row_bytes, 0));
        x.l = __builtin_amdgcn_raw_buffer_load_b32(rs.l, offL, (unsigned)j * pitchL_bytes, 0);
        x.rr = __builtin_amdgcn_raw_buffer_load_b32(rs.r, offR, (unsigned)j * pitchR_bytes, 0);
        return x;
    };
    auto advance = [&](int n) {
        e_row += (size_t)n * row_stride;
        l_row += (size_t)n * a.pitchL;
        r_row += (size_t)n * a.pitchR;
    };
    int r = 0;
    for (; r < A; ++r) {  // warm-up
        const row_in x = load_row(rs_rows(0), 0);
        prefix(x.e, x.l, x.rr, r);
        advance(1);
    }
    // steady state: quads of rows, two quads in registers, the loop unrolled over the pair (no register copies between trips)
    struct quad { row_in x[4]; };
    // ROWDESC: descriptors re-based every row after all.  Six more scalar instructions per memory instruction, and still the
    // faster form when a row of the volume is more than 5 MB (10000^2 x 129: 33.8 against 35.6 ms for the quad form on one box,
    // 31.9 against 32.6 on another; 6000^2 x 129: 12.3 against 11.5, 4096^2 x 257: 13.2 against 12.4, 4096^2 x 129: 6.70 against
    // 6.05 and 6.48 against 6.47, 2048^2 x 129: 2.17 against 1.82) - the launcher picks by the row size.
    auto load_quad = [&](quad& g, int ahead) {
        if (ROWDESC) {
#pragma unroll
            for (int j = 0; j < 4; ++j) g.x[j] = load_row(rs_rows(ahead + j), 0);
        } else {
            const row_rs rs = rs_rows(ahead);
#pragma unroll
            for (int j = 0; j < 4; ++j) g.x[j] = load_row(rs, j);
        }
    };
    auto run_quad = [&](const quad& g) {
#pragma unroll
        for (int j = 0; j < 4; ++j) prefix(g.x[j].e, g.x[j].l, g.x[j].rr, r + j);
        const __amdgpu_buffer_rsrc_t rsO = rs_at(out_row, st_bytes);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (ROWDESC) emit(g.x[j].in, r + j - A, 3 - j, rs_at(out_row + (size_t)j * row_stride, st_bytes), 0u);
            else emit(g.x[j].in, r + j - A, 3 - j, rsO, (unsigned)j * row_bytes);
        }
        r += 4;
        advance(4);
        in_row += (size_t)4 * row_stride;
        out_row += (size_t)4 * row_stride;
    };
    if (r + 4 <= Hc) {
        quad ga, gb;
        load_quad(ga, 0);
        for (;;) {  // invariant: the quad about to run holds rows r .. r+3, all inside the image
            if (r + 8 > Hc) { run_quad(ga); break; }
            load_quad(gb, 4);
            run_quad(ga);
            if (r + 8 > Hc) { run_quad(gb); break; }
            load_quad(ga, 4);
            run_quad(gb);
        }
    }
    for (; r < Hc; ++r) {  // leftover rows
        const row_in x = load_row(rs_rows(0), 0);
        prefix(x.e, x.l, x.rr, r);
        advance(1);
        emit(x.in, r - A, 0, rs_at(out_row, st_bytes), 0u);
        in_row += row_stride;
        out_row += row_stride;
    }
    for (; r < Hc + A; ++r) {  // drain
        if (SIGN) nh <<= 1;
        emit(SIGN ? 0.f : ld(in_row), r - A, 0, rs_at(out_row, st_bytes), 0u);
        in_row += row_stride;
        out_row += row_stride;
    }
}

// ---- census costs + both scans in ONE marching kernel: exact integer sums ----------------------------------------------------
// With census costs every quantity of the aggregation is a small integer: a cost is at most 32 (one code word), a horizontal
// segment sum at most 9 x 32, the column prefix of those stays below 2^24 for any image of fewer than 58 000 rows - every float32
// operation of aggregation.cpp:28-121 is exact whatever its order, and the only rounding of a cell is its final division.  So the
// horizontal sums need not come from a row-long prefix, and the scratch volume E_h (written by pass H, read by pass V: 8 of the
// 12 bytes per cell the two passes move) need not exist.  A workgroup owns NC columns (NC * D threads, thread = (column,
// disparity), cells contiguous in the volume's rows as in pass V) and marches down the rows in quads:
//   a) the codes and byte arms of the quad's rows arrive in LDS (one or two dwords per thread and quad, loaded two quads ahead):
//      NC + 8 left codes, NC + 8 + D - 1 right codes, NC + NC + D - 1 arms per row;
//   b) every thread computes the cost of its own cell and of its share of the 8 halo columns (4 on either side: the longest arm)
//      and leaves it as a BYTE in a row of 16 per disparity; one barrier;
//   c) a cell's horizontal segment sum is four v_dot4_u32_u8 of its disparity's 16 bytes with a 0/1 byte mask looked up by
//      (first, last) column; the column prefix S3 and the count N are 16-bit fields of ONE ring word (differences of prefixes are
//      below 2^16, so the fields may wrap: v_pk_sub_u16 takes both differences at once).  The vertical step: with arms of at
//      most 4 rows the row that leaves at step j of a quad is the row prefixed at step j of the quad before - its (top, bottom)
//      wait in four registers, the ring holds 12 rows, and the final division of two small integers is the hardware reciprocal
//      plus one Newton step (small_int_div: every pair checked against the IEEE division).
// HBM traffic of the aggregation: the 4-byte store per cell and the two small images.  Legal for a census source with one code
// word per pixel, subpix 1, arms of at most 4 (cbca_distance <= 5), D <= 1024 (pmx_launch_cbca decides).
struct cbca_march {
    int NC, T;                  // useful columns and threads of a workgroup
    int nsh;                    // halo cells per thread (1 or 2)
    unsigned ring_stride;       // bytes of a ring slot
    unsigned cost_off, code_off, arms_off, tab_off;  // LDS byte offsets (the ring starts at 0)
    unsigned cost_buf, code_buf, arms_buf;           // bytes of one of the two buffers of each
    int wprc, wpra;             // staged words per row: codes (NC + 8 left, NC + 8 + D - 1 right), arms (NC left, NC + D - 1 right)
    unsigned arms_bytes;        // one descriptor over armsL8 .. the end of armsR8
    unsigned armsR_word;        // where armsR8 starts in it (words)
};
static constexpr int kMarchSlots = 2;   // staged words per thread and quad, of each kind
static constexpr int kMarchRing = 12;
static constexpr int kMarchTab = 144 * 16;  // byte masks by (first column 0 .. 11, last column 4 .. 15)

#define PMX_AS3 __attribute__((address_space(3)))
__device__ __forceinline__ uint32_t lds_r32(uint32_t addr) { return *(const uint32_t PMX_AS3*)(uintptr_t)addr; }
__device__ __forceinline__ u32x4 lds_r128(uint32_t addr) { return *(const u32x4 PMX_AS3*)(uintptr_t)addr; }
__device__ __forceinline__ void lds_w32(uint32_t addr, uint32_t v) { *(uint32_t PMX_AS3*)(uintptr_t)addr = v; }
__device__ __forceinline__ void lds_w8(uint32_t addr, uint32_t v) { *(uint8_t PMX_AS3*)(uintptr_t)addr = (uint8_t)v; }
__device__ __forceinline__ void lds_w128(uint32_t addr, u32x4 v) { *(u32x4 PMX_AS3*)(uintptr_t)addr = v; }

// a.lo * b.lo + a.hi * b.hi + c on 16-bit halves in ONE instruction (v_dot2_u32_u16 / v_dot2_i32_i16): a packed pair of arms
// times two strides is an LDS address
typedef unsigned short pmx_us2 __attribute__((ext_vector_type(2)));
typedef short pmx_s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t udot2(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_udot2(__builtin_bit_cast(pmx_us2, a), __builtin_bit_cast(pmx_us2, b), c, false);
}
__device__ __forceinline__ int32_t sdot2(uint32_t a, uint32_t b, int32_t c) {
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(pmx_s2, a), __builtin_bit_cast(pmx_s2, b), c, false);
}

// a / b for integers 0 <= a < 65536, 1 <= b <= 1024 as float32, correctly rounded: one Newton step on the hardware reciprocal's
// quotient.  pmx_debug_small_division compares every pair with the IEEE division (tests/test_gpu_parity.py).
__device__ __forceinline__ float small_int_div(float af, float bf) {
    const float y = __builtin_amdgcn_rcpf(bf);
    const float q0 = af * y;
    const float r = __builtin_fmaf(-bf, q0, af);
    return __builtin_fmaf(r, y, q0);
}

__global__ __launch_bounds__(256) void small_division_check_kernel(unsigned* __restrict__ bad) {
    const unsigned i = blockIdx.x * 256 + threadIdx.x;  // a = i & 0xffff, b = (i >> 16) + 1
    const float af = (float)(i & 0xffffu), bf = (float)((i >> 16) + 1u);
    float ieee = af / bf;
    asm volatile("" : "+v"(ieee));
    if (__float_as_uint(small_int_div(af, bf)) != __float_as_uint(ieee)) atomicAdd(bad, 1u);
}

int pmx_launch_small_division_check(pmx_ctx* ctx, unsigned* host_count) {
    unsigned* dev = nullptr;
    PMX_HIP(pmx_pool_alloc(ctx, (void**)&dev, sizeof(unsigned)));
    hipError_t e = hipMemsetAsync(dev, 0, sizeof(unsigned), ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(small_division_check_kernel, dim3((1024u << 16) / 256u), dim3(256), 0, ctx->stream, dev);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(host_count, dev, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    pmx_pool_free(ctx, dev);  // (on every path)
    PMX_HIP(e);
    return PMX_OK;
}

// LDS of a workgroup (bytes): ring [12][NC * D] words | cost bytes [2][4 rows][D][16] | masks [144][16] | codes [2][wprc][4 rows]
// words | arms [2][2][wpra][4 rows] words: (left, right) and (top, bottom) as 16-bit pairs, expanded once by the staging thread.
// Codes and arms have the four rows of a quad as their FASTEST index: a thread's four rows of one pixel are one 16-byte read, and
// consecutive disparities read consecutive 16 bytes (no bank conflict; rows of 16 bytes 64 apart would collide eight ways).
// (8 wavefronts per SIMD: 64 registers, so that two workgroups of 15 wavefronts share a CU and fill each other's barriers -
// 24.7 against 26.8 ms at 10000^2 x 129 for the 71 registers the compiler would take)
template <int SRC>  // 3: census geometry with the crop equal to the census border; 1: census geometry, any crop; 2: the valid intervals cv_masked left
__global__ __launch_bounds__(1024, 8) void cbca_census_march_kernel(cbca_args a, cbca_march m) {
    if (__builtin_amdgcn_groupstaticsize() != 0) __builtin_trap();  // (LDS addresses below are absolute: the dynamic block starts at 0)
    const int D = a.D, NC = m.NC, T = m.T, Wc = a.Wc, Hc = a.Hc, W = a.W, o = a.o;
    const int tid = threadIdx.x;
    const int c0 = blockIdx.x * NC;
    // No lane-varying branch anywhere in the march.  A lane without a cell of its own (the last wavefront's tail) doubles the
    // workgroup's last cell - same reads, same values written to the same LDS bytes, its stores out of range; a lane without a
    // halo cell doubles the last halo cell.
    const int ci = min(tid, NC * D - 1);  // this thread's cell
    const int cloc = ci / D, k = ci - cloc * D;
    const int c = c0 + cloc, q = c + a.d0 + k;
    const bool live = (tid < NC * D) & (c < Wc);
    const bool inside = (q >= 0) & (q < Wc);
    // ring: bytes per slot (a multiple of the 32 banks: which bank a cell's word lies in does not depend on the slot, so lanes that
    // read different slots do not collide), this cell's word in a slot
    const uint32_t RS4 = m.ring_stride, own = (uint32_t)ci * 4u;
    // column tests of a cost (its row test is uniform): geometry of the census windows
    const unsigned wvalid = (unsigned)(W - 2 * a.cb);
    auto col_ok = [&](int cc, int kk) -> bool {  // cropped column cc, disparity index kk
        if (SRC == 3) return (unsigned)(cc + a.d0 + kk) < (unsigned)Wc;
        if (SRC == 2) return true;  // (the valid interval of the pixel decides, row by row)
        return (bool)(((unsigned)(cc + o - a.cb) < wvalid) & ((unsigned)(cc + o + a.d0 + kk - a.cb) < wvalid));
    };
    auto row_mask = [&](int r) -> uint32_t { return (SRC != 1 || ((r + o >= a.cb) & (r + o < a.H - a.cb))) ? 0xffffffffu : 0u; };
    const bool own_ok = col_ok(c, k);
    // ---- one-time LDS set-up: the byte masks, the empty ring
    for (int e = tid; e < 144; e += T) {
        const int lo = e / 12, hi = e - lo * 12 + 4;
        u32x4 w;
        for (int i = 0; i < 4; ++i) {
            uint32_t x = 0;
            for (int b = 0; b < 4; ++b) x |= (4 * i + b >= lo && 4 * i + b <= hi) ? (1u << (8 * b)) : 0u;
            w[i] = x;
        }
        lds_w128(m.tab_off + (uint32_t)e * 16u, w);
    }
#pragma unroll
    for (int s = 0; s < kMarchRing; ++s) lds_w32((uint32_t)s * RS4 + own, 0u);  // row -1: zero sum, zero count
    // ---- staging: which words of a quad's rows this thread brings in (the same for every quad).  Word idx of a kind is
    // (position w, row j) = (idx / 4, idx mod 4)
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void*)a.codes, 0, a.codes_bytes, kRsrcWord3);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.armsL8, 0, m.arms_bytes, kRsrcWord3);
    const int CWL = NC + 8;
    const uint32_t tb_off = (uint32_t)m.wpra * 16u;  // the (top, bottom) half of an arms buffer
    unsigned gc[kMarchSlots], lc[kMarchSlots], ga[kMarchSlots], la[kMarchSlots], adv_a[kMarchSlots];
    bool is_rng[kMarchSlots];
    constexpr bool kRange = SRC == 2;
    const int CWR = CWL + D - 1;
    const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc((void*)a.range, 0, kRange ? (unsigned)a.H * (unsigned)W * 4u : 0u, kRsrcWord3);
#pragma unroll
    for (int s = 0; s < kMarchSlots; ++s) {
        {
            const int idx = min(tid + s * T, 4 * m.wprc - 1);  // (a thread beyond the last word doubles it)
            const int w = idx >> 2, j = idx & 3;  // consecutive lanes, consecutive LDS words (a word's LDS index IS idx)
            const unsigned row0 = (unsigned)(j + o) * (unsigned)W + (unsigned)(o + c0 - 4);
            const unsigned word = w < CWL ? a.offCL + row0 + (unsigned)w : a.offCR + row0 + (unsigned)(a.d0 + (w - CWL));
            is_rng[s] = kRange && w >= CWL + CWR;  // SRC 2: the pixels' valid intervals ride behind the codes (another buffer)
            gc[s] = is_rng[s] ? (row0 + (unsigned)(w - CWL - CWR)) * 4u : word * 4u;
            lc[s] = m.code_off + (uint32_t)(w * 4 + j) * 4u;
        }
        {
            const int idx = min(tid + s * T, 4 * m.wpra - 1);
            const int w = idx >> 2, j = idx & 3;
            const bool left = w < NC;
            const unsigned word = left ? (unsigned)j * (unsigned)a.pitchL + (unsigned)(c0 + w)
                                       : m.armsR_word + (unsigned)j * (unsigned)a.pitchR + (unsigned)(a.padR + c0 + a.d0 + (w - NC));
            ga[s] = word * 4u;
            la[s] = m.arms_off + (uint32_t)(w * 4 + j) * 4u;
            adv_a[s] = (left ? (unsigned)a.pitchL : (unsigned)a.pitchR) * 16u;
        }
    }
    const unsigned adv_c = (unsigned)W * 16u;  // four rows of code words
    uint32_t sc[kMarchSlots], sa[kMarchSlots];
    auto issue = [&]() {  // the next quad's words into registers, offsets advanced
#pragma unroll
        for (int s = 0; s < kMarchSlots; ++s) {
            sc[s] = __builtin_amdgcn_raw_buffer_load_b32(rsC, (kRange && is_rng[s]) ? kOob : gc[s], 0, 0);
            if (kRange) sc[s] |= __builtin_amdgcn_raw_buffer_load_b32(rsG, is_rng[s] ? gc[s] : kOob, 0, 0);  // (the load out of range returns 0)
            gc[s] += adv_c;
            sa[s] = __builtin_amdgcn_raw_buffer_load_b32(rsA, ga[s], 0, 0);
            ga[s] += adv_a[s];
        }
    };
    auto land = [&](uint32_t cbuf, uint32_t abuf) {  // the registers into the buffers at these byte offsets
#pragma unroll
        for (int s = 0; s < kMarchSlots; ++s) {
            lds_w32(lc[s] + cbuf, sc[s]);
            const uint32_t at = la[s] + abuf;
            lds_w32(at, cb_lr16(sa[s]));
            lds_w32(at + tb_off, cb_tb16(sa[s]));
        }
    };
    // ---- per-thread addresses (the buffer of a quad and its rows are added as a scalar / as immediate offsets)
    const uint32_t xc = 4u + (uint32_t)cloc;
    const uint32_t code_l = m.code_off + xc * 16u;
    const uint32_t code_r = own_ok ? m.code_off + ((uint32_t)CWL + xc + (uint32_t)k) * 16u : code_l;  // (no cost: a code against itself)
    const uint32_t arms_l = m.arms_off + (uint32_t)cloc * 16u, arms_r = m.arms_off + (uint32_t)(NC + cloc + k) * 16u;  // (+ tb_off)
    const uint32_t cost_row = m.cost_off + (uint32_t)k * 16u;  // the 16 bytes of this disparity (+ row * rowb)
    const uint32_t rowb = (uint32_t)D * 16u;
    const uint32_t own_dst = cost_row + xc;
    const uint32_t rng_own = m.code_off + ((uint32_t)(CWL + CWR) + xc) * 16u;  // SRC 2: the valid intervals of this column's four rows
    const uint32_t tab_own = m.tab_off + (xc * 13u - 4u) * 16u;  // + (right - 12 * left) * 16
    const uint32_t kTabStrides = ((0u - 192u) & 0xffffu) | (16u << 16);  // (left, right) . (-192, 16)
    // halo cells: 8 columns x D disparities shared out over the workgroup (a second one only where a wavefront has any)
    uint32_t h_l[2], h_r[2], h_dst[2], h_g[2], h_k[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int h = min(tid + s * T, 8 * D - 1);
        const int hx = h / D, hk = h - hx * D;
        const int x = hx < 4 ? hx : NC + hx;  // columns 0 .. 3 and NC + 4 .. NC + 7 of the row of 16
        h_l[s] = m.code_off + (uint32_t)x * 16u;
        h_r[s] = col_ok(c0 - 4 + x, hk) ? m.code_off + (uint32_t)(CWL + x + hk) * 16u : h_l[s];
        h_dst[s] = m.cost_off + (uint32_t)hk * 16u + (uint32_t)x;
        h_g[s] = m.code_off + (uint32_t)(CWL + CWR + x) * 16u;
        h_k[s] = (uint32_t)hk;
    }
    const bool wave_h1 = m.nsh > 1 && __builtin_amdgcn_ballot_w64(tid + T < 8 * D) != 0;  // (uniform: a scalar branch)
    // ---- the march
    const unsigned row_bytes = (unsigned)W * (unsigned)D * 4u;
    const unsigned voff_st = live ? ((unsigned)(c + o) * (unsigned)D + (unsigned)k) * 4u : kOob;
    const size_t row_stride = (size_t)W * D;
    float* out_row = a.cv + (ptrdiff_t)(o - 4) * (ptrdiff_t)row_stride;  // output row of the quad's first leaving row (4n - 4)
    uint32_t acc = 0, nacc = 0;
    uint32_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;  // (top | bottom << 16) of the four rows that leave next
    bool ok_old[4] = {false, false, false, false}, ok_new[4] = {false, false, false, false};  // SRC 2: was the cell's own cost a number?
    issue();
    land(0u, 0u);
    issue();
    __syncthreads();
    const int NQ = (Hc + 4 + 3) / 4;  // rows 0 .. Hc + 3: the last four only leave
    int s0 = 0;                       // ring slot of the quad's first row: 4 * (n mod 3)
    for (int n = 0; n < NQ; ++n) {
        const int par = n & 1, r = 4 * n;
        const uint32_t cb_ = (uint32_t)par * m.code_buf, ab_ = (uint32_t)par * m.arms_buf, kb_ = (uint32_t)par * m.cost_buf;
        // a) the next quad's words land, the one after is requested; this quad's costs (bytes) and arms
        land(m.code_buf - cb_, m.arms_buf - ab_);
        issue();
        {
            const u32x4 cl = lds_r128(code_l + cb_), cr = lds_r128(code_r + cb_);
            const u32x4 hl = lds_r128(h_l[0] + cb_), hr = lds_r128(h_r[0] + cb_);
            u32x4 og{}, hg{};
            if (kRange) {
                og = lds_r128(rng_own + cb_);
                hg = lds_r128(h_g[0] + cb_);
            }
            uint32_t od = own_dst + kb_, hd = h_dst[0] + kb_;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t rm = row_mask(r + j);
                uint32_t oc = (uint32_t)__popc(cl[j] ^ cr[j]) & rm, hc = (uint32_t)__popc(hl[j] ^ hr[j]) & rm;
                if (kRange) {  // lo <= k < hi of the pixel's interval (lo | hi << 16)
                    ok_new[j] = (uint32_t)k - (og[j] & 0xffffu) < (og[j] >> 16) - (og[j] & 0xffffu);
                    oc = ok_new[j] ? oc : 0u;
                    hc = (h_k[0] - (hg[j] & 0xffffu) < (hg[j] >> 16) - (hg[j] & 0xffffu)) ? hc : 0u;
                }
                lds_w8(od, oc);
                lds_w8(hd, hc);
                od += rowb;
                hd += rowb;
            }
        }
        if (wave_h1) {
            const u32x4 hl = lds_r128(h_l[1] + cb_), hr = lds_r128(h_r[1] + cb_);
            u32x4 hg{};
            if (kRange) hg = lds_r128(h_g[1] + cb_);
            uint32_t hd = h_dst[1] + kb_;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t hc = (uint32_t)__popc(hl[j] ^ hr[j]) & row_mask(r + j);
                if (kRange) hc = (h_k[1] - (hg[j] & 0xffffu) < (hg[j] >> 16) - (hg[j] & 0xffffu)) ? hc : 0u;
                lds_w8(hd, hc);
                hd += rowb;
            }
        }
        uint32_t lr[4], tb[4];
        {
            // the four rows' (left, right) and (top, bottom) pairs of the two pixels
            const u32x4 ll = lds_r128(arms_l + ab_), lt = lds_r128(arms_l + ab_ + tb_off);
            const u32x4 rl = lds_r128(arms_r + ab_), rt = lds_r128(arms_r + ab_ + tb_off);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                lr[j] = r + j < Hc ? cb_pk_min(ll[j], rl[j]) : 0u;  // a row past the image has no arms (what was staged for it is not arms)
                tb[j] = cb_pk_min(lt[j], rt[j]);
            }
        }
        __syncthreads();
        // c) segment sums, prefixes, and the four rows that leave.  A row past the image (the last quads) runs like any other
        // on zero codes and no arms: what it leaves in the ring is never read (a ring of 12: row r + 3 takes the slot of row
        // r - 9, which the first leaving row may still need - hence its reads come first).
        uint32_t ent[4];
        const uint32_t crow = cost_row + kb_;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            u32x4 w[2], mk[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int j = 2 * h + i;
                w[i] = lds_r128(crow + (uint32_t)j * rowb);
                mk[i] = lds_r128((uint32_t)sdot2(lr[j], kTabStrides, (int32_t)tab_own));  // tab_own + (right - 12 * left) * 16
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int j = 2 * h + i;
                uint32_t e = __builtin_amdgcn_udot4(w[i].x, mk[i].x, 0u, false);
                e = __builtin_amdgcn_udot4(w[i].y, mk[i].y, e, false);
                e = __builtin_amdgcn_udot4(w[i].z, mk[i].z, e, false);
                e = __builtin_amdgcn_udot4(w[i].w, mk[i].w, e, false);
                acc += e;
                nacc = __builtin_amdgcn_sad_u16(lr[j], 0u, nacc) + 1u;
                ent[j] = __builtin_amdgcn_perm(nacc, acc, 0x05040100u);  // (S3 & 0xffff) | N << 16
            }
        }
        struct pair_hl { uint32_t hi, lo; };
        const int e0 = s0 >= 4 ? s0 - 4 : 8;  // slot of row r - 4
        const uint32_t ring_bytes = (uint32_t)kMarchRing * RS4, neg_rs4 = (0u - RS4) & 0xffffu;  // (RS4 < 32768: a ring slot of at most 8 K cells)
        auto fetch = [&](uint32_t tbv, int j) {
            // byte addresses: own + ((e0 + j + bot) mod 12) * RS4 and own + ((e0 + j - 1 - top) mod 12) * RS4 - one dot product of the
            // packed (top, bottom) with (0, RS4) / (-RS4, 0) each, then the wrap (x - 12 RS4 is huge when x did not reach it)
            const uint32_t b0 = (uint32_t)(e0 + j) * RS4 + own;
            uint32_t u = udot2(tbv, RS4 << 16, b0), v = (uint32_t)sdot2(tbv, neg_rs4, (int32_t)(b0 + (uint32_t)(kMarchRing - 1) * RS4));
            u = min(u, u - ring_bytes);
            v = min(v, v - ring_bytes);
            return pair_hl{lds_r32(u), lds_r32(v)};
        };
        const uint32_t ring_w = (uint32_t)s0 * RS4 + own;
        lds_w32(ring_w, ent[0]);
        lds_w32(ring_w + RS4, ent[1]);
        lds_w32(ring_w + 2u * RS4, ent[2]);
        const pair_hl p0 = fetch(t0, 0);
        lds_w32(ring_w + 3u * RS4, ent[3]);
        const pair_hl p1 = fetch(t1, 1), p2 = fetch(t2, 2), p3 = fetch(t3, 3);
        auto emit = [&](const pair_hl& p, int j) {
            const uint32_t d = __builtin_bit_cast(uint32_t, __builtin_bit_cast(cb_us2, p.hi) - __builtin_bit_cast(cb_us2, p.lo));
            const float quot = small_int_div((float)(d & 0xffffu), (float)(d >> 16));
            float res;
            if (SRC == 3) res = inside ? quot : c_nan();
            else if (SRC == 2) res = ok_old[j] ? (inside ? quot : 0.f) : c_nan();
            else res = (own_ok && row_mask(r - 4 + j)) ? (inside ? quot : 0.f) : c_nan();
            const bool row_in = (r - 4 + j >= 0) & (r - 4 + j < Hc);  // (uniform: a row that does not exist gets an empty descriptor)
            const __amdgpu_buffer_rsrc_t rsO =
                __builtin_amdgcn_make_buffer_rsrc((void*)(out_row + (size_t)j * row_stride), 0, row_in ? row_bytes : 0u, kRsrcWord3);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(res), rsO, voff_st, 0, 0);
        };
        emit(p0, 0); emit(p1, 1); emit(p2, 2); emit(p3, 3);
        t0 = tb[0]; t1 = tb[1]; t2 = tb[2]; t3 = tb[3];
        if (kRange) {
#pragma unroll
            for (int j = 0; j < 4; ++j) ok_old[j] = ok_new[j];
        }
        s0 = s0 == 8 ? 0 : s0 + 4;
        out_row += (size_t)4 * row_stride;
    }
}

// ---- four disparities per thread (subpix 1, cbca_distance <= 5) -----------------------------------------------------------
// The scans above move 4 bytes per lane per memory instruction and are bound by the texture addresser (~30 cycles per vector
// memory instruction per CU whatever its width: 4 - 5 of them per 64 cells).  Here a thread owns FOUR consecutive disparities of
// its row / column: 16-byte loads and stores, the left image's arms loaded once for the four, the right image's four arms as
// two 16-byte loads from rows padded by 4 pixels.  Per cell the arithmetic and its order are unchanged.  Both passes work IN
// PLACE on the volume (a scan's output cell was read by the same thread A steps earlier), so no second volume is needed, and
// pass V learns which input costs were NaN from 4 bits per cell group written by pass H instead of re-reading the costs.
static constexpr int kBlock4 = 128;
static constexpr int kRing4 = 16;  // >= 3A+3 for A <= 4
static constexpr int kPF4 = 8;     // columns / rows of read-ahead

typedef unsigned int cb_u4 __attribute__((ext_vector_type(4)));
typedef unsigned int cb_u2 __attribute__((ext_vector_type(2)));
constexpr unsigned kCbRsrc3 = 0x00020000;  // raw buffer descriptor word 3 (as k_sgmfam.hip): 4-byte alignment is enough for 16-byte accesses
constexpr unsigned kCbOob = 0x80000000u;   // an offset past every buffer: the store is dropped

struct cb_arms4 { uint32_t lr[4], tb[4]; };
// 4 consecutive pixels of an arms row = 32 bytes at byte offset `off` of the buffer
__device__ __forceinline__ cb_arms4 cb_load4(__amdgpu_buffer_rsrc_t rs, unsigned off) {
    const cb_u4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0), b = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16, 0, 0);
    return {{a.x, a.z, b.x, b.z}, {a.y, a.w, b.y, b.w}};
}
__device__ __forceinline__ float4 cb_loadf4(__amdgpu_buffer_rsrc_t rs, unsigned off) {
    const cb_u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
typedef unsigned int cb_u3 __attribute__((ext_vector_type(3)));
// stores the first nk of four values (nk = 4, or tailn = D % 4 on a pixel's last group): one 16-byte store for the full groups
// and ONE narrower store (tailn is the same for every pixel) for the last ones; lanes a store does not concern give it an
// out-of-range offset - no branch on lane-varying data around a memory instruction
__device__ __forceinline__ void cb_storef4(__amdgpu_buffer_rsrc_t rs, unsigned off, int nk, int tailn, bool live, const float (&v)[4]) {
    cb_u4 t;
    t.x = __float_as_uint(v[0]); t.y = __float_as_uint(v[1]); t.z = __float_as_uint(v[2]); t.w = __float_as_uint(v[3]);
    __builtin_amdgcn_raw_buffer_store_b128(t, rs, (live && nk == 4) ? off : kCbOob, 0, 0);
    const unsigned toff = (live && nk < 4) ? off : kCbOob;
    if (tailn == 1) {  // uniform
        __builtin_amdgcn_raw_buffer_store_b32(t.x, rs, toff, 0, 0);
    } else if (tailn == 2) {
        cb_u2 h; h.x = t.x; h.y = t.y;
        __builtin_amdgcn_raw_buffer_store_b64(h, rs, toff, 0, 0);
    } else if (tailn == 3) {
        cb_u3 h; h.x = t.x; h.y = t.y; h.z = t.z;
        __builtin_amdgcn_raw_buffer_store_b96(h, rs, toff, 0, 0);
    }
}
__device__ __forceinline__ float comp(const float4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
__device__ __forceinline__ unsigned cb_span(size_t bytes) { return bytes > 0xffffffffull ? 0xffffffffu : (unsigned)bytes; }

__global__ __launch_bounds__(kBlock4) void cbca_h4_kernel(cbca_args a) {
    __shared__ float ring[kRing4 * kBlock4 * 4];  // [slot][disparity of the four][thread]: lanes on consecutive banks
    const int t = blockIdx.x * kBlock4 + threadIdx.x;
    const int total = a.Hc * a.G;
    const bool live = t < total;
    const int tt = live ? t : total - 1;
    const int r = tt / a.G, g = tt - r * a.G, k0 = 4 * g;
    const int nk = a.D - k0 < 4 ? a.D - k0 : 4;  // disparities this thread owns (the last group of a pixel may own fewer)
    const int tailn = a.D & 3;
    const int dq = a.d0 + k0;
    constexpr int mask = kRing4 - 1;
    const int A = a.A, Wc = a.Wc, D = a.D;
    float* my = ring + threadIdx.x;
    for (int s = 0; s < kRing4 * 4; ++s) my[s * kBlock4] = 0.f;  // S1 of columns < 0
    // buffer descriptors: the volume from the block's first row on (a block spans a few rows: 32-bit offsets), the arms images
    const int r0 = (blockIdx.x * kBlock4) / a.G;
    const size_t vol_bytes = (size_t)a.H * a.W * D * 4 + 256;
    const size_t base_el = ((size_t)(r0 + a.o) * a.W + a.o) * D;
    // (the descriptor ends behind the block's LAST row: kCbOob, the offset of a lane that must not store, has to lie beyond it.  Until
    //  round 6 it reached to the volume's end or 4 GB - at 4096^2 x 257 the tail store of every lane with four disparities, meant to
    //  be dropped, landed 2 GB behind the block's first row; inside the aggregated area its owner wrote the cell again later, in
    //  the image's border rows nobody did: one cell of the NaN border became a number - found when four processes shared the GPU,
    //  the second volume could not be had and this route ran at full size for the first time)
    const int r_last = min(a.Hc - 1, (int)((blockIdx.x * kBlock4 + kBlock4 - 1) / a.G));
    const size_t blk_bytes = (size_t)(r_last - r0 + 1) * a.W * D * 4 + 256;
    const size_t left_bytes = vol_bytes - base_el * 4;
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)(a.cv + base_el), 0, cb_span(blk_bytes < left_bytes ? blk_bytes : left_bytes), kCbRsrc3);
    const __amdgpu_buffer_rsrc_t rsL = __builtin_amdgcn_make_buffer_rsrc((void*)a.armsL, 0, cb_span((size_t)a.Hc * Wc * 8), kCbRsrc3);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)a.armsRpad, 0, cb_span((size_t)a.Hc * (Wc + 8) * 8), kCbRsrc3);
    unsigned ov = (unsigned)(((size_t)(r - r0) * a.W * D + k0) * 4);  // cost of column c
    unsigned oe = ov;                                                   // segment sum of column ce = c - A (in place)
    const unsigned step_b = (unsigned)D * 4;
    const unsigned oL = (unsigned)((size_t)r * Wc * 8);
    const unsigned oR = (unsigned)(((size_t)r * (Wc + 8) + 4) * 8);  // pixel q at oR + 8 q
    uint32_t* pn = a.nanbits + ((size_t)r * ((Wc + 7) / 8)) * a.G + g;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    uint32_t nanword = 0;
    auto prefix = [&](const float4& v, int c) {
        uint32_t bits = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float x = comp(v, i);
            acc[i] = (x == x) ? acc[i] + x : acc[i];  // NaN is skipped, the running sum carries on
            bits |= (fabsf(x) < c_inf()) ? 0u : (1u << i);  // what pass V needs of the input: is `in * 0` a NaN (NaN or infinite cost)
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) my[((c & mask) * 4 + i) * kBlock4] = acc[i];
        nanword |= bits << (4 * (c & 7));
        if ((c & 7) == 7 || c == Wc - 1) {
            if (live) pn[(size_t)(c >> 3) * a.G] = nanword;
            nanword = 0;
        }
    };
    auto right_off = [&](int ce) {  // first of the four right-image pixels of column ce, clamped into the padded row
        const int q0 = ce + dq;
        return oR + (unsigned)((q0 < -4 ? -4 : (q0 > Wc ? Wc : q0)) * 8);
    };
    auto left_arms = [&](int ce) { return __builtin_amdgcn_raw_buffer_load_b64(rsL, oL + (unsigned)ce * 8, 0, 0); };
    auto emit = [&](cb_u2 l, const cb_arms4& rr, int ce) {
        float e[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = ce + dq + i;
            const bool inside = (q >= 0) & (q <= Wc - 1);
            const uint32_t lr = cb_pk_min(l.x, rr.lr[i]);
            const int left = (int)(lr & 0xffffu), right = (int)(lr >> 16);
            const float hi_v = my[(((ce + right) & mask) * 4 + i) * kBlock4];
            const float lo_v = my[(((ce - left - 1) & mask) * 4 + i) * kBlock4];
            e[i] = inside ? hi_v - lo_v : 0.f;
        }
        cb_storef4(rsV, oe, nk, tailn, live, e);
        oe += step_b;
    };
    int c = 0;
    for (; c < A; ++c) {  // warm-up: columns whose segment cannot be closed yet
        prefix(cb_loadf4(rsV, ov), c);
        ov += step_b;
    }
    // steady state: a register ring of kPF4 columns in flight (a thread is nearly alone on its SIMD: nothing else hides HBM
    // latency); slot j holds column c + j, is consumed and refilled with column c + j + kPF4 (clamped: surplus loads re-read
    // the last column)
    float4 vb[kPF4];
    cb_u2 lb[kPF4];
    cb_arms4 rb[kPF4];
    auto fill = [&](int j, int col) {
        const int cc = min(col, Wc - 1);
        vb[j] = cb_loadf4(rsV, ov + (unsigned)(cc - c) * step_b);
        const int ce = min(col - A, Wc - 1);
        lb[j] = left_arms(ce);
        rb[j] = cb_load4(rsR, right_off(ce));
    };
#pragma unroll
    for (int j = 0; j < kPF4; ++j) fill(j, c + j);
    for (; c + kPF4 <= Wc; c += kPF4) {
#pragma unroll
        for (int j = 0; j < kPF4; ++j) {
            const float4 v = vb[j];
            const cb_u2 l = lb[j];
            const cb_arms4 rr = rb[j];
            fill(j, c + j + kPF4);
            prefix(v, c + j);
            emit(l, rr, c + j - A);
        }
        ov += (unsigned)kPF4 * step_b;
    }
#pragma unroll
    for (int j = 0; j < kPF4 - 1; ++j) {  // leftover columns: their data is already in the ring
        if (c + j < Wc) {
            prefix(vb[j], c + j);
            emit(lb[j], rb[j], c + j - A);
        }
    }
    c = Wc;
    for (; c < Wc + A; ++c)  // drain
        emit(left_arms(c - A), cb_load4(rsR, right_off(c - A)), c - A);
}

__global__ __launch_bounds__(kBlock4) void cbca_v4_kernel(cbca_args a) {
    __shared__ float ring[2 * kRing4 * kBlock4 * 4];  // [2][slot][disparity of the four][thread]: column prefix sums; packed (N, top, bot)
    const int t = blockIdx.x * kBlock4 + threadIdx.x;
    const int total = a.Wc * a.G;
    const bool live = t < total;
    const int tt = live ? t : total - 1;
    const int c = tt / a.G, g = tt - c * a.G, k0 = 4 * g;
    const int nk = a.D - k0 < 4 ? a.D - k0 : 4;
    const int tailn = a.D & 3;
    const int q0 = c + a.d0 + k0;
    constexpr int mask = kRing4 - 1;
    const int A = a.A, Hc = a.Hc, Wc = a.Wc;
    bool inside[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) inside[i] = (q0 + i >= 0) & (q0 + i <= Wc - 1);  // the same right columns for every row
    float* s3 = ring + threadIdx.x;
    uint32_t* info = reinterpret_cast<uint32_t*>(ring) + kRing4 * kBlock4 * 4 + threadIdx.x;
    for (int s = 0; s < kRing4 * 4; ++s) {  // row -1: zero sums, zero counts
        s3[s * kBlock4] = 0.f;
        info[s * kBlock4] = 0u;
    }
    // every thread of the block is on the same image row: the row is the (wave-uniform) descriptor, the thread's offset inside it
    // never changes
    const size_t row_bytes = (size_t)a.W * a.D * 4;
    const unsigned ocol = (unsigned)(((size_t)(c + a.o) * a.D + k0) * 4);
    auto row_rsrc = [&](int r) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)((char*)a.cv + (size_t)(r + a.o) * row_bytes), 0, cb_span(row_bytes + 256), kCbRsrc3);
    };
    const __amdgpu_buffer_rsrc_t rsL = __builtin_amdgcn_make_buffer_rsrc((void*)a.armsL, 0, cb_span((size_t)Hc * Wc * 8), kCbRsrc3);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)a.armsRpad, 0, cb_span((size_t)Hc * (Wc + 8) * 8), kCbRsrc3);
    const unsigned oL = (unsigned)c * 8, oR = (unsigned)(((q0 < -4 ? -4 : (q0 > Wc ? Wc : q0)) + 4) * 8);
    const unsigned strideL = (unsigned)Wc * 8, strideR = (unsigned)(Wc + 8) * 8;  // bytes per arms row
    const uint32_t* pn = a.nanbits + (size_t)(c >> 3) * a.G + g;
    const size_t strideN = (size_t)((Wc + 7) / 8) * a.G;
    const int nshift = 4 * (c & 7);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    uint32_t nacc[4] = {0u, 0u, 0u, 0u};
    auto prefix = [&](const float4& e4, cb_u2 l, const cb_arms4& rr, int r) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float e = comp(e4, i);
            acc[i] = (r == 0) ? e : acc[i] + e;
            const uint32_t lr = cb_pk_min(l.x, rr.lr[i]), tb = cb_pk_min(l.y, rr.tb[i]);
            // ring word: running count N(r) of n_h = left + right in bits 0..19, top in bits 20..25, bot in bits 26..31
            // (63, 63 = this cell is outside the right image, n_h = 0)
            nacc[i] += inside[i] ? (lr & 0xffffu) + (lr >> 16) : 0u;
            w[i] = nacc[i] | (inside[i] ? (((tb & 0xffffu) << 20) | ((tb >> 16) << 26)) : ((63u << 20) | (63u << 26)));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s3[((r & mask) * 4 + i) * kBlock4] = acc[i];
            info[((r & mask) * 4 + i) * kBlock4] = w[i];
        }
    };
    auto emit = [&](uint32_t nanword, int re) {
        float out[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t w = info[((re & mask) * 4 + i) * kBlock4];
            const int top = (w >> 20) & 63, bot = w >> 26;
            const bool cell = top != 63;
            const int hi_i = cell ? re + bot : re, lo_i = cell ? re - top - 1 : re;
            const float step = s3[((hi_i & mask) * 4 + i) * kBlock4] - s3[((lo_i & mask) * 4 + i) * kBlock4];
            const uint32_t n = (info[((hi_i & mask) * 4 + i) * kBlock4] & 0xfffffu) - (info[((lo_i & mask) * 4 + i) * kBlock4] & 0xfffffu) +
                               (uint32_t)(top + bot);
            const float step4 = cell ? step : 0.f;
            const float sum4 = (cell ? (float)n : 0.f) + 1.f;  // small exact integers: any order
            const float in0 = ((nanword >> (nshift + i)) & 1u) ? c_nan() : 0.f;  // in * 0: NaN stays NaN (cbca.py:145-146,168-171)
            out[i] = (in0 + step4) / sum4;
        }
        cb_storef4(row_rsrc(re), ocol, nk, tailn, live, out);  // over the E_h read A steps ago
    };
    auto fetch = [&](int row, float4& e, cb_u2& l, cb_arms4& rr) {
        e = cb_loadf4(row_rsrc(row), ocol);  // E_h of the row (pass H left it in place of the costs)
        l = __builtin_amdgcn_raw_buffer_load_b64(rsL, oL + (unsigned)row * strideL, 0, 0);
        rr = cb_load4(rsR, oR + (unsigned)row * strideR);
    };
    int r = 0;
    for (; r < A; ++r) {  // warm-up
        float4 e;
        cb_u2 l;
        cb_arms4 rr;
        fetch(r, e, l, rr);
        prefix(e, l, rr, r);
    }
    // steady state: a register ring of kPF4 rows in flight; the NaN word of row re = r - A travels with row r
    float4 eb[kPF4];
    cb_u2 lb[kPF4];
    cb_arms4 rb[kPF4];
    uint32_t nb[kPF4];
    auto fill = [&](int j, int row) {
        fetch(min(row, Hc - 1), eb[j], lb[j], rb[j]);  // clamped: surplus loads re-read the last row
        nb[j] = pn[(size_t)(min(row, Hc - 1 + A) - A) * strideN];  // row - A >= 0 here
    };
#pragma unroll
    for (int j = 0; j < kPF4; ++j) fill(j, r + j);
    for (; r + kPF4 <= Hc; r += kPF4) {
#pragma unroll
        for (int j = 0; j < kPF4; ++j) {
            const float4 e = eb[j];
            const cb_u2 l = lb[j];
            const cb_arms4 rr = rb[j];
            const uint32_t nw = nb[j];
            fill(j, r + j + kPF4);
            prefix(e, l, rr, r + j);
            emit(nw, r + j - A);
        }
    }
    {
        const int r0 = r;
#pragma unroll
        for (int j = 0; j < kPF4 - 1; ++j) {  // leftover rows: their data is already in the ring
            if (r0 + j < Hc) {
                prefix(eb[j], lb[j], rb[j], r0 + j);
                emit(nb[j], r0 + j - A);
                r = r0 + j + 1;
            }
        }
    }
    for (; r < Hc + A; ++r) emit(pn[(size_t)(r - A) * strideN], r - A);  // drain
}

static int cbca_ring_slots(int A) {  // 2A+2 live columns + A+1 still-zero slots that stand for the columns before the first
    int ring = 4;
    while (ring < 3 * A + 3) ring <<= 1;
    return ring;
}

// rows per workgroup of the whole-row pass H: R*D threads rounded up to whole wavefronts - the fewer idle lanes the better
// (D = 129: 2 rows are 5 wavefronts, the fifth with 2 lanes; 4 rows are 9 with 4) -, within kRowsT threads and 150 KB of LDS
// (ring + output stage); 0: the kernel does not fit (very long arms with many disparities)
static int cbca_rows_per_block(const pmx_ctx* ctx, int D, int ring) {
    const char* er = pmx_opt(ctx, "CBCA_ROWS");
    int best = 0;
    double best_waste = 1e9;
    for (int R = 1; R <= 8; ++R) {
        const int T = ((R * D + 63) / 64) * 64;
        if (T > kRowsT) break;
        if (((size_t)ring * T + (size_t)R * kStage * D) * sizeof(float) > (size_t)150 * 1024) break;
        if (er && atoi(er) == R) return R;
        const double waste = (double)T / (R * D);
        if (waste < best_waste - 0.02) { best_waste = waste; best = R; }
    }
    return best;
}

// can pass H compute the census costs itself (no float volume)?  One code word per pixel, whole-row kernel applicable.
bool pmx_cbca_can_fuse_census(const pmx_ctx* ctx, const pmx_cv* cv, int offset, int distance) {
    const char* ef = pmx_opt(ctx, "CBCA_FAST");
    const char* eu = pmx_opt(ctx, "CBCA_FUSE");
    if ((ef && atoi(ef) != 1) || (eu && eu[0] == '0')) return false;
    const int A = distance - 1 > 1 ? distance - 1 : 1;
    const int Hc = cv->H - 2 * offset, Wc = cv->W - 2 * offset;
    const int pitchR = (cv->d0 < 0 ? -cv->d0 : 0) + 4 + Wc + (cv->d0 + cv->D - 1 > 0 ? cv->d0 + cv->D - 1 : 0) + 8;
    // (a crop wider than the census border leaves cells with real costs outside the aggregated area: they need the volume)
    return cv->repr == PMX_REPR_CENSUS_DEFERRED && cv->subpix == 1 && cv->win * cv->win <= 32 && offset <= cv->win / 2 && cbca_ring_slots(A) <= 64 &&
           cbca_rows_per_block(ctx, cv->D, cbca_ring_slots(A)) > 0 && Wc >= 2 * A + 8 && Hc >= 2 * A + 8 && cv->codes_bytes < 0x7fffffffu &&
           (size_t)Hc * pitchR * 4 < 0x7fffffffu && (size_t)Hc * (Wc + 4) * 4 < 0x7fffffffu && (size_t)8 * cv->W * cv->D * 4 < 0x7fffffffu;
}

int pmx_launch_cbca(pmx_ctx* ctx, pmx_cv* cv, int offset, float intensity, int distance, bool census_src) {
    const int H = cv->H, W = cv->W, o = offset;
    const int Hc = H - 2 * o, Wc = W - 2 * o;
    if (Hc <= 0 || Wc <= 0) return PMX_OK;  // (one cropped column is a volume like any other: its columns still aggregate)
    cbca_args a{};
    a.A = distance - 1 > 1 ? distance - 1 : 1;
    // kernel choice.  Default: the phase-split scans when the scanned dimension is long enough, else the generic ones.  The
    // four-disparities-per-thread kernels work IN PLACE (no second volume: 51.6 GB less at 10000^2 x 129) but measured slower
    // (2048^2 x 129: 2.15 + 2.97 ms against 1.50 + 1.96: a quarter of the waves, nothing left to hide latency), so they run when
    // the second volume cannot be had, or on request.  PMX_CBCA_FAST (test hook): 0 generic, 2 phase-split, 4 in-place.
    const char* ef = pmx_opt(ctx, "CBCA_FAST");
    const int want = ef ? atoi(ef) : 1;
    const bool long_scans = Wc >= 2 * a.A + 8 && Hc >= 2 * a.A + 8;
    // (in-place kernels: a block's rows of the volume and the arms images lie behind descriptors that must end below the 2^31 marker of
    //  a lane that does not store)
    const bool four_ok = cv->subpix == 1 && 3 * a.A + 3 <= kRing4 && long_scans &&
                         ((size_t)kBlock4 / ((cv->D + 3) / 4) + 2) * W * cv->D * 4 + 256 < 0x80000000ull && (size_t)Hc * (Wc + 8) * 8 < 0xfffffff0ull;
    bool four = want == 4 && four_ok;
    if (!four && four_ok && want == 1 && ctx->scratch_bytes < cv->cells() * sizeof(float) + 256) {
        void* probe = nullptr;  // is there room for the second volume?
        if (pmx_pool_alloc(ctx, &probe, cv->cells() * sizeof(float) + 256) == hipSuccess) {
            pmx_pool_free(ctx, ctx->scratch);
            ctx->scratch = (float*)probe;
            ctx->scratch_bytes = cv->cells() * sizeof(float) + 256;
        } else {
            (void)hipGetLastError();
            four = true;
        }
    }
    const int G = (cv->D + 3) / 4;
    // whole-row pass H (the default for long scans): byte arms, row-major with pads
    const int dq_max = cv->d0 + (cv->D - 1) / cv->subpix;
    const int padR = (cv->d0 < 0 ? -cv->d0 : 0) + 4;
    const int pitchL = Wc + 4, pitchR = padR + Wc + (dq_max > 0 ? dq_max : 0) + 8;
    const size_t bL8 = (size_t)Hc * pitchL * 4, bR8 = (size_t)Hc * pitchR * 4;
    const int ring = cbca_ring_slots(a.A);
    const int rows = ring <= 64 ? cbca_rows_per_block(ctx, cv->D, ring) : 0;
    const bool rows_ok = !four && want == 1 && long_scans && rows > 0 && (size_t)cv->subpix * bR8 < 0x7fffffffu && bL8 < 0x7fffffffu &&
                         (size_t)8 * W * cv->D * 4 < 0x7fffffffu;
    if (census_src && !rows_ok) {
        pmx_set_error("pmx_cbca: internal: census source without the whole-row kernel");
        return PMX_ERR_STATE;
    }
    // census source, short arms: costs and both scans in one marching kernel, no E_h volume (PMX_CBCA_MARCH=0: test hook)
    cbca_march m{};
    size_t march_lds = 0;
    bool march = false;
    {
        const char* em = pmx_opt(ctx, "CBCA_MARCH");
        const bool shape_ok = census_src && rows_ok && a.A <= 4 && cv->subpix == 1 && cv->D <= 256 &&
                              (size_t)Hc * 9 * 32 < ((size_t)1 << 24) && bL8 + bR8 < 0xfffffff0u;
        // columns per workgroup: as many (up to 8: a disparity's costs of a row are 16 bytes with the 8 halo columns) as leave
        // room for two workgroups per CU, else as fit one
        auto layout = [&](int NC) {
            m.NC = NC;
            m.T = ((NC * cv->D + 63) / 64) * 64;
            m.wprc = 2 * (NC + 8) + cv->D - 1 + (cv->has_range ? NC + 8 : 0);  // (+ the valid intervals of the left columns)
            m.wpra = 2 * NC + cv->D - 1;
            m.nsh = (8 * cv->D + m.T - 1) / m.T;
            m.cost_buf = (unsigned)cv->D * 64u;
            m.code_buf = (unsigned)m.wprc * 16u;
            m.arms_buf = (unsigned)m.wpra * 32u;
            m.ring_stride = ((unsigned)(NC * cv->D) * 4u + 127u) & ~127u;
            m.cost_off = (unsigned)kMarchRing * m.ring_stride;
            m.tab_off = m.cost_off + 2u * m.cost_buf;
            m.code_off = m.tab_off + (unsigned)kMarchTab;
            m.arms_off = m.code_off + 2u * m.code_buf;
            march_lds = (size_t)m.arms_off + 2u * m.arms_buf;
            return m.T <= 1024 && m.nsh <= 2 && 4 * m.wprc <= kMarchSlots * m.T && 4 * m.wpra <= kMarchSlots * m.T;
        };
        int pick = 0;
        if (shape_ok) {
            for (int NC = 8; NC >= 4 && !pick; --NC)
                if (layout(NC) && march_lds <= (size_t)80 * 1024) pick = NC;
            for (int NC = 8; NC >= 4 && !pick; --NC)
                if (layout(NC) && march_lds <= (size_t)160 * 1024) pick = NC;
        }
        if (pick) {
            layout(pick);
            m.arms_bytes = (unsigned)(bL8 + bR8);
            m.armsR_word = (unsigned)(bL8 / 4);
            march = !(em && em[0] == '0');  // (faster than passes H + V at every size tried: cones 0.11 against 0.19 ms, 10000^2 x 129 24.7 against 50.5)
        }
    }
    const size_t wide_bytes = rows_ok ? bL8 + (size_t)cv->subpix * bR8 : 0;
    // small scratch: 2 float images + arms of left and of every shifted right image (+ padded right arms and NaN bits)
    const size_t img_bytes = (size_t)H * W * sizeof(float);
    const size_t arm_bytes = (size_t)Hc * Wc * 4 * sizeof(int16_t);
    const size_t pad_bytes = four ? (size_t)Hc * (Wc + 8) * 4 * sizeof(int16_t) : 0;
    const size_t nan_bytes = four ? (size_t)Hc * ((Wc + 7) / 8) * G * sizeof(uint32_t) : 0;
    int rc = pmx_need_small(ctx, 2 * img_bytes + arm_bytes * (1 + cv->subpix) + pad_bytes + nan_bytes + wide_bytes + 64);
    if (rc) return rc;
    if (!four && !march) {
        rc = pmx_need_scratch(ctx, cv->cells() * sizeof(float) + 256);
        if (rc) return rc;
    }
    char* base = (char*)ctx->small;
    float* tmp = (float*)base;
    a.cv = cv->data;
    a.eh = four ? nullptr : ctx->scratch;
    a.armsL = (int16_t*)(base + 2 * img_bytes);
    for (int k = 0; k < PMX_MAX_SUBPIX; ++k) a.armsR[k] = nullptr;
    int16_t* padded = (int16_t*)(base + 2 * img_bytes + arm_bytes * (1 + cv->subpix));
    a.armsRpad = padded;
    a.nanbits = (uint32_t*)((char*)padded + pad_bytes);
    a.G = G;
    char* wbase = (char*)(((uintptr_t)((char*)a.nanbits + nan_bytes) + 15) & ~(uintptr_t)15);
    a.armsL8 = (uint32_t*)wbase;
    a.armsR8 = (uint32_t*)(wbase + bL8);
    a.pitchL = pitchL; a.pitchR = pitchR; a.padR = padR;
    a.bytesL8 = (unsigned)bL8; a.bytesR8 = (unsigned)((size_t)cv->subpix * bR8); a.phase_words = (unsigned)(bR8 / 4);
    a.R = rows > 0 ? rows : 1;
    a.codes = cv->codes; a.codes_bytes = (unsigned)cv->codes_bytes;
    a.offCL = cv->codeL ? (unsigned)(cv->codeL - cv->codes) : 0u; a.offCR = cv->codeR ? (unsigned)(cv->codeR - cv->codes) : 0u;
    a.range = (census_src && cv->has_range) ? cv->range : nullptr;
    a.cb = cv->win / 2;
    // pass V through buffers (the choice is made here because it decides which form of the arms is needed)
    // (worth it when there are more wavefronts than the pointer kernel's 3 per SIMD can hold: 2048^2 x 129 has 4 per SIMD and
    // runs 1.71 against 1.82 ms with pointers, 10000^2 x 129 has 20 and runs 31 against 35 ms with buffers)
    const char* ev = pmx_opt(ctx, "CBCA_VBUF");  // 0: the phase-split kernel with pointers (test hook)
    const bool vbuf = rows_ok && (ev ? ev[0] != '0' : (size_t)Wc * cv->D >= (size_t)6144 * 64);
    const bool want16 = !(march || vbuf);  // the short4 arms: only the kernels that read them through pointers
    {
        pmx_stage_scope t(ctx, PMX_STAGE_CBCA_ARMS);
        // the pads are zero arms (the branch-free kernel writes them itself when it writes the packed rows)
        if (rows_ok && (want16 || !arms_flat(ctx, distance))) PMX_HIP(hipMemsetAsync(wbase, 0, wide_bytes, ctx->stream));
        rc = rows_ok ? build_arms(ctx, 0, o, intensity, distance, tmp, (int16_t*)a.armsL, 0, (uint32_t*)a.armsL8, pitchL, 0, want16)
                     : build_arms(ctx, 0, o, intensity, distance, tmp, (int16_t*)a.armsL);
        if (rc) return rc;
        if (four) {
            PMX_HIP(hipMemsetAsync(padded, 0, pad_bytes, ctx->stream));
            rc = build_arms(ctx, 1, o, intensity, distance, tmp, padded, 4);
            if (rc) return rc;
        } else {
            for (int k = 0; k < cv->subpix; ++k) {
                int16_t* dst = (int16_t*)(base + 2 * img_bytes + arm_bytes * (1 + k));
                a.armsR[k] = dst;
                rc = rows_ok ? build_arms(ctx, k + 1, o, intensity, distance, tmp, dst, 0, (uint32_t*)a.armsR8 + (size_t)k * a.phase_words, pitchR, padR, want16)
                             : build_arms(ctx, k + 1, o, intensity, distance, tmp, dst);
                if (rc) return rc;
            }
        }
    }
    a.H = H; a.W = W; a.D = cv->D; a.d0 = cv->d0; a.subpix = cv->subpix; a.o = o; a.Hc = Hc; a.Wc = Wc;
    a.ring = ring;
    if (four) {
        {
            pmx_stage_scope t(ctx, PMX_STAGE_CBCA_H);
            const int total = Hc * G;
            hipLaunchKernelGGL(cbca_h4_kernel, dim3((total + kBlock4 - 1) / kBlock4), dim3(kBlock4), 0, ctx->stream, a);
        }
        {
            pmx_stage_scope t(ctx, PMX_STAGE_CBCA_V);
            const int total = Wc * G;
            hipLaunchKernelGGL(cbca_v4_kernel, dim3((total + kBlock4 - 1) / kBlock4), dim3(kBlock4), 0, ctx->stream, a);
        }
        PMX_HIP(hipGetLastError());
        return PMX_OK;
    }
    // arms of 21 columns and more (cbca_distance > 21) need a ring of 128 slots: 256 KB of LDS for pass V with 256 threads.  Those
    // run through the generic kernels with 64 threads per workgroup (64 KB).
    const bool small_blocks = ring > 64;
    const bool fast_ok = want != 0 && !small_blocks;
    // costs >= +0: the NaN flags of the input travel in the sign bit of E_h (PMX_CBCA_SIGN=0: test hook; the census source has
    // no input volume to ask)
    const char* es = pmx_opt(ctx, "CBCA_SIGN");
    const bool sign = rows_ok && (census_src || (cv->nonneg && !(es && es[0] == '0')));
    if (march) {
        pmx_stage_scope t(ctx, PMX_STAGE_CBCA_V);
        if (o > 0) {
            const size_t cells = ((size_t)2 * o * W + (size_t)(H - 2 * o) * 2 * o) * cv->D;
            const unsigned nb = (unsigned)((cells + 255) / 256 < 8192 ? (cells + 255) / 256 : 8192);
            hipLaunchKernelGGL(cbca_border_nan_kernel, dim3(nb), dim3(256), 0, ctx->stream, cv->data, H, W, cv->D, o);
        }
        const dim3 grid((Wc + m.NC - 1) / m.NC);
        void (*kern)(cbca_args, cbca_march) = a.range ? cbca_census_march_kernel<2> : o == a.cb ? cbca_census_march_kernel<3> : cbca_census_march_kernel<1>;
        PMX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)march_lds));
        hipLaunchKernelGGL(kern, grid, dim3(m.T), march_lds, ctx->stream, a, m);
        PMX_HIP(hipGetLastError());
        return PMX_OK;
    }
    if (rows_ok) {
        pmx_stage_scope t(ctx, PMX_STAGE_CBCA_H);
        const int T = ((a.R * cv->D + 63) / 64) * 64;
        const dim3 grid((Hc + a.R - 1) / a.R);
        const size_t lds = ((size_t)ring * T + (size_t)a.R * kStage * cv->D) * sizeof(float);
        if (census_src) {
            if (o > 0) {
                const size_t cells = ((size_t)2 * o * W + (size_t)(H - 2 * o) * 2 * o) * cv->D;
                const unsigned nb = (unsigned)((cells + 255) / 256 < 8192 ? (cells + 255) / 256 : 8192);
                hipLaunchKernelGGL(cbca_border_nan_kernel, dim3(nb), dim3(256), 0, ctx->stream, cv->data, H, W, cv->D, o);
            }
            const char* eg = pmx_opt(ctx, "CBCA_GEO");  // 0: the general census source even where the geometry one applies (test hook)
            if (a.range) hipLaunchKernelGGL(HIP_KERNEL_NAME(cbca_h_rows_kernel<true, 2>), grid, dim3(T), lds, ctx->stream, a);
            else if (o == a.cb && !(eg && eg[0] == '0')) hipLaunchKernelGGL(HIP_KERNEL_NAME(cbca_h_rows_kernel<true, 3>), grid, dim3(T), lds, ctx->stream, a);
            else hipLaunchKernelGGL(HIP_KERNEL_NAME(cbca_h_rows_kernel<true, 1>), grid, dim3(T), lds, ctx->stream, a);
        }
        else if (sign) hipLaunchKernelGGL(HIP_KERNEL_NAME(cbca_h_rows_kernel<true, 0>), grid, dim3(T), lds, ctx->stream, a);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(cbca_h_rows_kernel<false, 0>), grid, dim3(T), lds, ctx->stream, a);
    } else {
        pmx_stage_scope t(ctx, PMX_STAGE_CBCA_H);
        int total = Hc * cv->D;
        if (fast_ok && Wc >= 2 * a.A + 8)
            hipLaunchKernelGGL(cbca_h_fast_kernel, dim3((total + kBlock - 1) / kBlock), dim3(kBlock), (size_t)ring * kBlock * sizeof(float),
                               ctx->stream, a);
        else if (small_blocks)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(cbca_h_kernel<64>), dim3((total + 63) / 64), dim3(64), (size_t)ring * 64 * sizeof(float), ctx->stream, a);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(cbca_h_kernel<kBlock>), dim3((total + kBlock - 1) / kBlock), dim3(kBlock),
                               (size_t)ring * kBlock * sizeof(float), ctx->stream, a);
    }
    {
        pmx_stage_scope t(ctx, PMX_STAGE_CBCA_V);
        int total = Wc * cv->D;
        // workgroups of 512 threads read 2 KB contiguous per row: a little kinder to the DRAM pages when the workgroups of a launch
        // have drifted rows apart (10000^2 x 129: 32.3 against 35.1 ms; at 2048^2 x 129 the coarser grid costs more than it gains)
        const char* eb = pmx_opt(ctx, "CBCA_VBS");
        int vbs = eb ? atoi(eb) : ((size_t)total >= ((size_t)1 << 20) ? 512 : 256);
        while (vbs > 256 && (size_t)2 * ring * vbs * sizeof(float) > (size_t)64 * 1024) vbs >>= 1;  // (long arms: the ring decides)
        // (descriptors per row or per quad of rows: see the kernel - the per-row form wins once a row of the volume is several MB)
        const char* er = pmx_opt(ctx, "CBCA_ROWDESC");
        const bool rowdesc = er ? er[0] != '0' : (size_t)cv->W * cv->D * sizeof(float) > ((size_t)5 << 20);
#define PMX_VBUF(SIGNV, BSV)                                                                                                        \
    do {                                                                                                                            \
        if (rowdesc)                                                                                                                \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(cbca_v_buf_kernel<SIGNV, BSV, true>), dim3((total + BSV - 1) / BSV), dim3(BSV),      \
                               (size_t)2 * ring * BSV * sizeof(float), ctx->stream, a);                                             \
        else                                                                                                                        \
            hipLaunchKernelGGL(HIP_KERNEL_NAME(cbca_v_buf_kernel<SIGNV, BSV, false>), dim3((total + BSV - 1) / BSV), dim3(BSV),     \
                               (size_t)2 * ring * BSV * sizeof(float), ctx->stream, a);                                             \
    } while (0)
        if (vbuf && sign && vbs == 512) PMX_VBUF(true, 512);
        else if (vbuf && sign && vbs == 1024) PMX_VBUF(true, 1024);
        else if (vbuf && sign) PMX_VBUF(true, kBlock);
        else if (vbuf) PMX_VBUF(false, kBlock);
#undef PMX_VBUF
        else if (sign)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(cbca_v_fast_kernel<true>), dim3((total + kBlock - 1) / kBlock), dim3(kBlock),
                               (size_t)2 * ring * kBlock * sizeof(float), ctx->stream, a);
        else if (fast_ok && Hc >= 2 * a.A + 8)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(cbca_v_fast_kernel<false>), dim3((total + kBlock - 1) / kBlock), dim3(kBlock),
                               (size_t)2 * ring * kBlock * sizeof(float), ctx->stream, a);
        else if (small_blocks)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(cbca_v_kernel<64>), dim3((total + 63) / 64), dim3(64), (size_t)2 * ring * 64 * sizeof(float),
                               ctx->stream, a);
        else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(cbca_v_kernel<kBlock>), dim3((total + kBlock - 1) / kBlock), dim3(kBlock),
                               (size_t)2 * ring * kBlock * sizeof(float), ctx->stream, a);
    }
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}
