// k_sgmfam.hip - float32 SGM, "family" schedule: three path directions fused into one marching pass.  gfx950.
//
// The per-direction schedule of k_sgm.hip reads the cost volume C and rewrites the accumulator S once per direction
// (8 x 12 B/cell).  The three paths that advance one image row per step - (+1,0), (+1,+1), (+1,-1), or their mirror
// images - share every read of C and every read-modify-write of S when one kernel marches down the rows computing all
// three at each pixel:  R C + R S + W S = 12 B/cell for three directions instead of 36.  The definition (oracle.c
// orc_sgm, k_sgm.hip header) is unchanged: the same float32 operations in the same order, S += L_vertical, then
// L_(pred col-1), then L_(pred col+1).
//
// Decomposition.  Row r needs row r-1 of all three paths, so rows are sequential and columns are the parallel axis.  A
// workgroup owns a window of CW columns; the diagonal paths cross window borders.  With a fixed window the (+1,+1) path
// needs the left neighbour and the (+1,-1) path the right one: a two-sided dependency that locks neighbouring
// workgroups into step (one cross-CU hand-off latency per row).  The window therefore SLIDES LEFT by one column per row,
//     column(r, j) = base - r + j,   j = 0 .. CW-1 (local column), base = s * CW,
// which makes the (+1,-1) path stay in its lane group (registers), the vertical path come from local column j-1 and the
// (+1,+1) path from local column j-2: every cross-window dependency now points to the LEFT neighbour only.  Workgroups
// form a one-directional pipeline: nobody waits for anything its right neighbour produces, so the hand-off latency is
// paid once as pipeline lag and not once per row.  Windows are not wrapped around the image: workgroup s exists for
// s = 0 .. (W+H-2)/CW and is active on the rows where its window meets the image.  The window index comes from an
// atomic ticket, so a workgroup's left neighbour has always started before it: no co-residency assumption, no deadlock.
//
// Inside a workgroup the two shifting paths change lane group every row: they go through LDS (double-buffered by row
// parity, one barrier per row).  The left neighbour's last two columns arrive through global memory as 8-byte
// {tag = launch epoch, value} granules written by ONE sc1 store each (write-through, no fence, the data is the flag;
// cdna_hip_programming.md Guideline 16 form R2) and are read with relaxed agent-scope loads by a dedicated wave of the
// consumer, which spins (bounded) until every tag is this launch's epoch.  The granule buffer is full-size (one slot per
// row and window border), so nothing is ever overwritten within a launch and there is no back-pressure.
//
// Lane map: GL lanes per pixel (16 or 32), KPL consecutive disparities per lane, 64/GL pixels per wave, NW compute
// waves + 1 hand-off wave per workgroup.  Loads of C and S run PF rows ahead in a register ring.  No MFMA: HBM-bound.
#include <cstdlib>

#include "pmx_internal.h"

namespace {

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;

__device__ __forceinline__ float f_inf() { return __int_as_float(0x7f800000); }
__device__ __forceinline__ float f_nan() { return __int_as_float(0x7fc00000); }
__device__ __forceinline__ float fmin2(float a, float b) { return a < b ? a : b; }

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float oldv, float src) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(oldv), __float_as_int(src), CTRL, 0xf, 0xf, false));
}

// minimum over the GL lanes of a pixel, returned in every lane of the group
template <int GL>
__device__ __forceinline__ float group_min(float v) {
    v = fmin2(v, dpp_mov<0x121>(v, v));  // row_ror:1
    v = fmin2(v, dpp_mov<0x122>(v, v));  // row_ror:2
    v = fmin2(v, dpp_mov<0x124>(v, v));  // row_ror:4
    v = fmin2(v, dpp_mov<0x128>(v, v));  // row_ror:8 -> every lane of the 16-lane row holds the row minimum
    if (GL == 32) {                      // the other row of the pixel: lane ^ 16 (ds_swizzle bit mode, no LDS storage)
        const float o = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401f));
        v = fmin2(v, o);
    }
    return v;
}

struct fam_args {
    const float* C;  // raw cost volume [H][W][D] (NaN = invalid)
    float* S;        // accumulator, updated in place
    int H, W, D;
    int flip;        // 0: rows top -> bottom, paths (+1,0) (+1,+1) (+1,-1);  1: bottom -> top, paths (-1,0) (-1,+1) (-1,-1)
    float P1, P2, invalid_cost;
    int is_max, overcounting;
    int has_sin;     // S already holds earlier paths (else this pass starts the sum)
    int epilogue;    // last pass: overcounting, sign, NaN restore
    int dmask;       // bit 0 vertical path, bit 1 diagonal with predecessor column c-1, bit 2 diagonal with predecessor c+1
    unsigned long long* halo;  // granules [H][NB][NGP]
    int NB;          // window borders per row = ceil(W / CW)
    unsigned epoch;
    unsigned* ctl;   // [0] ticket counter (zero at launch), [1] error word
};

constexpr unsigned kSpinLimit = 1u << 21;  // polls before a hand-off gives up (seconds; a healthy wait is microseconds)

template <int KPL>
struct lvals {
    float v[KPL];
};

template <int KPL>
__device__ __forceinline__ lvals<KPL> load_vals(const float* p) {
    lvals<KPL> r;
    __builtin_memcpy(&r, p, sizeof(float) * KPL);
    return r;
}

// One path, one pixel.  Lp: path costs of the predecessor pixel (+inf on padded disparities), M their minimum.
// Returns the minimum of the new costs over the pixel's lanes.
template <int GL, int KPL>
__device__ __forceinline__ float path_update(float (&Lp)[KPL], float M, bool restart, int nvalid, int l, const float (&cc)[KPL],
                                             float P1, float P2, float (&Ln)[KPL]) {
    if (restart) {  // predecessor outside the image: (Lp, M) = (0, 0) reproduces L = C' exactly
#pragma unroll
        for (int k = 0; k < KPL; ++k) Lp[k] = k < nvalid ? 0.f : f_inf();
        M = 0.f;
    }
    float below = dpp_mov<0x138>(f_inf(), Lp[KPL - 1]);  // wave_shr:1  lane l <- lane l-1
    float above = dpp_mov<0x130>(f_inf(), Lp[0]);        // wave_shl:1  lane l <- lane l+1
    if (l == 0) below = f_inf();                         // the neighbouring lane belongs to another pixel
    if (l == GL - 1) above = f_inf();
    const float mp2 = M + P2;
    float lmin = f_inf();
#pragma unroll
    for (int k = 0; k < KPL; ++k) {
        const float lo = (k > 0) ? Lp[k - 1] : below;
        const float hi = (k < KPL - 1) ? Lp[k + 1] : above;
        const float nb = fmin2(lo, hi) + P1;
        float t = fmin2(Lp[k], nb);
        t = fmin2(t, mp2);
        const float lv = cc[k] + (t - M);
        Ln[k] = k < nvalid ? lv : f_inf();
        lmin = fmin2(lmin, Ln[k]);
    }
    return group_min<GL>(lmin);
}

template <int GL, int KPL, int NW, int PF>
__global__ __launch_bounds__((NW + 1) * 64) void sgm_family_kernel(fam_args a) {
    constexpr int NPW = 64 / GL;           // pixels per wave
    constexpr int CW = NW * NPW;           // columns per workgroup window
    constexpr int KS = (KPL + 3) & ~3;     // LDS floats per lane slice (16-byte aligned)
    constexpr int ES = GL * KS + 4;        // LDS floats per (path, column): slices + the minimum
    constexpr int EDIR = (CW + 2) * ES;    // one path: column slots -2 .. CW-1
    constexpr int EBUF = 2 * EDIR;         // one row parity: vertical path, diagonal path
    constexpr int NV = GL * KPL;           // values per handed-off vector
    constexpr int NG = 3 * NV + 3;         // granules per (row, border): V[CW-1], A[CW-1], A[CW-2] + their minima
    constexpr int NQ = (NG + 63) / 64;
    constexpr int NGP = NQ * 64;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    volatile int* ctl = (volatile int*)(lds + 2 * EBUF);  // [0] window index, [1], [2] abort flag by row parity

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) {
        ctl[0] = (int)atomicAdd(a.ctl, 1u);
        ctl[1] = 0;
        ctl[2] = 0;
    }
    __syncthreads();
    const int s = ctl[0];
    const int H = a.H, W = a.W, D = a.D;
    const int base = s * CW;
    const int r_lo = base - W + 1 > 0 ? base - W + 1 : 0;
    const int r_hi = base + CW - 1 < H - 1 ? base + CW - 1 : H - 1;
    if (r_lo > r_hi) return;
    gu32* errw = (gu32*)(a.ctl + 1);

    if (wave == NW) {
        // ---- hand-off wave: brings the left neighbour's columns CW-2, CW-1 of row t into column slots -2, -1 ----
        int ldsoff[NQ];
        bool real[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int idx = q * 64 + lane;
            int vec, off;
            if (idx < 3 * NV) {
                vec = idx / NV;
                const int rem = idx - vec * NV;
                off = (rem / KPL) * KS + rem % KPL;
            } else {
                vec = idx - 3 * NV;
                off = GL * KS;
            }
            // vec 0: vertical path of column CW-1 -> slot -1;  1: diagonal of CW-1 -> slot -1;  2: diagonal of CW-2 -> slot -2
            ldsoff[q] = (vec == 0 ? 0 : EDIR) + (vec == 2 ? 0 : ES) + off;
            // only the vectors of paths that run are ever published
            real[q] = idx < NG && (a.dmask & (vec == 0 ? 1 : 2)) != 0;
        }
        auto fetch_row = [&](int t) -> bool {
            const int cb = base - (t + 1);  // image column of the neighbour's last pixel on row t (0 <= cb < W here)
            const gu64* g = (const gu64*)(a.halo + ((size_t)t * a.NB + cb / CW) * NGP) + lane;
            unsigned long long x[NQ];
            for (unsigned spins = 0;; ++spins) {
                bool ok = true;
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    x[q] = __hip_atomic_load(g + q * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok &= !real[q] || (unsigned)(x[q] >> 32) == a.epoch;
                }
                if (__all(ok)) break;
                if ((spins & 31) == 31 && __hip_atomic_load(errw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
                if (spins > kSpinLimit) {
                    if (lane == 0) __hip_atomic_store(errw, 1u + (unsigned)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    return false;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            float* Eb = lds + (t & 1) * EBUF;
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                if (real[q]) Eb[ldsoff[q]] = __uint_as_float((unsigned)x[q]);
            return true;
        };
        // row r of this window needs the neighbour's row r-1 iff 1 <= r <= base (then 0 <= base - r < W)
        if (r_lo >= 1 && r_lo <= base && !fetch_row(r_lo - 1)) ctl[1 + ((r_lo - 1) & 1)] = 1;
        __syncthreads();
        if (ctl[1 + ((r_lo - 1) & 1)]) return;
        for (int r = r_lo; r <= r_hi; ++r) {
            if (r + 1 <= r_hi && r + 1 <= base && !fetch_row(r)) ctl[1 + (r & 1)] = 1;
            __syncthreads();
            if (ctl[1 + (r & 1)]) return;
        }
        return;
    }

    // ---- compute waves -------------------------------------------------------------------------------------------
    const int g = lane / GL;
    const int l = lane - g * GL;
    const int j = wave * NPW + g;
    const int d0 = l * KPL;
    const bool lane_active = d0 < D;
    const int nvalid = lane_active ? (D - d0 < KPL ? D - d0 : KPL) : 0;
    const int dload = lane_active ? d0 : 0;  // lanes without a disparity read the pixel's d = 0 (loads stay unconditional)
    const bool full = nvalid == KPL;

    // prefetch cursor
    int pr = r_lo;
    auto elem_off = [&](int r, int dd) -> size_t {
        int c = base - r + j;
        c = c < 0 ? 0 : (c > W - 1 ? W - 1 : c);
        const int rimg = a.flip ? H - 1 - r : r;
        return ((size_t)rimg * W + c) * (size_t)D + dd;
    };
    lvals<KPL> cbuf[PF], sbuf[PF];
    auto prefetch = [&](lvals<KPL>& cslot, lvals<KPL>& sslot) {
        const size_t off = elem_off(pr, dload);
        cslot = load_vals<KPL>(a.C + off);
        if (a.has_sin) sslot = load_vals<KPL>(a.S + off);
        if (pr < r_hi) ++pr;
    };
#pragma unroll
    for (int i = 0; i < PF; ++i) prefetch(cbuf[i], sbuf[i]);

    float LB[KPL];  // the path that stays in its lane group (predecessor column c+1)
    float MB = 0.f;
#pragma unroll
    for (int k = 0; k < KPL; ++k) LB[k] = f_inf();

    // hand-off target: this window's columns CW-1 (vectors 0, 1) and CW-2 (vector 2) go to window s+1
    const bool prod_hi = (j == CW - 1), prod_lo = (j == CW - 2);

    __syncthreads();
    if (ctl[1 + ((r_lo - 1) & 1)]) return;

    auto step = [&](int r, lvals<KPL>& cslot, lvals<KPL>& sslot) {
        const int c = base - r + j;
        const bool pix = c >= 0 && c < W;
        const float* Ep = lds + ((r - 1) & 1) * EBUF;
        float* En = lds + (r & 1) * EBUF;
        float cc[KPL];
#pragma unroll
        for (int k = 0; k < KPL; ++k) {
            const float cr = cslot.v[k];
            cc[k] = (cr != cr) ? a.invalid_cost : (a.is_max ? -cr : cr);
        }
        float acc[KPL];
#pragma unroll
        for (int k = 0; k < KPL; ++k) acc[k] = a.has_sin ? sslot.v[k] : 0.f;
        const bool r0 = (r == 0);
        float Ln[KPL];
        // vertical path: predecessor (r-1, c) = local column j-1 of the previous row
        if (a.dmask & 1) {
            float Lp[KPL];
            const float* src = Ep + (j + 1) * ES;
#pragma unroll
            for (int q = 0; q < KS / 4; ++q) {
                const float4 t = *(const float4*)(src + l * KS + 4 * q);
                if (4 * q + 0 < KPL) Lp[4 * q + 0] = t.x;
                if (4 * q + 1 < KPL) Lp[4 * q + 1] = t.y;
                if (4 * q + 2 < KPL) Lp[4 * q + 2] = t.z;
                if (4 * q + 3 < KPL) Lp[4 * q + 3] = t.w;
            }
            const float M = src[GL * KS];
            const float mn = path_update<GL, KPL>(Lp, M, r0, nvalid, l, cc, a.P1, a.P2, Ln);
            float* dst = En + (j + 2) * ES;
#pragma unroll
            for (int q = 0; q < KS / 4; ++q) {
                float4 t;
                t.x = 4 * q + 0 < KPL ? Ln[4 * q + 0] : 0.f;
                t.y = 4 * q + 1 < KPL ? Ln[4 * q + 1] : 0.f;
                t.z = 4 * q + 2 < KPL ? Ln[4 * q + 2] : 0.f;
                t.w = 4 * q + 3 < KPL ? Ln[4 * q + 3] : 0.f;
                *(float4*)(dst + l * KS + 4 * q) = t;
            }
            if (l == 0) dst[GL * KS] = mn;
#pragma unroll
            for (int k = 0; k < KPL; ++k) acc[k] = acc[k] + Ln[k];
            if (prod_hi) {
                const int cb = base + CW - 1 - r;
                if (cb < W && r < H - 1) {
                    gu64* gp = (gu64*)(a.halo + ((size_t)r * a.NB + cb / CW) * NGP);
#pragma unroll
                    for (int k = 0; k < KPL; ++k)
                        __hip_atomic_store(gp + l * KPL + k, ((unsigned long long)a.epoch << 32) | __float_as_uint(Ln[k]),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (l == 0)
                        __hip_atomic_store(gp + 3 * NV, ((unsigned long long)a.epoch << 32) | __float_as_uint(mn), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        // diagonal path with predecessor (r-1, c-1) = local column j-2 of the previous row
        if (a.dmask & 2) {
            float Lp[KPL];
            const float* src = Ep + EDIR + j * ES;
#pragma unroll
            for (int q = 0; q < KS / 4; ++q) {
                const float4 t = *(const float4*)(src + l * KS + 4 * q);
                if (4 * q + 0 < KPL) Lp[4 * q + 0] = t.x;
                if (4 * q + 1 < KPL) Lp[4 * q + 1] = t.y;
                if (4 * q + 2 < KPL) Lp[4 * q + 2] = t.z;
                if (4 * q + 3 < KPL) Lp[4 * q + 3] = t.w;
            }
            const float M = src[GL * KS];
            const float mn = path_update<GL, KPL>(Lp, M, r0 || c == 0, nvalid, l, cc, a.P1, a.P2, Ln);
            float* dst = En + EDIR + (j + 2) * ES;
#pragma unroll
            for (int q = 0; q < KS / 4; ++q) {
                float4 t;
                t.x = 4 * q + 0 < KPL ? Ln[4 * q + 0] : 0.f;
                t.y = 4 * q + 1 < KPL ? Ln[4 * q + 1] : 0.f;
                t.z = 4 * q + 2 < KPL ? Ln[4 * q + 2] : 0.f;
                t.w = 4 * q + 3 < KPL ? Ln[4 * q + 3] : 0.f;
                *(float4*)(dst + l * KS + 4 * q) = t;
            }
            if (l == 0) dst[GL * KS] = mn;
#pragma unroll
            for (int k = 0; k < KPL; ++k) acc[k] = acc[k] + Ln[k];
            if (prod_hi || prod_lo) {
                const int cb = base + CW - 1 - r;
                if (cb < W && r < H - 1) {
                    const int vec = prod_hi ? 1 : 2;
                    gu64* gp = (gu64*)(a.halo + ((size_t)r * a.NB + cb / CW) * NGP);
#pragma unroll
                    for (int k = 0; k < KPL; ++k)
                        __hip_atomic_store(gp + vec * NV + l * KPL + k, ((unsigned long long)a.epoch << 32) | __float_as_uint(Ln[k]),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (l == 0)
                        __hip_atomic_store(gp + 3 * NV + vec, ((unsigned long long)a.epoch << 32) | __float_as_uint(mn),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        // diagonal path with predecessor (r-1, c+1): same local column, stays in registers
        if (a.dmask & 4) {
            MB = path_update<GL, KPL>(LB, MB, r0 || c == W - 1, nvalid, l, cc, a.P1, a.P2, Ln);
#pragma unroll
            for (int k = 0; k < KPL; ++k) {
                LB[k] = Ln[k];
                acc[k] = acc[k] + Ln[k];
            }
        }
        if (a.epilogue) {
#pragma unroll
            for (int k = 0; k < KPL; ++k) {
                float sv = acc[k];
                if (a.overcounting) sv = sv - 7.0f * cc[k];
                if (a.is_max) sv = -sv;
                const float cr = cslot.v[k];
                if (cr != cr) sv = f_nan();
                acc[k] = sv;
            }
        }
        if (pix) {
            const int rimg = a.flip ? H - 1 - r : r;
            float* dst = a.S + ((size_t)rimg * W + c) * (size_t)D + d0;
            if (full) {
                lvals<KPL> out;
#pragma unroll
                for (int k = 0; k < KPL; ++k) out.v[k] = acc[k];
                __builtin_memcpy(dst, &out, sizeof(float) * KPL);
            } else {
#pragma unroll
                for (int k = 0; k < KPL; ++k)
                    if (k < nvalid) dst[k] = acc[k];
            }
        }
        // refill this ring slot with row r + PF (issued after the slot's last use: same registers, no copy)
        prefetch(cslot, sslot);
        __syncthreads();
    };

    int r = r_lo;
    bool dead = false;
    for (; r + PF <= r_hi + 1 && !dead; r += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            if (!dead) {
                step(r + u, cbuf[u], sbuf[u]);
                dead = ctl[1 + ((r + u) & 1)] != 0;
            }
        }
    }
    if (dead) return;
#pragma unroll
    for (int u = 0; u < PF - 1; ++u) {
        if (r + u <= r_hi && !dead) {
            step(r + u, cbuf[u], sbuf[u]);
            dead = ctl[1 + ((r + u) & 1)] != 0;
        }
    }
}

struct fam_shape {
    int gl, kpl, nw;
};

// lane maps that are instantiated: GL 16 up to 144 disparities, GL 32 up to 512
bool pick_shape(int D, int W, fam_shape* out) {
    static const int k16[] = {3, 5, 7, 9}, k32[] = {6, 9, 12, 16};
    fam_shape f{0, 0, 0};
    if (D <= 144) {
        f.gl = 16;
        for (int k : k16)
            if (16 * k >= D) { f.kpl = k; break; }
    } else if (D <= 512) {
        f.gl = 32;
        for (int k : k32)
            if (32 * k >= D) { f.kpl = k; break; }
    }
    if (!f.kpl) return false;
    // window width CW = nw * 64 / gl: the widest one that still gives every CU a window (the hand-off volume per cell
    // falls with CW, the number of busy CUs with W / CW)
    const int npw = 64 / f.gl;
    f.nw = (W / (8 * npw) >= 224) ? 8 : 4;
    if (out) *out = f;
    return true;
}

template <int GL, int KPL, int NW>
int launch_family(pmx_ctx* ctx, const fam_args& a, int nwg) {
    constexpr int PF = KPL > 12 ? 2 : 3;
    constexpr int NPW = 64 / GL, CW = NW * NPW, KS = (KPL + 3) & ~3, ES = GL * KS + 4;
    const size_t lds_bytes = (size_t)(2 * 2 * (CW + 2) * ES + 4) * sizeof(float);
    auto kern = sgm_family_kernel<GL, KPL, NW, PF>;
    PMX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(kern, dim3(nwg), dim3((NW + 1) * 64), lds_bytes, ctx->stream, a);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

int dispatch_family(pmx_ctx* ctx, const fam_shape& f, const fam_args& a, int nwg) {
#define PMX_FAM(GL, KPL)                                                                  \
    if (f.gl == GL && f.kpl == KPL)                                                       \
        return f.nw == 8 ? launch_family<GL, KPL, 8>(ctx, a, nwg) : launch_family<GL, KPL, 4>(ctx, a, nwg);
    PMX_FAM(16, 3)
    PMX_FAM(16, 5)
    PMX_FAM(16, 7)
    PMX_FAM(16, 9)
    PMX_FAM(32, 6)
    PMX_FAM(32, 9)
    PMX_FAM(32, 12)
    PMX_FAM(32, 16)
#undef PMX_FAM
    pmx_set_error("pmx_sgm (family schedule): no kernel for lane map %dx%d", f.gl, f.kpl);
    return PMX_ERR_STATE;
}

}  // namespace

bool pmx_sgm_family_supported(const pmx_cv* cv) { return cv->H >= 2 && pick_shape(cv->D, cv->W, nullptr); }

int pmx_launch_sgm_families(pmx_ctx* ctx, pmx_cv* cv, float* S, float P1, float P2, int is_max, float invalid_cost, int overcounting,
                            int mask) {
    fam_shape f;
    PMX_CHECK(pick_shape(cv->D, cv->W, &f), PMX_ERR_UNSUPPORTED, "pmx_sgm (family schedule): D = %d not supported", cv->D);
    const int npw = 64 / f.gl, CW = f.nw * npw;
    const int NB = (cv->W + CW - 1) / CW;
    const int NG = 3 * f.gl * f.kpl + 3, NGP = (NG + 63) / 64 * 64;
    const size_t halo_bytes = (size_t)cv->H * NB * NGP * sizeof(unsigned long long);
    if (ctx->fam_halo_bytes < halo_bytes) {
        if (ctx->fam_halo) PMX_HIP(hipFree(ctx->fam_halo));
        ctx->fam_halo = nullptr;
        ctx->fam_halo_bytes = 0;
        PMX_HIP(hipMalloc((void**)&ctx->fam_halo, halo_bytes));
        ctx->fam_halo_bytes = halo_bytes;
        // stale tags must never equal a future epoch: zero once, count epochs from 1
        PMX_HIP(hipMemsetAsync(ctx->fam_halo, 0, halo_bytes, ctx->stream));
        ctx->fam_epoch = 0;
    }
    if (!ctx->fam_ctl) {
        PMX_HIP(hipMalloc((void**)&ctx->fam_ctl, 2 * sizeof(unsigned)));
        PMX_HIP(hipMemsetAsync(ctx->fam_ctl, 0, 2 * sizeof(unsigned), ctx->stream));
        PMX_HIP(hipHostMalloc((void**)&ctx->fam_err_host, sizeof(unsigned), hipHostMallocDefault));
        *ctx->fam_err_host = 0;
    }
    const int nwg = (cv->W + cv->H - 2) / CW + 1;
    for (int fam = 0; fam < 2; ++fam) {
        const int bits = (mask >> (2 + 3 * fam)) & 7;  // definition order: vertical, predecessor c-1, predecessor c+1
        if (!bits) continue;
        if (ctx->fam_epoch == 0xffffffffu) {  // epoch space used up: start over on a clean buffer
            PMX_HIP(hipMemsetAsync(ctx->fam_halo, 0, ctx->fam_halo_bytes, ctx->stream));
            ctx->fam_epoch = 0;
        }
        fam_args a;
        a.C = cv->data;
        a.S = S;
        a.H = cv->H; a.W = cv->W; a.D = cv->D;
        a.flip = fam;
        a.P1 = P1; a.P2 = P2; a.invalid_cost = invalid_cost;
        a.is_max = is_max; a.overcounting = overcounting;
        a.has_sin = (mask & ((1 << (2 + 3 * fam)) - 1)) != 0;
        a.epilogue = (mask >> (5 + 3 * fam)) == 0;
        a.dmask = bits;
        a.halo = ctx->fam_halo;
        a.NB = NB;
        a.epoch = ++ctx->fam_epoch;
        a.ctl = ctx->fam_ctl;
        PMX_HIP(hipMemsetAsync(ctx->fam_ctl, 0, sizeof(unsigned), ctx->stream));  // the ticket; the error word is sticky
        {
            pmx_stage_scope t(ctx, PMX_STAGE_SGM_FAMILY);
            int rc = dispatch_family(ctx, f, a, nwg);
            if (rc) return rc;
        }
        // the error word travels to pinned host memory behind the launch; pmx_check_async_error reads it after a sync
        PMX_HIP(hipMemcpyAsync(ctx->fam_err_host, ctx->fam_ctl + 1, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    }
    return PMX_OK;
}
