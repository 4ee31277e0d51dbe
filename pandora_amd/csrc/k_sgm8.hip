// k_sgm8.hip - the integer fast path with PACKED 16-bit arithmetic: census Hamming costs are written once as
// uint8 [H][W][Dp] (1 byte per cell), then all 8 SGM directions run in one launch on two disparities per 32-bit
// register (v_pk_min_u16 / v_pk_add_u16).  gfx950.
//
// Why: the popcount-fused kernel of k_fused.hip sits on the instruction-issue bound (about 176 wave-instructions per
// step for 12 disparities per lane, one instruction per ~4.2 cycles per SIMD, profiles/r01_c_pmc_sq.csv); two of its six
// instructions per disparity recompute the matching cost in every one of the 8 directions.  Reading a precomputed
// byte instead costs 8 B/cell of extra HBM reads - traffic this path has to spare (5.9 GB against 10.8 GB algorithmic)
// - and lets the recurrence run two disparities per instruction:
//     per 4 disparities: 3 unpack (v_and_or / shift), 2 v_alignbit (neighbours d-1 / d+1 across registers),
//     2 x (pk_min, pk_add P1, pk_min, pk_min, pk_sub M, pk_add C) = 17 instructions instead of 24, no validity
//     logic at all (invalid cells carry invalid_cost in the byte), one 4*Q-byte load and one store per step.
// Same semantics as k_fused.hip / k_sgm.hip / the oracle: L = C + min(Lp[d], min(Lp[d-1], Lp[d+1]) + P1, M + P2) - M,
// borders start from (Lp, M) = (0, 0), diagonals wrap, results are exact small integers.
//
// Register layout of a lane (16 lanes per scanline, KPL = 4*Q disparities per lane, d = sub*KPL + 4q + i):
//     A[q] = (L[4q], L[4q+2]) as (lo16, hi16),  B[q] = (L[4q+1], L[4q+3]).
// Disparities >= D ("pads") carry kInf16 in their cost, so they never win a minimum; their stored bytes are garbage
// above the lane's real bytes and are never read.
//
// The path kernel is HBM-bound (9.3 GB per launch at C3, 4.4 of them the 8 reads of the cost bytes) with issue slots to
// spare, so when every cost fits 5 bits (invalid_cost <= 31: census windows up to 5x5) the costs are stored SIX per dword
// (CBITS = 5): a lane's 12 costs are 8 bytes instead of 12, laid out so that one shift + v_and_or_b32 yields a (lo16, hi16)
// register of the recurrence (three such pairs per dword).
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "pmx_buf.h"
#include "pmx_internal.h"

static constexpr int kWaves8 = 4;       // wavefronts per workgroup
static constexpr int kLines8 = 4;       // scanlines per wavefront (16 lanes each)
static constexpr int kRing8 = 4;        // read-ahead (pixels)
static constexpr uint32_t kInf16 = 0x7f00u;
static constexpr uint32_t kInfPk = 0x7f007f00u;

typedef unsigned short us2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
}
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, (us2)(__builtin_bit_cast(us2, a) + __builtin_bit_cast(us2, b)));
}
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, (us2)(__builtin_bit_cast(us2, a) - __builtin_bit_cast(us2, b)));
}
// Small integers compare as positive f16 values exactly as they compare as integers (the bit patterns of positive halves are
// monotone), so the MINIMA of the recurrence may run on the f16 pipe: gfx950's v_pk_minimum3_f16 takes three operands (the
// integer pipe has no packed min3).  Additions stay integer: a 32-bit add of two packed pairs is the packed add as long as no half
// overflows, and v_add3_u32 folds "- M + C" into one instruction (the 32-bit two's complement of (M | M << 16) subtracts M from
// both halves exactly when every half of the result is >= 0).  tools/ubench/pk_probe.hip checks the arithmetic exhaustively.
// Padded disparities carry kPad16 (not a NaN pattern, room above it for P2 and a byte of cost).
static constexpr uint32_t kPad16 = 0x7000u;
static constexpr uint32_t kPadPk = 0x70007000u;
__device__ __forceinline__ uint32_t hmin(uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_pk_min_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t hmin3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t d;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ uint32_t add3(uint32_t a, uint32_t b, uint32_t c) {  // (the compiler would split a + b + (0 - M) into an add and a sub)
    uint32_t d;
    asm("v_add3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

template <int CTRL>
__device__ __forceinline__ uint32_t dpp8(uint32_t oldv, uint32_t src) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)oldv, (int)src, CTRL, 0xf, 0xf, false);
}

// ---- matching costs as bytes ---------------------------------------------------------------------------------
struct cost8_args {
    const uint32_t* codeL;  // [H][W][NW]
    const uint32_t* codeR;  // [H][W][NW], guard dwords on both sides
    uint8_t* cost;          // [H][W][Dp]
    const uint32_t* range;  // [H][W] lo | hi << 16 (cv_masked on this path) or nullptr = census geometry
    int H, W, D, Dp, d0, o;
    uint32_t invalid_cost;
};

// Four pixels per wavefront (one per 16-lane row), lane `sub` owns the same KPL disparities it owns in the path kernel:
// KPL right codes come in as 16-byte loads, KPL bytes leave as one store.  Invalid census cells (window outside the
// image on either side; census.cpp:97-180 leaves them NaN) carry invalid_cost, bytes at d >= D are don't-cares.
template <int NW, int KPL, int CBITS>
__global__ __launch_bounds__(256) void census_cost_u8_kernel(cost8_args a) {
    constexpr int PER = CBITS == 8 ? 4 : 6;          // costs per dword
    constexpr int NDW = (KPL + PER - 1) / PER;       // dwords per lane
    const int lane = threadIdx.x & 63;
    const int sub = lane & 15, grp = lane >> 4;
    const size_t npix = (size_t)a.H * a.W;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t nwaves = (size_t)gridDim.x * 4;
    const int d_first = sub * KPL;
    const bool lane_active = d_first < a.D;
    const uint32_t wvalid = (uint32_t)(a.W - 2 * a.o);
    for (size_t quad = wave; quad * 4 < npix; quad += nwaves) {
        const size_t pix = min(quad * 4 + grp, npix - 1);  // surplus rows repeat the last pixel (same bytes)
        const int r = (int)(pix / a.W), c = (int)(pix - (size_t)r * a.W);
        const bool pix_ok = (r >= a.o) & (r < a.H - a.o) & (c >= a.o) & (c < a.W - a.o);
        uint32_t lc[NW], rc[KPL * NW];
        __builtin_memcpy(lc, a.codeL + pix * NW, sizeof(uint32_t) * NW);
        __builtin_memcpy(rc, a.codeR + ((ptrdiff_t)pix + a.d0 + (lane_active ? d_first : 0)) * NW, sizeof(uint32_t) * KPL * NW);
        const uint32_t u = (uint32_t)(c + a.d0 + d_first - a.o);
        const uint32_t rg = a.range ? a.range[pix] : 0u;
        const int rlo = (int)(rg & 0xffffu) - d_first, rhi = (int)(rg >> 16) - d_first;  // the lane's slots that are numbers
        uint32_t out[NDW];
#pragma unroll
        for (int j = 0; j < NDW; ++j) out[j] = 0;
        // In the interior of the image every cell of every lane of the wavefront is a number (at 4096 x 4096, d = [0, 256]: 94 % of
        // the wavefronts): one wave-uniform test spares them the per-cell validity arithmetic, most of this kernel's instructions
        const bool lane_full = !lane_active || (a.range ? (rlo <= 0 && rhi >= KPL) : (pix_ok && u < wvalid && u + (uint32_t)(KPL - 1) < wvalid));
        const bool all_full = __builtin_amdgcn_ballot_w64(!lane_full) == 0ull;
        auto place = [&](int k, uint32_t v) {
            if (CBITS == 8) {
                out[k / 4] |= v << (8 * (k % 4));
            } else {
                // five-bit costs sit where the path kernel wants them: pair j = (cost 4q+i, cost 4q+i+2), i = j & 1, is the
                // (lo16, hi16) couple of one register, three pairs per dword at bits 0 / 5 / 10 of each half
                const int j = 2 * (k / 4) + (k & 1), half = (k >> 1) & 1;
                out[j / 3] |= v << (5 * (j % 3) + 16 * half);
            }
        };
        auto hamming = [&](int k) {
            uint32_t pop = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) pop += __popc(lc[w] ^ rc[k * NW + w]);
            return pop;
        };
        if (all_full) {  // (uniform)
#pragma unroll
            for (int k = 0; k < KPL; ++k) place(k, hamming(k));
        } else {
#pragma unroll
            for (int k = 0; k < KPL; ++k) {
                const bool ok = a.range ? (k >= rlo && k < rhi) : (pix_ok && (u + (uint32_t)k < wvalid));
                place(k, ok ? hamming(k) : a.invalid_cost);
            }
        }
        if (lane_active) __builtin_memcpy(a.cost + pix * a.Dp + (size_t)sub * NDW * 4, out, 4 * NDW);
    }
}

// ---- the 8 paths ---------------------------------------------------------------------------------------------------
struct sgm8_args {
    const uint8_t* cost;  // [H][W][Dc]: costs of a pixel, lane by lane (CBITS = 8: one byte each; 5: six per dword)
    uint8_t* ldir;        // [8][H][W][Dp], direction volumes dstride bytes apart
    size_t dstride;
    int H, W, D, Dp, Dc;
    uint32_t P1, P2;
};

template <int KPL, int CBITS, bool HF>
__global__ __launch_bounds__(kWaves8 * 64, 4) void sgm_u8_packed_kernel(sgm8_args a) {
    constexpr int Q = KPL / 4;
    constexpr uint32_t kI16 = HF ? kPad16 : kInf16, kIPk = HF ? kPadPk : kInfPk;
    constexpr int PER = CBITS == 8 ? 4 : 6;     // costs per dword of the cost volume
    constexpr int NDW = (KPL + PER - 1) / PER;  // cost dwords per lane
    static_assert(KPL % 4 == 0, "whole dwords per lane");
    const int lane = threadIdx.x & 63;
    const int sub = lane & 15, grp = lane >> 4;
    const int gwave = __builtin_amdgcn_readfirstlane(blockIdx.x * kWaves8 + (threadIdx.x >> 6));
    const int H = a.H, W = a.W, D = a.D;
    const int wavesH = (H + kLines8 - 1) / kLines8, wavesW = (W + kLines8 - 1) / kLines8;
    int dir, l0;  // directions in the order of k_sgm.hip / the oracle; 0,1 walk rows, 2..7 walk columns / diagonals
    if (gwave < 2 * wavesH) {
        dir = gwave / wavesH;
        l0 = (gwave - dir * wavesH) * kLines8;
    } else {
        const int t = gwave - 2 * wavesH;
        dir = 2 + t / wavesW;
        if (dir >= 8) return;
        l0 = (t - (dir - 2) * wavesW) * kLines8;
    }
    const int dr = (dir < 2) ? 0 : ((dir & 1) ? -1 : 1);                                                   // 0 0 +1 -1 +1 -1 +1 -1
    const int dc = (dir == 0) ? 1 : (dir == 1) ? -1 : (dir < 4) ? 0 : ((dir == 4 || dir == 7) ? 1 : -1);  // +1 -1 0 0 +1 -1 -1 +1
    const bool horizontal = (dr == 0);
    const bool diagonal = (dr != 0) && (dc != 0);
    const int nlines = horizontal ? H : W;
    const int nsteps = horizontal ? W : H;
    const int line = min(l0 + grp, nlines - 1);  // surplus groups of the last wave repeat the last line (same bytes)
    const int d_first = sub * KPL;
    const bool lane_active = d_first < D;

    int r = horizontal ? line : (dr > 0 ? 0 : H - 1);
    int c = horizontal ? (dc > 0 ? 0 : W - 1) : line;
    int pc = c;
    int pleft = nsteps - 1;
    const int stride = dr * W + dc;  // pixel stride of one step (before wrapping)
    const uint8_t* pC = a.cost + ((size_t)r * W + c) * a.Dc + (lane_active ? sub * NDW * 4 : 0);
    uint8_t* pO = a.ldir + (size_t)dir * a.dstride + ((size_t)r * W + c) * a.Dp + d_first;

    struct slot_t { uint32_t x[NDW]; };
    slot_t ring[kRing8];
    auto prefetch = [&](slot_t& s) {
        __builtin_memcpy(s.x, pC, 4 * NDW);
        if (pleft > 0) {  // wave-uniform; past the end the last pixel is re-read
            --pleft;
            pC += (ptrdiff_t)stride * a.Dc;
            if (diagonal) {
                pc += dc;
                const bool hi = pc >= W, lo = pc < 0;
                const int fix = hi ? -W : (lo ? W : 0);
                pc += fix;
                pC += (ptrdiff_t)fix * a.Dc;
            }
        }
    };
#pragma unroll
    for (int i = 0; i < kRing8; ++i) prefetch(ring[i]);

    // pad masks (also the restart state of a path): kInf16 in the halves that hold a disparity >= D
    uint32_t padA[Q], padB[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int d = d_first + 4 * q;
        padA[q] = ((d < D) ? 0u : kI16) | (((d + 2 < D) ? 0u : kI16) << 16);
        padB[q] = ((d + 1 < D) ? 0u : kI16) | (((d + 3 < D) ? 0u : kI16) << 16);
    }
    uint32_t A[Q], B[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) { A[q] = padA[q]; B[q] = padB[q]; }
    uint32_t M = 0u;  // group minimum of the previous pixel, in both halves
    const uint32_t P1pk = a.P1 | (a.P1 << 16), P2pk = a.P2 | (a.P2 << 16);

    auto step = [&](slot_t& s) {
        const uint32_t belowB = dpp8<0x111>(kIPk, B[Q - 1]);  // row_shr:1 - previous lane's (.., L[d_first-1])
        const uint32_t aboveA = dpp8<0x101>(kIPk, A[0]);      // row_shl:1 - next lane's (L[d_first+KPL], ..)
        const uint32_t mp2 = HF ? M + P2pk : pk_add(M, P2pk);
        const uint32_t negM = 0u - M;
        uint32_t nA[Q], nB[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            uint32_t ccA, ccB;  // costs of (d, d+2) and of (d+1, d+3)
            if (CBITS == 8) {
                ccA = (s.x[q] & 0x00ff00ffu) | padA[q];
                ccB = ((s.x[q] >> 8) & 0x00ff00ffu) | padB[q];
            } else {  // pair j of the lane: bits 5*(j%3) of both halves of dword j/3 (see census_cost_u8_kernel)
                constexpr uint32_t m5 = 0x001f001fu;
                ccA = ((s.x[(2 * q) / 3] >> (5 * ((2 * q) % 3))) & m5) | padA[q];
                ccB = ((s.x[(2 * q + 1) / 3] >> (5 * ((2 * q + 1) % 3))) & m5) | padB[q];
            }
            // neighbours: A = (d, d+2) has lo = (d-1, d+1), hi = (d+1, d+3) = B;  B has lo = A, hi = (d+2, d+4)
            const uint32_t loA = __builtin_amdgcn_alignbit(B[q], q > 0 ? B[q > 0 ? q - 1 : 0] : belowB, 16);
            const uint32_t hiB = __builtin_amdgcn_alignbit(q < Q - 1 ? A[q < Q - 1 ? q + 1 : 0] : aboveA, A[q], 16);
            if (HF) {
                const uint32_t tA = hmin3(A[q], hmin(loA, B[q]) + P1pk, mp2);
                const uint32_t tB = hmin3(B[q], hmin(A[q], hiB) + P1pk, mp2);
                nA[q] = add3(tA, ccA, negM);
                nB[q] = add3(tB, ccB, negM);
            } else {
                const uint32_t tA = pk_min(pk_min(A[q], pk_add(pk_min(loA, B[q]), P1pk)), mp2);
                const uint32_t tB = pk_min(pk_min(B[q], pk_add(pk_min(A[q], hiB), P1pk)), mp2);
                nA[q] = pk_add(ccA, pk_sub(tA, M));
                nB[q] = pk_add(ccB, pk_sub(tB, M));
            }
        }
        if (lane_active) {
            uint32_t packed[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q) packed[q] = nA[q] | (nB[q] << 8);  // bytes d, d+1, d+2, d+3 (pads spill upwards only)
            __builtin_memcpy(pO, packed, 4 * Q);
        }
        uint32_t m;
        if (HF) {
            m = hmin(nA[0], nB[0]);
#pragma unroll
            for (int q = 1; q < Q; ++q) m = hmin3(m, nA[q], nB[q]);
        } else {
            m = pk_min(nA[0], nB[0]);
#pragma unroll
            for (int q = 1; q < Q; ++q) m = pk_min(m, pk_min(nA[q], nB[q]));
        }
        uint32_t m1 = m & 0xffffu, m2 = m >> 16;
        uint32_t lmin = m1 < m2 ? m1 : m2;
        prefetch(s);
        // min over the 16 lanes of the line, in every lane (rotate butterfly; old = identity lets the DPP fold into v_min)
        {
            uint32_t t;
            t = dpp8<0x128>(0xffffffffu, lmin); lmin = lmin < t ? lmin : t;
            t = dpp8<0x124>(0xffffffffu, lmin); lmin = lmin < t ? lmin : t;
            t = dpp8<0x122>(0xffffffffu, lmin); lmin = lmin < t ? lmin : t;
            t = dpp8<0x121>(0xffffffffu, lmin); lmin = lmin < t ? lmin : t;
        }
        M = lmin | (lmin << 16);
#pragma unroll
        for (int q = 0; q < Q; ++q) { A[q] = nA[q]; B[q] = nB[q]; }
        // advance; a diagonal line that leaves the image re-enters on the other side and the path restarts
        c += dc;
        pO += (ptrdiff_t)stride * a.Dp;
        if (diagonal) {
            const bool hi = c >= W, lo = c < 0;
            const int fix = hi ? -W : (lo ? W : 0);
            c += fix;
            pO += (ptrdiff_t)fix * a.Dp;
            const bool wrapped = hi || lo;
            if (__builtin_amdgcn_ballot_w64(wrapped) != 0ull) {  // once per line and image width: a wave-uniform branch, not 2Q + 1 selects per step
                asm volatile("; path restart" ::);                // (keeps the compiler from flattening the branch into those selects)
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    A[q] = wrapped ? padA[q] : A[q];
                    B[q] = wrapped ? padB[q] : B[q];
                }
                M = wrapped ? 0u : M;
            }
        }
    };

    int i = 0;
    for (; i + kRing8 <= nsteps; i += kRing8) {
#pragma unroll
        for (int j = 0; j < kRing8; ++j) step(ring[j]);
    }
#pragma unroll
    for (int j = 0; j < kRing8 - 1; ++j)
        if (i + j < nsteps) step(ring[j]);
}

// ---- the horizontal pair, summed -----------------------------------------------------------------------------------------
// Family form of the integer path (k_sgmfam8.hip): three byte volumes instead of eight.  The two horizontal paths of a row cannot
// meet in registers (a row's forward costs are W x D values), so the backward pass ADDS into the volume the forward pass wrote:
// a wavefront owns 4 rows, walks them left to right storing L_(0,+1), then right to left adding L_(0,-1) to what it reads back
// through a second read-ahead ring (R cost + W, then R cost + R + W = 4.4 B/cell for two paths; the sums are <= 2 (invalid_cost
// + P2) and bytes add as plain 32-bit adds).  Same recurrence, registers and cost formats as sgm_u8_packed_kernel<.., true>.
// The kernel moves 20 GB at 4096 x 4096 x 257 in 4.6 ms = 4.4 TB/s: HBM-bound even with one wavefront per SIMD.  The two-sided
// walk below (twice the wavefronts, half the steps) ran alone at the same 4.6 ms there and, beside the marching kernel, slowed
// that one from 10.1 to 12.1 ms; it is the form for SHORT images, where this kernel's few wavefronts are latency-bound.
template <int KPL, int CBITS>
__global__ __launch_bounds__(kWaves8 * 64, 4) void sgm_u8_hpair_kernel(sgm8_args a) {
    constexpr int Q = KPL / 4;
    constexpr int PER = CBITS == 8 ? 4 : 6;
    constexpr int NDW = (KPL + PER - 1) / PER;
    static_assert(KPL % 4 == 0, "whole dwords per lane");
    const int lane = threadIdx.x & 63;
    const int sub = lane & 15, grp = lane >> 4;
    const int gwave = __builtin_amdgcn_readfirstlane(blockIdx.x * kWaves8 + (threadIdx.x >> 6));
    const int H = a.H, W = a.W, D = a.D;
    if (gwave * kLines8 >= H) return;
    const int line = min(gwave * kLines8 + grp, H - 1);  // surplus groups of the last wave repeat the last row (same bytes)
    const int d_first = sub * KPL;
    const bool lane_active = d_first < D;
    uint32_t padA[Q], padB[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int d = d_first + 4 * q;
        padA[q] = ((d < D) ? 0u : kPad16) | (((d + 2 < D) ? 0u : kPad16) << 16);
        padB[q] = ((d + 1 < D) ? 0u : kPad16) | (((d + 3 < D) ? 0u : kPad16) << 16);
    }
    const uint32_t P1pk = a.P1 | (a.P1 << 16), P2pk = a.P2 | (a.P2 << 16);
    struct slot_t { uint32_t x[NDW]; };
    struct sum_t { uint32_t x[Q]; };

    auto pass = [&](auto acc_tag) {
        constexpr bool ACC = decltype(acc_tag)::value;
        const int dc = ACC ? -1 : 1;
        const int c0 = ACC ? W - 1 : 0;
        const uint8_t* pC = a.cost + ((size_t)line * W + c0) * a.Dc + (lane_active ? sub * NDW * 4 : 0);
        uint8_t* pO = a.ldir + ((size_t)line * W + c0) * a.Dp + d_first;
        const uint8_t* pI = pO - (lane_active ? 0 : d_first);  // lanes without a disparity re-read lane 0 (nothing is stored)
        int pleft = W - 1;
        slot_t ring[kRing8];
        sum_t prev[kRing8];
        auto prefetch = [&](slot_t& sl, sum_t& pv) {
            __builtin_memcpy(sl.x, pC, 4 * NDW);
            if (ACC) __builtin_memcpy(pv.x, pI, 4 * Q);
            if (pleft > 0) {  // wave-uniform; past the end the last pixel is re-read
                --pleft;
                pC += (ptrdiff_t)dc * a.Dc;
                pI += (ptrdiff_t)dc * a.Dp;
            }
        };
#pragma unroll
        for (int i = 0; i < kRing8; ++i) prefetch(ring[i], prev[i]);
        uint32_t A[Q], B[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) { A[q] = padA[q]; B[q] = padB[q]; }
        uint32_t M = 0u;
        auto step = [&](slot_t& sl, sum_t& pv) {
            const uint32_t belowB = dpp8<0x111>(kPadPk, B[Q - 1]);
            const uint32_t aboveA = dpp8<0x101>(kPadPk, A[0]);
            const uint32_t mp2 = M + P2pk, negM = 0u - M;
            uint32_t nA[Q], nB[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                uint32_t ccA, ccB;
                if (CBITS == 8) {
                    ccA = (sl.x[q] & 0x00ff00ffu) | padA[q];
                    ccB = ((sl.x[q] >> 8) & 0x00ff00ffu) | padB[q];
                } else {
                    constexpr uint32_t m5 = 0x001f001fu;
                    ccA = ((sl.x[(2 * q) / 3] >> (5 * ((2 * q) % 3))) & m5) | padA[q];
                    ccB = ((sl.x[(2 * q + 1) / 3] >> (5 * ((2 * q + 1) % 3))) & m5) | padB[q];
                }
                const uint32_t loA = __builtin_amdgcn_alignbit(B[q], q > 0 ? B[q > 0 ? q - 1 : 0] : belowB, 16);
                const uint32_t hiB = __builtin_amdgcn_alignbit(q < Q - 1 ? A[q < Q - 1 ? q + 1 : 0] : aboveA, A[q], 16);
                const uint32_t tA = hmin3(A[q], hmin(loA, B[q]) + P1pk, mp2);
                const uint32_t tB = hmin3(B[q], hmin(A[q], hiB) + P1pk, mp2);
                nA[q] = add3(tA, ccA, negM);
                nB[q] = add3(tB, ccB, negM);
            }
            if (lane_active) {
                uint32_t packed[Q];
#pragma unroll
                for (int q = 0; q < Q; ++q) packed[q] = (nA[q] | (nB[q] << 8)) + (ACC ? pv.x[q] : 0u);  // bytes d .. d+3 (pads spill upwards only)
                __builtin_memcpy(pO, packed, 4 * Q);
            }
            uint32_t m = hmin(nA[0], nB[0]);
#pragma unroll
            for (int q = 1; q < Q; ++q) m = hmin3(m, nA[q], nB[q]);
            uint32_t m1 = m & 0xffffu, m2 = m >> 16;
            uint32_t lmin = m1 < m2 ? m1 : m2;
            prefetch(sl, pv);
            {
                uint32_t t;
                t = dpp8<0x128>(0xffffffffu, lmin); lmin = lmin < t ? lmin : t;
                t = dpp8<0x124>(0xffffffffu, lmin); lmin = lmin < t ? lmin : t;
                t = dpp8<0x122>(0xffffffffu, lmin); lmin = lmin < t ? lmin : t;
                t = dpp8<0x121>(0xffffffffu, lmin); lmin = lmin < t ? lmin : t;
            }
            M = lmin | (lmin << 16);
#pragma unroll
            for (int q = 0; q < Q; ++q) { A[q] = nA[q]; B[q] = nB[q]; }
            pO += (ptrdiff_t)dc * a.Dp;
        };
        int i = 0;
        for (; i + kRing8 <= W; i += kRing8) {
#pragma unroll
            for (int jj = 0; jj < kRing8; ++jj) step(ring[jj], prev[jj]);
        }
#pragma unroll
        for (int jj = 0; jj < kRing8 - 1; ++jj)
            if (i + jj < W) step(ring[jj], prev[jj]);
    };
    pass(std::false_type{});
    pass(std::true_type{});
}

// The horizontal pair for SHORT images (row tiles of a multi-GPU run): with one wavefront per four rows a 592-row tile has 148
// wavefronts for 1024 SIMDs and every one of them walks 2 x W latency-bound steps (4.1 ms at W = 4096 whatever the height).  Here
// the walk is TWO-SIDED: a workgroup's wavefronts 0, 1 walk four rows each from the left, wavefronts 2, 3 the same rows from the
// right; each stores on the first half of its walk (nobody has been there), all four meet at one barrier, and each adds on the
// second half to what the partner stored (read through a second look-ahead ring with L1-bypassing loads: the bytes came from
// another wavefront).  Twice the wavefronts, half the steps.  At full height it is no faster than the one-sided kernel (both are
// then HBM-bound) and takes more from the marching kernel beside it, so the launcher picks by height.
template <int KPL, int CBITS>
__global__ __launch_bounds__(kWaves8 * 64, 4) void sgm_u8_hpair2_kernel(sgm8_args a) {
    constexpr int Q = KPL / 4;
    constexpr int PER = CBITS == 8 ? 4 : 6;
    constexpr int NDW = (KPL + PER - 1) / PER;
    static_assert(KPL % 4 == 0 && kWaves8 == 4, "whole dwords per lane; two row groups per workgroup");
    const int lane = threadIdx.x & 63;
    const int sub = lane & 15, grp = lane >> 4;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rgroup = blockIdx.x * 2 + (wv & 1);  // four rows
    const bool backward = wv >= 2;
    const int H = a.H, W = a.W, D = a.D;
    if (rgroup * kLines8 >= H) return;  // (both wavefronts of the row group leave together)
    const int line = min(rgroup * kLines8 + grp, H - 1);  // surplus groups of the last wave repeat the last row (same bytes)
    const int d_first = sub * KPL;
    const bool lane_active = d_first < D;
    uint32_t padA[Q], padB[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int d = d_first + 4 * q;
        padA[q] = ((d < D) ? 0u : kPad16) | (((d + 2 < D) ? 0u : kPad16) << 16);
        padB[q] = ((d + 1 < D) ? 0u : kPad16) | (((d + 3 < D) ? 0u : kPad16) << 16);
    }
    const uint32_t P1pk = a.P1 | (a.P1 << 16), P2pk = a.P2 | (a.P2 << 16);
    struct slot_t { uint32_t x[NDW]; };
    struct sum_t { uint32_t x[Q]; };

    const int dc = backward ? -1 : 1;
    const int c0 = backward ? W - 1 : 0;
    const int nfirst = backward ? W - W / 2 : W / 2;  // columns [0, W/2) belong to the forward walk's first half
    const uint8_t* pC = a.cost + ((size_t)line * W + c0) * a.Dc + (lane_active ? sub * NDW * 4 : 0);
    uint8_t* pO = a.ldir + ((size_t)line * W + c0) * a.Dp + d_first;
    int pleft = W - 1;
    slot_t ring[kRing8];
    sum_t prev[kRing8];
    auto fetch_cost = [&](slot_t& sl) {
        __builtin_memcpy(sl.x, pC, 4 * NDW);
        if (pleft > 0) {  // wave-uniform; past the end the last pixel is re-read
            --pleft;
            pC += (ptrdiff_t)dc * a.Dc;
        }
    };
#pragma unroll
    for (int i = 0; i < kRing8; ++i) fetch_cost(ring[i]);
    uint32_t A[Q], B[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) { A[q] = padA[q]; B[q] = padB[q]; }
    uint32_t M = 0u;

    // the partner's bytes: this lane's Q dwords of pixel (line, ci), through the L1-bypassing path
    const __amdgpu_buffer_rsrc_t rsum = __builtin_amdgcn_make_buffer_rsrc((void*)(a.ldir + (size_t)(rgroup * kLines8) * W * a.Dp), 0,
                                                                            (unsigned)(min(kLines8, H - rgroup * kLines8) * W * a.Dp), kRsrcWord3);
    int ci = c0 + dc * nfirst;  // column the sum ring reads next
    int ileft = W - nfirst;
    auto fetch_sum = [&](sum_t& pv) {
        const unsigned off = (unsigned)((line - rgroup * kLines8) * W + ci) * (unsigned)a.Dp + (lane_active ? (unsigned)d_first : 0u);
        if constexpr (Q == 1) {
            pv.x[0] = __builtin_amdgcn_raw_buffer_load_b32(rsum, off, 0, 16);
        } else if constexpr (Q == 2) {
            const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rsum, off, 0, 16);
            pv.x[0] = t.x; pv.x[1] = t.y;
        } else if constexpr (Q == 3) {
            const u32x3 t = __builtin_amdgcn_raw_buffer_load_b96(rsum, off, 0, 16);
            pv.x[0] = t.x; pv.x[1] = t.y; pv.x[2] = t.z;
        } else {
            const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rsum, off, 0, 16);
            pv.x[0] = t.x; pv.x[1] = t.y; pv.x[2] = t.z; pv.x[3] = t.w;
            if constexpr (Q == 5) pv.x[4] = __builtin_amdgcn_raw_buffer_load_b32(rsum, off + 16, 0, 16);
        }
        if (ileft > 1) {  // wave-uniform
            --ileft;
            ci += dc;
        }
    };

    auto step = [&](slot_t& sl, sum_t& pv, auto acc_tag) {
        constexpr bool ACC = decltype(acc_tag)::value;
        const uint32_t belowB = dpp8<0x111>(kPadPk, B[Q - 1]);
        const uint32_t aboveA = dpp8<0x101>(kPadPk, A[0]);
        const uint32_t mp2 = M + P2pk, negM = 0u - M;
        uint32_t nA[Q], nB[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            uint32_t ccA, ccB;
            if (CBITS == 8) {
                ccA = (sl.x[q] & 0x00ff00ffu) | padA[q];
                ccB = ((sl.x[q] >> 8) & 0x00ff00ffu) | padB[q];
            } else {
                constexpr uint32_t m5 = 0x001f001fu;
                ccA = ((sl.x[(2 * q) / 3] >> (5 * ((2 * q) % 3))) & m5) | padA[q];
                ccB = ((sl.x[(2 * q + 1) / 3] >> (5 * ((2 * q + 1) % 3))) & m5) | padB[q];
            }
            const uint32_t loA = __builtin_amdgcn_alignbit(B[q], q > 0 ? B[q > 0 ? q - 1 : 0] : belowB, 16);
            const uint32_t hiB = __builtin_amdgcn_alignbit(q < Q - 1 ? A[q < Q - 1 ? q + 1 : 0] : aboveA, A[q], 16);
            const uint32_t tA = hmin3(A[q], hmin(loA, B[q]) + P1pk, mp2);
            const uint32_t tB = hmin3(B[q], hmin(A[q], hiB) + P1pk, mp2);
            nA[q] = add3(tA, ccA, negM);
            nB[q] = add3(tB, ccB, negM);
        }
        if (lane_active) {
            uint32_t packed[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q) packed[q] = (nA[q] | (nB[q] << 8)) + (ACC ? pv.x[q] : 0u);  // bytes d .. d+3 (pads spill upwards only)
            __builtin_memcpy(pO, packed, 4 * Q);
        }
        uint32_t m = hmin(nA[0], nB[0]);
#pragma unroll
        for (int q = 1; q < Q; ++q) m = hmin3(m, nA[q], nB[q]);
        uint32_t m1 = m & 0xffffu, m2 = m >> 16;
        uint32_t lmin = m1 < m2 ? m1 : m2;
        fetch_cost(sl);
        if (ACC) fetch_sum(pv);
        {
            uint32_t t;
            t = dpp8<0x128>(0xffffffffu, lmin); lmin = lmin < t ? lmin : t;
            t = dpp8<0x124>(0xffffffffu, lmin); lmin = lmin < t ? lmin : t;
            t = dpp8<0x122>(0xffffffffu, lmin); lmin = lmin < t ? lmin : t;
            t = dpp8<0x121>(0xffffffffu, lmin); lmin = lmin < t ? lmin : t;
        }
        M = lmin | (lmin << 16);
#pragma unroll
        for (int q = 0; q < Q; ++q) { A[q] = nA[q]; B[q] = nB[q]; }
        pO += (ptrdiff_t)dc * a.Dp;
    };
    // first half of the walk: nobody has been here, store.  (The cost ring's slot of step i is i % kRing8 throughout.)
    int i = 0;
    for (; i + kRing8 <= nfirst; i += kRing8) {
#pragma unroll
        for (int jj = 0; jj < kRing8; ++jj) step(ring[jj], prev[jj], std::false_type{});
    }
    const int rem = nfirst - i;  // < kRing8, wave-uniform
#pragma unroll
    for (int jj = 0; jj < kRing8 - 1; ++jj)
        if (jj < rem) step(ring[jj], prev[jj], std::false_type{});
    // everybody's first half is in memory before anybody's second half reads it
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // second half: the partner walking the other way has stored its costs here, add.  Step k of this half uses cost slot
    // (rem + k) % kRing8 (the cost ring runs on) and sum slot k % kRing8: compile-time indices per value of rem.
    const int nsecond = W - nfirst;
#pragma unroll
    for (int jj = 0; jj < kRing8; ++jj) fetch_sum(prev[jj]);
    auto second = [&](auto rtag) {
        constexpr int R = decltype(rtag)::value;
        int k = 0;
        for (; k + kRing8 <= nsecond; k += kRing8) {
#pragma unroll
            for (int jj = 0; jj < kRing8; ++jj) step(ring[(R + jj) % kRing8], prev[jj], std::true_type{});
        }
#pragma unroll
        for (int jj = 0; jj < kRing8 - 1; ++jj)
            if (k + jj < nsecond) step(ring[(R + jj) % kRing8], prev[jj], std::true_type{});
    };
    static_assert(kRing8 == 4, "dispatch below");
    switch (rem) {
        case 0: second(std::integral_constant<int, 0>{}); break;
        case 1: second(std::integral_constant<int, 1>{}); break;
        case 2: second(std::integral_constant<int, 2>{}); break;
        default: second(std::integral_constant<int, 3>{}); break;
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------
// Spacing of the eight path volumes: H*W*Dp rounded to 256 bytes.  (A skew between them - 4 KB ... 1 MB, round 1 - showed no
// benefit at C3: the 7 % differences seen between runs follow the box's clock state, not the placement of the buffers.)
size_t pmx_dir_stride(int H, int W, int Dp) {
    return ((size_t)H * W * Dp + 255) & ~(size_t)255;
}

bool pmx_sgm8_supported(int gl, int kpl, int nw) { return gl == 16 && (kpl % 4) == 0 && kpl >= 4 && kpl <= 20 && nw <= 6 && nw != 5; }

int pmx_launch_sgm8(pmx_ctx* ctx, pmx_cv* cv, int kpl, uint32_t P1, uint32_t P2, uint32_t invalid_cost) {
    const int H = cv->H, W = cv->W;
    const int nact = (cv->D + kpl - 1) / kpl;
    const int Dp = nact * kpl;  // multiple of 4
    const size_t vol = pmx_dir_stride(H, W, Dp);
    const int nw = (cv->win * cv->win + 31) / 32;
    // five-bit costs when they fit (PMX_COST5=0 keeps bytes: test hook)
    const char* e5 = getenv("PMX_COST5");
    const bool five = invalid_cost <= 31 && (uint32_t)(cv->win * cv->win) <= 31 && !(e5 && e5[0] == '0');
    const int ndw = five ? (kpl + 5) / 6 : kpl / 4;
    const int Dc = nact * ndw * 4;
    const size_t cvol = (size_t)H * W * Dc;
    if (cv->cost8_bytes < cvol) {
        pmx_pool_free(ctx, cv->cost8);
        cv->cost8 = nullptr;
        cv->cost8_bytes = 0;
        PMX_HIP(pmx_pool_alloc(ctx, (void**)&cv->cost8, cvol + 64));
        cv->cost8_bytes = cvol;
    }
    // Direction families (k_sgmfam8.hip) when a family's sum fits a byte: three volumes (horizontal pair, downward family, upward
    // family) instead of eight.  PMX_SGM8_FAM=0 keeps the eight path volumes, =1 takes the families whatever the size (test hooks).
    // By default for wide images: the marching kernels want one 32-column window per CU and family (2560 columns x 2128 rows:
    // 6.8 ms against 7.6; 2048 x 2048 x 129: 3.7 against 2.9), and a few hundred rows to amortise the pipeline of windows
    // (4096 columns: 336 rows 2.7 ms against 2.8, 592 rows 3.7 against 4.1, 1104 rows 5.5 against 6.6, 2128 rows 8.9 against 12.5, 3072 rows
    // 12.3 against 16.7) - profiles/r03_b_shapes.txt, r03_e_shapes.txt.
    bool fam = pmx_fam8_supported(kpl, H) && 3u * (invalid_cost + P2) <= 255u && W >= 2560 && H >= 480;
    if (const char* ef = getenv("PMX_SGM8_FAM")) {
        if (ef[0] == '0') fam = false;
        if (ef[0] == '1') fam = pmx_fam8_supported(kpl, H) && 3u * (invalid_cost + P2) <= 255u;
    }
    const int nvol = fam ? 3 : 8;
    if (cv->ldir_bytes < (size_t)nvol * vol) {
        pmx_pool_free(ctx, cv->ldir);
        cv->ldir = nullptr;
        cv->ldir_bytes = 0;
        PMX_HIP(pmx_pool_alloc(ctx, (void**)&cv->ldir, (size_t)nvol * vol + 64));
        cv->ldir_bytes = (size_t)nvol * vol;
    }
    cv->nvol = nvol;
    cv->Dp = Dp; cv->gl = 16; cv->kpl = kpl; cv->dstride = vol;
    {
        pmx_stage_scope t(ctx, PMX_STAGE_CENSUS_COST);
        cost8_args c;
        c.codeL = cv->codeL; c.codeR = cv->codeR; c.cost = cv->cost8;
        c.range = cv->has_range ? cv->range : nullptr;
        c.H = H; c.W = W; c.D = cv->D; c.Dp = Dc; c.d0 = cv->d0; c.o = cv->win / 2;
        c.invalid_cost = invalid_cost;
        const size_t want = ((size_t)H * W + 15) / 16;  // 4 pixels per wave, 4 waves per block
        const dim3 grid((unsigned)(want < 65536 ? want : 65536));
#define PMX_COST8(NWV, KPLV)                                                                                                  \
    if (five && NWV == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(census_cost_u8_kernel<NWV, KPLV, (NWV == 1 ? 5 : 8)>), grid, dim3(256), 0, ctx->stream, c); \
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(census_cost_u8_kernel<NWV, KPLV, 8>), grid, dim3(256), 0, ctx->stream, c)
#define PMX_COST8_KPL(NWV)                 \
    switch (kpl) {                         \
        case 4: PMX_COST8(NWV, 4); break;  \
        case 8: PMX_COST8(NWV, 8); break;  \
        case 12: PMX_COST8(NWV, 12); break;\
        case 16: PMX_COST8(NWV, 16); break;\
        default: PMX_COST8(NWV, 20); break;\
    }
        switch (nw) {  // census windows 3x3 / 5x5: one code word, 7x7: two, 9x9: three, 11x11: four, 13x13: six
            case 1: PMX_COST8_KPL(1) break;
            case 2: PMX_COST8_KPL(2) break;
            case 3: PMX_COST8_KPL(3) break;
            case 4: PMX_COST8_KPL(4) break;
            default: PMX_COST8_KPL(6) break;
        }
#undef PMX_COST8_KPL
#undef PMX_COST8
    }
    PMX_HIP(hipGetLastError());
    sgm8_args a;
    a.cost = cv->cost8; a.ldir = cv->ldir; a.dstride = cv->dstride;
    a.H = H; a.W = W; a.D = cv->D; a.Dp = Dp; a.Dc = Dc; a.P1 = P1; a.P2 = P2;
    if (fam) {
        // The horizontal pair has one wavefront per four rows (1024 at 4096 rows: one per SIMD, latency-bound on its own) and is
        // independent of the vertical families: it runs on the context's second stream, beside the marching kernel
        // (PMX_SGM8_OVERLAP=0 keeps everything in line: A/B hook).
        const char* eo = getenv("PMX_SGM8_OVERLAP");
        const bool overlap = !(eo && eo[0] == '0');
        pmx_stage_scope span(ctx, PMX_STAGE_SGM_SPAN);  // fork ... join on the context's stream: the SGM step as the pipeline sees it
        hipStream_t hs = ctx->stream;
        if (overlap) {
            if (!ctx->aux_stream) {
                PMX_HIP(hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking));
                PMX_HIP(hipEventCreateWithFlags(&ctx->aux_fork, hipEventDisableTiming));
                PMX_HIP(hipEventCreateWithFlags(&ctx->aux_join, hipEventDisableTiming));
            }
            PMX_HIP(hipEventRecord(ctx->aux_fork, ctx->stream));  // behind the cost kernel
            PMX_HIP(hipStreamWaitEvent(ctx->aux_stream, ctx->aux_fork, 0));
            hs = ctx->aux_stream;
        }
        {   // volume 0: the horizontal pair
            pmx_stage_scope t(ctx, PMX_STAGE_SGM_FUSED, hs);
            // one wavefront per four rows, or the two-sided walk for short images (PMX_SGM8_HPAIR=1 / 2 forces one: A/B hook)
            const char* ehp = getenv("PMX_SGM8_HPAIR");
            const bool two_sided = ehp ? ehp[0] == '2' : H < 2560;  // (2128 rows: 8.9 against 9.7 ms; 3072 rows: 13.0 against 12.3)
            const int ngroups = (H + kLines8 - 1) / kLines8;
            const dim3 hgrid(two_sided ? (ngroups + 1) / 2 : (ngroups + kWaves8 - 1) / kWaves8), hblock(kWaves8 * 64);
#define PMX_HP(KPLV)                                                                                                       \
    if (two_sided && five) hipLaunchKernelGGL(HIP_KERNEL_NAME(sgm_u8_hpair2_kernel<KPLV, 5>), hgrid, hblock, 0, hs, a);    \
    else if (two_sided) hipLaunchKernelGGL(HIP_KERNEL_NAME(sgm_u8_hpair2_kernel<KPLV, 8>), hgrid, hblock, 0, hs, a);       \
    else if (five) hipLaunchKernelGGL(HIP_KERNEL_NAME(sgm_u8_hpair_kernel<KPLV, 5>), hgrid, hblock, 0, hs, a);             \
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(sgm_u8_hpair_kernel<KPLV, 8>), hgrid, hblock, 0, hs, a)
            switch (kpl) {
                case 4: PMX_HP(4); break;
                case 8: PMX_HP(8); break;
                case 12: PMX_HP(12); break;
                case 16: PMX_HP(16); break;
                default: PMX_HP(20); break;
            }
#undef PMX_HP
        }
        PMX_HIP(hipGetLastError());
        if (overlap) PMX_HIP(hipEventRecord(ctx->aux_join, ctx->aux_stream));
        // volumes 1, 2: the downward and the upward family, one launch
        const int rcf = pmx_launch_sgm_fam8(ctx, cv, kpl, five, Dc, cv->ldir + vol, vol, P1, P2, 3);
        if (overlap) PMX_HIP(hipStreamWaitEvent(ctx->stream, ctx->aux_join, 0));  // whoever reads the volumes next waits for both
        return rcf;
    }
    const int nwaves = 2 * ((H + kLines8 - 1) / kLines8) + 6 * ((W + kLines8 - 1) / kLines8);
    const dim3 grid((nwaves + kWaves8 - 1) / kWaves8), block(kWaves8 * 64);
    {
        pmx_stage_scope t(ctx, PMX_STAGE_SGM_FUSED);
        // PMX_SGM8_HF=0: the minima on the integer pipe (v_pk_min_u16 chains), the round-2 form - an A/B hook
        static const bool hf = !(getenv("PMX_SGM8_HF") && getenv("PMX_SGM8_HF")[0] == '0');
#define PMX_SGM8(KPLV)                                                                                                   \
    if (five && hf) hipLaunchKernelGGL(HIP_KERNEL_NAME(sgm_u8_packed_kernel<KPLV, 5, true>), grid, block, 0, ctx->stream, a); \
    else if (five) hipLaunchKernelGGL(HIP_KERNEL_NAME(sgm_u8_packed_kernel<KPLV, 5, false>), grid, block, 0, ctx->stream, a); \
    else if (hf) hipLaunchKernelGGL(HIP_KERNEL_NAME(sgm_u8_packed_kernel<KPLV, 8, true>), grid, block, 0, ctx->stream, a);    \
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(sgm_u8_packed_kernel<KPLV, 8, false>), grid, block, 0, ctx->stream, a)
        switch (kpl) {
            case 4: PMX_SGM8(4); break;
            case 8: PMX_SGM8(8); break;
            case 12: PMX_SGM8(12); break;
            case 16: PMX_SGM8(16); break;
            default: PMX_SGM8(20); break;
        }
#undef PMX_SGM8
    }
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}
