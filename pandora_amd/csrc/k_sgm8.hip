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
//
// Also in this file: the horizontal pair of the direction-family form (k_sgmfam8.hip runs the six other paths) in its four
// shapes - sgm_u8_hpair_kernel (one wavefront per four rows, cost volume), sgm_u8_hpair2_kernel (the same rows walked from both
// ends), sgm_u8_hpair_codes_kernel (costs made from the census words, round 4) and sgm_u8_hrow_codes_kernel (one row per
// workgroup, a wavefront from each end, costs from the words: short images) - and pmx_launch_sgm8, which picks among them.
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
// the same with a wave-uniform last operand left in its scalar register (one constant-bus read per instruction is allowed; the
// compiler copies a value used by several instructions into a vector register first, once per step of a walk)
__device__ __forceinline__ uint32_t hmin3_s(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t d;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
    return d;
}
__device__ __forceinline__ uint32_t add3_s(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t d;
    asm("v_add3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
    return d;
}
__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc) {  // popcount(x) + acc as ONE instruction, whatever acc is
    uint32_t d;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(acc));
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
    int nact;  // lanes of a pixel that own disparities in the path kernels = units of a pixel in the volume (Dp = nact units)
};

// A lane makes ONE unit of the volume per step: the KPL costs that lane `s` of pixel `pixel` owns in the path kernels (NDW dwords:
// 16 bytes at KPL = 20 and five-bit costs), units numbered t = pixel * nact + s as they lie in memory.  Every lane of a wavefront
// has a unit, a wavefront's store is 64 consecutive units - whole cache lines - and (pixel, s, row, column) advance with the unit
// instead of being divided out per step.  KPL right codes come in as 16-byte loads.  Invalid census cells (window outside the image on
// either side; census.cpp:97-180 leaves them NaN) carry invalid_cost, bytes at d >= D are don't-cares.
// Round 6 (docs/experiments.md 7.34): the kernel is bound by what its SIMDs issue - with nothing loaded and nothing stored it took 1.0
// of its 1.27 ms at 4096 x 4096 x 257 - and a cell is three instructions (xor, bit count, shift-or) whatever the map; until round 6
// four pixels shared a wavefront, one per 16-lane row, which left lanes nact .. 15 of every row idle (3 of 16 at D = 257) and paid a
// 64-bit division per step: 1.23 - 1.33 ms, this form 1.17 - 1.23 (2048 x 2048 x 129: 0.20 -> 0.15), alternated on one box.
constexpr int kUnitSteps = 16;  // steps of 256 units per workgroup (4 ... 16: the same time, 32: slower)
template <int NW, int KPL, int CBITS>
__global__ __launch_bounds__(256) void census_cost_u8_kernel(cost8_args a) {
    constexpr int PER = CBITS == 8 ? 4 : 6;          // costs per dword
    constexpr int NDW = (KPL + PER - 1) / PER;       // dwords per unit
    const int nact = a.nact;
    const size_t nunits = (size_t)a.H * a.W * (size_t)nact;
    const uint32_t wvalid = (uint32_t)(a.W - 2 * a.o);
    const bool ranged = a.range != nullptr;  // (uniform)
    size_t t = (size_t)blockIdx.x * (256 * kUnitSteps) + threadIdx.x;
    if (t >= nunits) return;  // (a lane whose unit of a later step does not exist skips that step)
    size_t pixel = t / (size_t)nact;
    int s = (int)(t - pixel * (size_t)nact);
    int r = (int)(pixel / (size_t)a.W), c = (int)(pixel - (size_t)r * a.W);
    const int ds = 256 % nact, dp = 256 / nact;  // (uniform) the unit 256 further: s + ds, pixel + dp, with a carry
    uint8_t* pC = a.cost + t * (size_t)(NDW * 4);
#pragma unroll 1
    for (int it = 0; it < kUnitSteps; ++it) {
        if (t < nunits) {
            const int d_first = s * KPL;
            uint32_t lc[NW], rc[KPL * NW];
            __builtin_memcpy(lc, a.codeL + pixel * NW, sizeof(uint32_t) * NW);
            __builtin_memcpy(rc, a.codeR + ((ptrdiff_t)pixel + a.d0 + d_first) * NW, sizeof(uint32_t) * KPL * NW);
            const bool pix_ok = (r >= a.o) & (r < a.H - a.o) & (c >= a.o) & (c < a.W - a.o);
            const uint32_t u = (uint32_t)(c + a.d0 + d_first - a.o);
            const uint32_t rg = ranged ? a.range[pixel] : 0u;
            const int rlo = (int)(rg & 0xffffu) - d_first, rhi = (int)(rg >> 16) - d_first;  // the unit's slots that are numbers
            uint32_t out[NDW];
#pragma unroll
            for (int j = 0; j < NDW; ++j) out[j] = 0;
            // In the interior of the image every cell of every lane of the wavefront is a number (at 4096 x 4096, d = [0, 256]: 94 % of
            // the wavefronts): one wave-uniform test spares them the per-cell validity arithmetic, most of this kernel's instructions
            const bool full = ranged ? (rlo <= 0 && rhi >= KPL) : (pix_ok && u < wvalid && u + (uint32_t)(KPL - 1) < wvalid);
            const bool all_full = __builtin_amdgcn_ballot_w64(!full) == 0ull;
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
                    const bool ok = ranged ? (k >= rlo && k < rhi) : (pix_ok && (u + (uint32_t)k < wvalid));
                    place(k, ok ? hamming(k) : a.invalid_cost);
                }
            }
            __builtin_memcpy(pC, out, 4 * NDW);
        }
        t += 256;
        pC += 256 * NDW * 4;
        s += ds;
        int adv = dp;
        if (s >= nact) { s -= nact; ++adv; }
        pixel += adv;
        c += adv;
        while (c >= a.W) { c -= a.W; ++r; }
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
// The kernel moves 20 GB at 4096 x 4096 x 257 in 4.3 - 4.5 ms = 4.5 TB/s: HBM-bound even with one wavefront per SIMD.  The two-sided
// walk below (twice the wavefronts, half the steps) ran alone at the same time there and, beside the marching kernel, slowed
// that one from 10.1 to 12.1 ms; it is a form for SHORT images, where this kernel's few wavefronts are latency-bound (the other,
// round 4's default where it is legal: one row per workgroup from the census words, sgm_u8_hrow_codes_kernel).
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
    const int row0 = gwave * kLines8;
    const int nrows = min(kLines8, H - row0);
    const int lrow = min(grp, nrows - 1);  // surplus groups of the last wave repeat the last row (same bytes)
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
    // Every access goes through a descriptor of the wavefront's rows with a per-lane byte offset: lanes without a disparity store
    // to an out-of-range offset (dropped) instead of sitting out under an exec mask.  A memory instruction inside a conditional
    // block - even one that is always taken - is not counted by the compiler's wait-count pass on the paths behind it: with the
    // store under `if (lane_active)` the four-deep read-ahead ring was waited for with vmcnt(3), one step of look-ahead instead
    // of four; now the counts are the steady state's (PMX_LOOP_ENTRY_DRAIN, pmx_buf.h).
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void*)(a.cost + (size_t)row0 * W * a.Dc), 0, (unsigned)(nrows * W * a.Dc), kRsrcWord3);
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)(a.ldir + (size_t)row0 * W * a.Dp), 0, (unsigned)(nrows * W * a.Dp), kRsrcWord3);
    struct slot_t { uint32_t x[NDW]; };
    struct sum_t { uint32_t x[Q]; };

    auto pass = [&](auto acc_tag) {
        constexpr bool ACC = decltype(acc_tag)::value;
        const int dc = ACC ? -1 : 1;
        const int c0 = ACC ? W - 1 : 0;
        unsigned offC = (unsigned)((lrow * W + c0) * a.Dc) + (lane_active ? (unsigned)sub * NDW * 4u : 0u);
        unsigned offI = (unsigned)((lrow * W + c0) * a.Dp) + (lane_active ? (unsigned)d_first : 0u);  // lanes without a disparity re-read lane 0
        unsigned offO = lane_active ? (unsigned)((lrow * W + c0) * a.Dp) + (unsigned)d_first : kOob;  // ... and store nothing
        const unsigned stepC = (unsigned)(dc * a.Dc), stepP = (unsigned)(dc * a.Dp);
        int pleft = W - 1;
        slot_t ring[kRing8];
        sum_t prev[kRing8];
        auto prefetch = [&](slot_t& sl, sum_t& pv) {
            load_dwords<NDW>(rsC, offC, sl.x);
            if (ACC) load_dwords<Q>(rsO, offI, pv.x);
            if (pleft > 0) {  // wave-uniform; past the end the last pixel is re-read
                --pleft;
                offC += stepC;
                offI += stepP;
            }
        };
#pragma unroll
        for (int i = 0; i < kRing8; ++i) prefetch(ring[i], prev[i]);
        PMX_LOOP_ENTRY_DRAIN();
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
            {
                uint32_t packed[Q];
#pragma unroll
                for (int q = 0; q < Q; ++q) packed[q] = (nA[q] | (nB[q] << 8)) + (ACC ? pv.x[q] : 0u);  // bytes d .. d+3 (pads spill upwards only)
                store_dwords<Q>(rsO, offO, packed);
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
            offO = lane_active ? offO + stepP : kOob;
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

// Cells that are not numbers (census.cpp:132-172 leaves them NaN: the window at column c of the left or at column c + d of the
// right image leaves the image) take invalid_cost.  `vm` = bit k set where the lane's cell k is a number; a pair register holds
// cells (k, k + 2): each half keeps its Hamming cost or takes invalid_cost, pads of disparities >= D are put back by the caller.
__device__ __forceinline__ uint32_t cells_that_are_numbers(bool pix_ok, int u0, int wvalid, int kpl) {
    // cell k is a number iff 0 <= u0 + k < wvalid
    int klo = -u0, khi = wvalid - u0;
    klo = klo < 0 ? 0 : klo;
    khi = khi > kpl ? kpl : khi;
    const uint32_t m = (khi > klo && pix_ok) ? (((1u << khi) - 1u) & ~((1u << klo) - 1u)) : 0u;
    return m;
}
__device__ __forceinline__ uint32_t keep_numbers(uint32_t cc, uint32_t vm, int k, uint32_t invpk) {
    const uint32_t m0 = (uint32_t)__builtin_amdgcn_sbfe((int)vm, k, 1), m2 = (uint32_t)__builtin_amdgcn_sbfe((int)vm, k + 2, 1);
    const uint32_t mask = __builtin_amdgcn_perm(m2, m0, 0x07060100u);  // low half from m0, high half from m2
    return (cc & mask) | (invpk & ~mask);
}

struct hpc_args {
    const uint32_t* codeL;  // [H][W], zeroed guards of kCodePad dwords around the image (k_matching.hip census_codes)
    const uint32_t* codeR;
    uint8_t* ldir;          // volume 0: the pair's sums, [H][W][Dp] bytes in lane-map order
    int H, W, D, Dp, d0, o;
    uint32_t P1, P2, invalid_cost;
};

template <int KPL>
__global__ __launch_bounds__(kWaves8 * 64, 4) void sgm_u8_hpair_codes_kernel(hpc_args a) {
    constexpr int Q = KPL / 4;
    constexpr int NQ = Q + 1;  // 16-byte pieces of right words per group of four pixels
    static_assert(KPL % 4 == 0, "whole dwords per lane");
    const int lane = threadIdx.x & 63;
    const int sub = lane & 15, grp = lane >> 4;
    const int gwave = __builtin_amdgcn_readfirstlane(blockIdx.x * kWaves8 + (threadIdx.x >> 6));
    const int H = a.H, W = a.W, D = a.D, o = a.o;
    if (gwave * kLines8 >= H) return;
    const int line = min(gwave * kLines8 + grp, H - 1);  // surplus groups of the last wave repeat the last row (same bytes)
    const int d_first = sub * KPL;
    const bool lane_active = d_first < D;
    const int nact = (D + KPL - 1) / KPL;
    const int subc = sub < nact ? sub : nact - 1;  // lanes without a disparity read the last lane's words (in bounds, unused)
    uint32_t padA[Q], padB[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int d = d_first + 4 * q;
        padA[q] = ((d < D) ? 0u : kPad16) | (((d + 2 < D) ? 0u : kPad16) << 16);
        padB[q] = ((d + 1 < D) ? 0u : kPad16) | (((d + 3 < D) ? 0u : kPad16) << 16);
    }
    const uint32_t P1pk = a.P1 | (a.P1 << 16), P2pk = a.P2 | (a.P2 << 16);
    const uint32_t* const rowR = a.codeR + (ptrdiff_t)line * W + a.d0 + subc * KPL;  // right word of cell (c, k): rowR[c + k]
    const uint32_t* const rowL = a.codeL + (ptrdiff_t)line * W;
    const bool rows_ok = gwave * kLines8 >= o && gwave * kLines8 + kLines8 - 1 < H - o;  // (uniform)
    const bool line_ok = line >= o && line < H - o;
    const uint32_t wvalid = (uint32_t)(W - 2 * o);
    const uint32_t invpk = a.invalid_cost | (a.invalid_cost << 16);
    const int ngroups = (W + 3) / 4;
    struct words_t { uint32_t r[4 * NQ]; uint32_t l[4]; };
    struct sum_t { uint32_t x[Q]; };

    auto pass = [&](auto acc_tag) {
        constexpr bool ACC = decltype(acc_tag)::value;
        uint8_t* const pix0 = a.ldir + (size_t)line * W * a.Dp + d_first;        // this lane's bytes of pixel (line, 0)
        const uint8_t* const pin0 = pix0 - (lane_active ? 0 : d_first);          // lanes without a disparity re-read lane 0
        auto fetch = [&](int jg, words_t& w) {
            const uint32_t* pr = rowR + 4 * jg;
#pragma unroll
            for (int i = 0; i < NQ; ++i) __builtin_memcpy(&w.r[4 * i], pr + 4 * i, 16);  // (4-byte aligned: the rows start anywhere)
            __builtin_memcpy(w.l, rowL + 4 * jg, 16);
        };
        sum_t prev[4];
        if (ACC) {
#pragma unroll
            for (int uu = 0; uu < 4; ++uu) {  // the first four columns of the walk (from the right): slot = column & 3
                int c = (W - 1) - ((W - 1 - uu) & 3);
                c = c < 0 ? 0 : c;
                __builtin_memcpy(prev[uu].x, pin0 + (size_t)c * a.Dp, 4 * Q);
            }
        }
        uint32_t A[Q], B[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) { A[q] = padA[q]; B[q] = padB[q]; }
        uint32_t M = 0u;

        auto step = [&](int c, const words_t& w, auto u_tag, bool all_ok) {
            constexpr int U = decltype(u_tag)::value;
            // costs of the pixel: (d, d+2) and (d+1, d+3) pairs, padded disparities carry kPad16
            uint32_t ccA[Q], ccB[Q];
            const uint32_t lw = w.l[U];
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const uint32_t a0 = __builtin_popcount(lw ^ w.r[U + 4 * q]) + (padA[q] & 0xffffu);
                const uint32_t a2 = __builtin_popcount(lw ^ w.r[U + 4 * q + 2]) + (padA[q] >> 16);
                const uint32_t b1 = __builtin_popcount(lw ^ w.r[U + 4 * q + 1]) + (padB[q] & 0xffffu);
                const uint32_t b3 = __builtin_popcount(lw ^ w.r[U + 4 * q + 3]) + (padB[q] >> 16);
                ccA[q] = a0 | (a2 << 16);
                ccB[q] = b1 | (b3 << 16);
            }
            if (!all_ok) {  // (uniform; the image's borders)
                asm volatile("; cells that are not numbers" ::);
                const uint32_t vm = cells_that_are_numbers(line_ok && c >= o && c < W - o, c + a.d0 + d_first - o, (int)wvalid, KPL);
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    ccA[q] = keep_numbers(ccA[q], vm, 4 * q, invpk) | padA[q];
                    ccB[q] = keep_numbers(ccB[q], vm, 4 * q + 1, invpk) | padB[q];
                }
            }
            const uint32_t belowB = dpp8<0x111>(kPadPk, B[Q - 1]);
            const uint32_t aboveA = dpp8<0x101>(kPadPk, A[0]);
            const uint32_t mp2 = M + P2pk, negM = 0u - M;
            uint32_t nA[Q], nB[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const uint32_t loA = __builtin_amdgcn_alignbit(B[q], q > 0 ? B[q > 0 ? q - 1 : 0] : belowB, 16);
                const uint32_t hiB = __builtin_amdgcn_alignbit(q < Q - 1 ? A[q < Q - 1 ? q + 1 : 0] : aboveA, A[q], 16);
                const uint32_t tA = hmin3(A[q], hmin(loA, B[q]) + P1pk, mp2);
                const uint32_t tB = hmin3(B[q], hmin(A[q], hiB) + P1pk, mp2);
                nA[q] = add3(tA, ccA[q], negM);
                nB[q] = add3(tB, ccB[q], negM);
            }
            if (lane_active) {
                uint32_t packed[Q];
#pragma unroll
                for (int q = 0; q < Q; ++q) packed[q] = (nA[q] | (nB[q] << 8)) + (ACC ? prev[U].x[q] : 0u);  // bytes d .. d+3 (pads spill upwards only)
                __builtin_memcpy(pix0 + (size_t)c * a.Dp, packed, 4 * Q);
            }
            if (ACC && c >= 4) __builtin_memcpy(prev[U].x, pin0 + (size_t)(c - 4) * a.Dp, 4 * Q);  // (uniform) four steps ahead
            uint32_t m = hmin(nA[0], nB[0]);
#pragma unroll
            for (int q = 1; q < Q; ++q) m = hmin3(m, nA[q], nB[q]);
            uint32_t m1 = m & 0xffffu, m2 = m >> 16;
            uint32_t lmin = m1 < m2 ? m1 : m2;
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
        };
        // one group of four pixels from `cur`, the next group's words on their way into `nxt`
        auto group = [&](int jg, const words_t& cur, words_t& nxt) {
            const int jn = ACC ? jg - 1 : jg + 1;
            if (jn >= 0 && jn < ngroups) fetch(jn, nxt);  // (uniform)
            const int c4 = 4 * jg;
            const bool all_ok = rows_ok && c4 >= o && c4 + 3 < W - o && c4 + a.d0 >= o && c4 + 3 + a.d0 + D - 1 < W - o;
            if (ACC) {
                if (c4 + 3 < W) step(c4 + 3, cur, std::integral_constant<int, 3>{}, all_ok);
                if (c4 + 2 < W) step(c4 + 2, cur, std::integral_constant<int, 2>{}, all_ok);
                if (c4 + 1 < W) step(c4 + 1, cur, std::integral_constant<int, 1>{}, all_ok);
                step(c4, cur, std::integral_constant<int, 0>{}, all_ok);
            } else {
                step(c4, cur, std::integral_constant<int, 0>{}, all_ok);
                if (c4 + 1 < W) step(c4 + 1, cur, std::integral_constant<int, 1>{}, all_ok);
                if (c4 + 2 < W) step(c4 + 2, cur, std::integral_constant<int, 2>{}, all_ok);
                if (c4 + 3 < W) step(c4 + 3, cur, std::integral_constant<int, 3>{}, all_ok);
            }
        };
        words_t wa, wb;
        int jg = ACC ? ngroups - 1 : 0;
        fetch(jg, wa);
        for (int left = ngroups; left > 0; --left) {  // (ONE copy of the four steps in the instruction stream: the words change
            group(jg, wa, wb);                        //  places at the end of a group, 4 (NQ + 1) moves per four steps)
            wa = wb;
            jg += ACC ? -1 : 1;
        }
    };
    pass(std::false_type{});
    pass(std::true_type{});
}

// ---- the horizontal pair for SHORT images, from the census words: one image row per wavefront ---------------------------------
// With four rows per wavefront a 592-row tile (an 8-rank run's) has 148 wavefronts for 1024 SIMDs, each a chain of 2 W steps of
// ~190 instructions: 2.9 of the tile's 3.7 ms with 85 % of the SIMDs empty.  Here a wavefront owns ONE row: 64 lanes x KPL = 4 or 8
// disparities (D <= 256 / 512), a third of the instructions per step and four times the wavefronts.  Same recurrence, same
// bits; the 64-lane minimum is six in-place v_min_u32_dpp and lives in a scalar register.  The costs come from the census words
// (no cost volume: a lane map of its own would need a cost format of its own): a lane's words of pixel c are the row's right words
// c + d_first .. c + d_first + KPL - 1, kept in a ring of 16 registers indexed by (column + k) & 15 - the loop is unrolled 16
// times, so every index is a constant - into which ONE new word per step arrives 16 - KPL steps before its first use; the left
// words of 64 columns sit in one register, lane = column & 63, and are handed out by v_readlane.  The sums leave in natural
// disparity order at the family lane map's pixel stride Dp (the last lane stores what fits).
// The wave's minimum of the 16-bit halves of the new path costs, and - in the two issue slots every cross-lane operand has to
// wait for its producer anyway (six dependent steps: 12 slots) - the matching costs of the NEXT column:
// x[k] = popcount(lw ^ w[k]) + p[k], lw = lane UN of the left words.  One asm block because neither the scheduler nor the hazard
// pass fills those slots (they emit s_nop 1 six times); the block holds its own distances: two instructions between the
// v_readlane and the first reader of its scalar, two between a vector write and the cross-lane read of it.
#define PMX_DPP_MIN(ctrl) "v_min_u32_dpp %0, %0, %0 " ctrl "\n\t"
template <int UN>
__device__ __forceinline__ uint32_t wave_min_and_costs(const uint32_t (&nA)[1], const uint32_t (&nB)[1], uint32_t lwords,
                                                       const uint32_t (&w)[4], const uint32_t (&p)[4], uint32_t (&ccA)[1],
                                                       uint32_t (&ccB)[1]) {
    uint32_t v, x0, x1, x2, x3, lw;
    asm volatile(
        "v_readlane_b32 %7, %10, %11\n\t"
        "v_pk_min_f16 %0, %8, %9\n\t"
        "v_min_u32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t"
        "v_xor_b32 %1, %7, %12\n\t"
        "v_xor_b32 %2, %7, %13\n\t"
        PMX_DPP_MIN("row_shr:1 row_mask:0xf bank_mask:0xf")
        "v_xor_b32 %3, %7, %14\n\t"
        "v_xor_b32 %4, %7, %15\n\t"
        PMX_DPP_MIN("row_shr:2 row_mask:0xf bank_mask:0xf")
        "v_bcnt_u32_b32 %1, %1, %16\n\t"
        "v_bcnt_u32_b32 %2, %2, %17\n\t"
        PMX_DPP_MIN("row_shr:4 row_mask:0xf bank_mask:0xf")
        "v_bcnt_u32_b32 %3, %3, %18\n\t"
        "v_bcnt_u32_b32 %4, %4, %19\n\t"
        PMX_DPP_MIN("row_shr:8 row_mask:0xf bank_mask:0xf")  // lane 15 of every row holds the row's minimum
        "v_lshl_or_b32 %5, %3, 16, %1\n\t"  // A = (cost 0, cost 2)
        "v_lshl_or_b32 %6, %4, 16, %2\n\t"  // B = (cost 1, cost 3)
        PMX_DPP_MIN("row_bcast:15 row_mask:0xa bank_mask:0xf")  // into rows 1, 3
        "s_nop 1\n\t"
        PMX_DPP_MIN("row_bcast:31 row_mask:0xc bank_mask:0xf")  // into rows 2, 3: lane 63 holds the minimum
        : "=&v"(v), "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(ccA[0]), "=&v"(ccB[0]), "=&s"(lw)
        : "v"(nA[0]), "v"(nB[0]), "v"(lwords), "n"(UN), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(p[0]), "v"(p[1]), "v"(p[2]),
          "v"(p[3]));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// ... the same with one more disparity (nE, its right word wE and addend pE): the 65th lane's worth that D = 4 * 64 + 1 needs
template <int UN>
__device__ __forceinline__ uint32_t wave_min_and_costs_x(const uint32_t (&nA)[1], const uint32_t (&nB)[1], uint32_t nE,
                                                         uint32_t lwords, const uint32_t (&w)[4], uint32_t wE, const uint32_t (&p)[4],
                                                         uint32_t pE, uint32_t (&ccA)[1], uint32_t (&ccB)[1], uint32_t& ccE) {
    uint32_t v, x0, x1, x2, x3, lw;
    asm volatile(
        "v_readlane_b32 %7, %12, %13\n\t"
        "v_pk_minimum3_f16 %0, %9, %10, %11\n\t"
        "v_min_u32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t"
        "v_xor_b32 %1, %7, %14\n\t"
        "v_xor_b32 %2, %7, %15\n\t"
        PMX_DPP_MIN("row_shr:1 row_mask:0xf bank_mask:0xf")
        "v_xor_b32 %3, %7, %16\n\t"
        "v_xor_b32 %4, %7, %17\n\t"
        PMX_DPP_MIN("row_shr:2 row_mask:0xf bank_mask:0xf")
        "v_bcnt_u32_b32 %1, %1, %19\n\t"
        "v_bcnt_u32_b32 %2, %2, %20\n\t"
        PMX_DPP_MIN("row_shr:4 row_mask:0xf bank_mask:0xf")
        "v_bcnt_u32_b32 %3, %3, %21\n\t"
        "v_bcnt_u32_b32 %4, %4, %22\n\t"
        PMX_DPP_MIN("row_shr:8 row_mask:0xf bank_mask:0xf")
        "v_lshl_or_b32 %5, %3, 16, %1\n\t"
        "v_lshl_or_b32 %6, %4, 16, %2\n\t"
        PMX_DPP_MIN("row_bcast:15 row_mask:0xa bank_mask:0xf")
        "v_xor_b32 %8, %7, %18\n\t"
        "v_bcnt_u32_b32 %8, %8, %23\n\t"
        PMX_DPP_MIN("row_bcast:31 row_mask:0xc bank_mask:0xf")
        : "=&v"(v), "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(ccA[0]), "=&v"(ccB[0]), "=&s"(lw), "=&v"(ccE)
        : "v"(nA[0]), "v"(nB[0]), "v"(nE), "v"(lwords), "n"(UN), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(wE), "v"(p[0]),
          "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(pE));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
template <int UN>
__device__ __forceinline__ uint32_t wave_min_and_costs(const uint32_t (&nA)[2], const uint32_t (&nB)[2], uint32_t lwords,
                                                       const uint32_t (&w)[8], const uint32_t (&p)[8], uint32_t (&ccA)[2],
                                                       uint32_t (&ccB)[2]) {
    uint32_t v, x0, x1, x2, x3, x4, x5, x6, x7, lw;
    asm volatile(
        "v_readlane_b32 %9, %14, %15\n\t"
        "v_pk_min_f16 %0, %10, %11\n\t"
        "v_pk_minimum3_f16 %0, %0, %12, %13\n\t"
        "v_min_u32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t"
        "v_xor_b32 %1, %9, %16\n\t"
        "v_xor_b32 %2, %9, %17\n\t"
        PMX_DPP_MIN("row_shr:1 row_mask:0xf bank_mask:0xf")
        "v_xor_b32 %3, %9, %18\n\t"
        "v_xor_b32 %4, %9, %19\n\t"
        PMX_DPP_MIN("row_shr:2 row_mask:0xf bank_mask:0xf")
        "v_xor_b32 %5, %9, %20\n\t"
        "v_xor_b32 %6, %9, %21\n\t"
        PMX_DPP_MIN("row_shr:4 row_mask:0xf bank_mask:0xf")
        "v_xor_b32 %7, %9, %22\n\t"
        "v_xor_b32 %8, %9, %23\n\t"
        PMX_DPP_MIN("row_shr:8 row_mask:0xf bank_mask:0xf")
        "v_bcnt_u32_b32 %1, %1, %24\n\t"
        "v_bcnt_u32_b32 %2, %2, %25\n\t"
        PMX_DPP_MIN("row_bcast:15 row_mask:0xa bank_mask:0xf")
        "v_bcnt_u32_b32 %3, %3, %26\n\t"
        "v_bcnt_u32_b32 %4, %4, %27\n\t"
        PMX_DPP_MIN("row_bcast:31 row_mask:0xc bank_mask:0xf")
        : "=&v"(v), "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(x4), "=&v"(x5), "=&v"(x6), "=&v"(x7), "=&s"(lw)
        : "v"(nA[0]), "v"(nB[0]), "v"(nA[1]), "v"(nB[1]), "v"(lwords), "n"(UN), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]),
          "v"(w[5]), "v"(w[6]), "v"(w[7]), "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]));
    ccA[0] = x0 | (x2 << 16);
    ccB[0] = x1 | (x3 << 16);
    ccA[1] = bcnt_acc(x4, p[4]) | (bcnt_acc(x6, p[6]) << 16);
    ccB[1] = bcnt_acc(x5, p[5]) | (bcnt_acc(x7, p[7]) << 16);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
#undef PMX_DPP_MIN

template <int KPL, bool EXTRA>
__global__ __launch_bounds__(128) void sgm_u8_hrow_codes_kernel(hpc_args a) {
    // EXTRA: D = 64 KPL + 1 (d = [0, 256] at KPL 4 - the headline's range): the last disparity rides in lane 63 as a fifth value
    // "E" (every lane computes one, 63 lanes' worth are pads) instead of doubling KPL for a 33rd lane: 10 more instructions per
    // step against 20.
    // A workgroup = one image row, two wavefronts: wavefront 0 walks it from the left, wavefront 1 from the right; each STORES
    // its path costs on the first half of its walk (nobody has been there), both meet at one barrier, and each ADDS on the second
    // half to what the other stored (read 16 columns ahead through the L1-bypassing path: the bytes came from another wavefront).
    // The walk is ONE loop over blocks of 16 columns whose body issues every load and store unconditionally - columns past the
    // row's end and the read-back of the first half carry out-of-range offsets (loads return 0, stores are dropped) - because the
    // compiler's wait-count pass counts only unconditional memory operations (pmx_buf.h): the ring of right words (one per
    // step, 9 to 12 steps ahead) and the read-back ring (16 steps ahead) are then waited for with the steady state's counts.
    constexpr int Q = KPL / 4;
    constexpr unsigned kGuard = 1024u * 4u;  // bytes of zeroed guard in front of a code image (k_matching.hip kCodePad)
    static_assert(KPL == 4 || KPL == 8, "64 lanes x 4 or 8 disparities");
    static_assert(!EXTRA || KPL == 4, "the extra disparity: one asm block written (Q = 1)");
    constexpr int NK = KPL + (EXTRA ? 1 : 0);  // right words a lane reads per column
    const int lane = threadIdx.x & 63;
    const bool backward = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) != 0;
    const int row = blockIdx.x;
    const int H = a.H, W = a.W, D = a.D, o = a.o;
    const int d_first = lane * KPL;
    const bool lane_active = d_first < D;
    const int nact = (D + KPL - 1) / KPL;
    const int lanec = lane < nact ? lane : nact - 1;  // lanes without a disparity read the last lane's words (in bounds, unused)
    uint32_t padA[Q], padB[Q], pad1[KPL];  // (pad1: per disparity, the addend of its popcount)
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int d = d_first + 4 * q;
        padA[q] = ((d < D) ? 0u : kPad16) | (((d + 2 < D) ? 0u : kPad16) << 16);
        padB[q] = ((d + 1 < D) ? 0u : kPad16) | (((d + 3 < D) ? 0u : kPad16) << 16);
    }
#pragma unroll
    for (int k = 0; k < KPL; ++k) pad1[k] = (d_first + k < D) ? 0u : kPad16;
    const uint32_t padE = (lane == 63 ? 0u : kPad16) | (kPad16 << 16);  // (EXTRA) the pair (L[64 KPL] in lane 63, nothing)
    const uint32_t P1pk = a.P1 | (a.P1 << 16), P2pk = a.P2 | (a.P2 << 16);
    const uint32_t invpk = a.invalid_cost | (a.invalid_cost << 16);
    const bool row_ok = row >= o && row < H - o;
    const int wvalid = W - 2 * o;
    // (descriptor of the row, lane-constant offset, scalar offset of the column): no per-step address arithmetic in the vector unit
    const unsigned row_bytes = (unsigned)W * (unsigned)a.Dp;
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc((void*)(a.ldir + (size_t)row * row_bytes), 0, row_bytes, kRsrcWord3);
    const unsigned code_span = (unsigned)W * 4u + 2u * kGuard;  // the row's words and a guard's length on both sides
    const __amdgpu_buffer_rsrc_t rsR =
        __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)(a.codeR + (ptrdiff_t)row * W) - kGuard), 0, code_span, kRsrcWord3);
    const __amdgpu_buffer_rsrc_t rsL =
        __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)(a.codeL + (ptrdiff_t)row * W) - kGuard), 0, code_span, kRsrcWord3);
    const unsigned offR = kGuard + (unsigned)((a.d0 + lanec * KPL) * 4);  // + 4 (c + k): right word of cell (c, k)
    const unsigned offL = kGuard + (unsigned)(lane & 15) * 4u;            // + 4 c16: left word of column c16 + (lane & 15)
    // ONE store per step: the launcher makes the volume's pixel stride a multiple of 8 for this walk (pmx_launch_sgm8), so a
    // lane's 4 Q bytes never reach into the next pixel.  (A workgroup's vector-memory instructions in flight are what this
    // kernel's step time hangs on - with four per step a step took 0.93 us, loads and stores out of range or not: DESIGN 7.22.)
    const unsigned offS = d_first < a.Dp ? (unsigned)d_first : kOob;
    const unsigned offP = lane_active ? (unsigned)d_first : 0u;  // read-back: lanes without a disparity re-read lane 0 (nothing is stored)
    const unsigned offSE = lane == 63 ? 64u * KPL : kOob, offPE = lane == 63 ? 64u * KPL : 0u;  // (EXTRA) the last disparity's byte
    bool blk_ok = false;  // (uniform) every cell of the block's 16 columns is a number
    bool second = false;  // (uniform) past the meeting point: the other wavefront's bytes are read back and added
    const int nblk = (W + 15) / 16, hb = nblk / 2;  // blocks [0, hb) are the forward walk's first half

    uint32_t wr[16];        // right words: slot (column + k) & 15
    uint32_t Lcur, Lnxt;    // left words of this block of 16 columns and of the next: lane & 15 = column & 15
    uint32_t prev[16][Q];   // what the other wavefront stored: slot column & 15, requested 16 columns ahead (zeros on the first half)
#pragma unroll
    for (int sl = 0; sl < 16; ++sl)
#pragma unroll
        for (int q = 0; q < Q; ++q) prev[sl][q] = 0u;
    uint32_t prevE[EXTRA ? 16 : 1];
#pragma unroll
    for (int sl = 0; sl < (EXTRA ? 16 : 1); ++sl) prevE[sl] = 0u;
    uint32_t A[Q], B[Q], E = padE;
#pragma unroll
    for (int q = 0; q < Q; ++q) { A[q] = padA[q]; B[q] = padB[q]; }
    uint32_t M = 0u;  // (wave-uniform: a scalar register)
    uint32_t belowB = kPadPk, aboveA = kPadPk;  // the neighbouring lanes' edge sums (step)

    auto load_prev = [&](int c, uint32_t (&pv)[Q]) {  // the other wavefront's bytes of column c (sc1: past this CU's L1)
        const unsigned so = (unsigned)c * (unsigned)a.Dp;
        if constexpr (Q == 1) {
            pv[0] = __builtin_amdgcn_raw_buffer_load_b32(rsO, offP, so, 16);
        } else {
            const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rsO, offP, so, 16);
            pv[0] = t.x; pv[1] = t.y;
        }
    };
    uint32_t nccA[Q], nccB[Q], nccE = padE;  // the costs of the column the next step works on
    auto costs_of = [&](auto u_tag, uint32_t lwords) {  // column (block) + U: its left word is lane U of lwords
        constexpr int U = decltype(u_tag)::value;
        const uint32_t lw = (uint32_t)__builtin_amdgcn_readlane((int)lwords, U);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const uint32_t a0 = bcnt_acc(lw ^ wr[(U + 4 * q) & 15], pad1[4 * q]);
            const uint32_t a2 = bcnt_acc(lw ^ wr[(U + 4 * q + 2) & 15], pad1[4 * q + 2]);
            const uint32_t b1 = bcnt_acc(lw ^ wr[(U + 4 * q + 1) & 15], pad1[4 * q + 1]);
            const uint32_t b3 = bcnt_acc(lw ^ wr[(U + 4 * q + 3) & 15], pad1[4 * q + 3]);
            nccA[q] = a0 | (a2 << 16);  // (v_lshl_or_b32)
            nccB[q] = b1 | (b3 << 16);
        }
        if constexpr (EXTRA) nccE = bcnt_acc(lw ^ wr[(U + KPL) & 15], padE);
    };
    auto step = [&](int c, auto u_tag, auto dir_tag) {
        constexpr int U = decltype(u_tag)::value;
        constexpr bool BACK = decltype(dir_tag)::value;
        uint32_t ccA[Q], ccB[Q];  // this column's costs: made during the previous step (costs_of below)
#pragma unroll
        for (int q = 0; q < Q; ++q) { ccA[q] = nccA[q]; ccB[q] = nccB[q]; }
        uint32_t ccE = nccE;
        // (uniform; the image's borders - no memory operation in here; blk_ok: nothing to look at in this block of 16 columns)
        if (!blk_ok && !(row_ok && c >= o && c < W - o && c + a.d0 >= o && c + a.d0 + D - 1 < W - o)) {
            asm volatile("; cells that are not numbers" ::);
            const uint32_t vm = cells_that_are_numbers(row_ok && c >= o && c < W - o, c + a.d0 + d_first - o, wvalid, KPL);
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                ccA[q] = keep_numbers(ccA[q], vm, 4 * q, invpk) | padA[q];
                ccB[q] = keep_numbers(ccB[q], vm, 4 * q + 1, invpk) | padB[q];
            }
            if constexpr (EXTRA) {  // (uniform: one disparity)
                const int ue = c + a.d0 + 64 * KPL - o;
                if (!(row_ok && c >= o && c < W - o && ue >= 0 && ue < wvalid)) ccE = a.invalid_cost | padE;
            }
            if (BACK && c >= W) {  // the backward walk's first block may begin past the row's end: the path starts at column W - 1
#pragma unroll
                for (int q = 0; q < Q; ++q) { A[q] = padA[q]; B[q] = padB[q]; }
                E = padE;
                M = 0u;
            }
        }
        // one new right word: forward, column c's first word is dead and column c + 16's arrives in its slot; backward, column
        // c's last word is dead and the first word of column c + NK - 17 arrives (the descriptor starts a guard's length before
        // the row: scalar offsets are unsigned)
        if (BACK) wr[(U + NK - 1) & 15] = __builtin_amdgcn_raw_buffer_load_b32(rsR, offR - 64u + (unsigned)(NK - 1) * 4u, (unsigned)c * 4u, 0);
        else wr[U] = __builtin_amdgcn_raw_buffer_load_b32(rsR, offR + 64u, (unsigned)c * 4u, 0);
        // wave_shr:1 - the previous lane's (.., L[d_first - 1]); wave_shl:1 - the next lane's (L[d_first + KPL], ..).  Lane 0 of
        // the first and lane 63 of the second are never written by the shift: they keep their +inf from before the loop, and
        // the registers stay where they are (no constant to load per step)
        belowB = dpp8<0x138>(belowB, B[Q - 1]);
        aboveA = dpp8<0x130>(EXTRA ? E : aboveA, A[0]);  // (EXTRA: lane 63's upper neighbour is its own fifth value)
        const uint32_t mp2 = M + P2pk, negM = 0u - M;  // (scalar registers, and read as such: hmin3_s, add3_s)
        uint32_t nA[Q], nB[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const uint32_t loA = __builtin_amdgcn_alignbit(B[q], q > 0 ? B[q > 0 ? q - 1 : 0] : belowB, 16);
            const uint32_t hiB = __builtin_amdgcn_alignbit(q < Q - 1 ? A[q < Q - 1 ? q + 1 : 0] : aboveA, A[q], 16);
            const uint32_t tA = hmin3_s(A[q], hmin(loA, B[q]) + P1pk, mp2);
            const uint32_t tB = hmin3_s(B[q], hmin(A[q], hiB) + P1pk, mp2);
            nA[q] = add3_s(tA, ccA[q], negM);
            nB[q] = add3_s(tB, ccB[q], negM);
        }
        uint32_t nE = 0u;
        if constexpr (EXTRA) {  // low half: L[64 KPL], whose lower neighbour is the lane's last B half; high half: a pad that stays one
            const uint32_t tE = hmin3_s(E, (B[Q - 1] >> 16) + P1pk, mp2);
            nE = add3_s(tE, ccE, negM);
        }
        {   // bytes d .. d+3 per dword (pads spill upwards only); on the second half the other wavefront's bytes are added
            const unsigned so = (unsigned)c * (unsigned)a.Dp;  // (past the row's end: out of range, dropped)
            uint32_t packed[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q) packed[q] = (nA[q] | (nB[q] << 8)) + prev[U][q];
            if constexpr (Q == 2) {
                u32x2 t; t.x = packed[0]; t.y = packed[1];
                __builtin_amdgcn_raw_buffer_store_b64(t, rsO, offS, so, 0);
            } else {
                __builtin_amdgcn_raw_buffer_store_b32(packed[0], rsO, offS, so, 0);
            }
            if constexpr (EXTRA) __builtin_amdgcn_raw_buffer_store_b8((uint8_t)(nE + prevE[U]), rsO, offSE, so, 0);
        }
        // the slot just used: the other wavefront's bytes 16 columns ahead (columns outside the row: zeros).  Under a (uniform)
        // branch: the compiler's wait counts then ignore these loads on the merged path, which only makes the waits for the
        // right words a few operations earlier than needed - they are requested 9 to 12 steps ahead
        if (second) {
            load_prev(BACK ? c - 16 : c + 16, prev[U]);
            if constexpr (EXTRA)
                prevE[U] = __builtin_amdgcn_raw_buffer_load_b8(rsO, offPE, (unsigned)(BACK ? c - 16 : c + 16) * (unsigned)a.Dp, 16);
        }
        // the wave's minimum; the NEXT column's costs do not hang on it and are made in the reduction's idle issue slots
        uint32_t lmin;
        {
            constexpr int UN = BACK ? (U + 15) & 15 : (U + 1) & 15;
            uint32_t wn[KPL];
#pragma unroll
            for (int kk = 0; kk < KPL; ++kk) wn[kk] = wr[(UN + kk) & 15];
            const uint32_t lwords = (BACK ? U == 0 : U == 15) ? Lnxt : Lcur;
            if constexpr (EXTRA) lmin = wave_min_and_costs_x<UN>(nA, nB, nE, lwords, wn, wr[(UN + KPL) & 15], pad1, padE, nccA, nccB, nccE);
            else lmin = wave_min_and_costs<UN>(nA, nB, lwords, wn, pad1, nccA, nccB);
        }
        asm("s_pack_ll_b32_b16 %0, %1, %1" : "=s"(M) : "s"(lmin));  // lmin | lmin << 16
#pragma unroll
        for (int q = 0; q < Q; ++q) { A[q] = nA[q]; B[q] = nB[q]; }
        E = nE;
    };
    auto walk = [&](auto dir_tag) {
        constexpr bool BACK = decltype(dir_tag)::value;
        const int b0 = BACK ? nblk - 1 : 0, db = BACK ? -1 : 1;
        {   // prologue: the first block's words (columns below 0 / past the row read the guards or other rows: never used)
            const int c16 = 16 * b0;
            const int top = BACK ? c16 + 15 + NK - 1 : c16 + 15;  // highest word index (column + k) of the first 16 needed
#pragma unroll
            for (int sl = 0; sl < 16; ++sl)
                wr[sl] = __builtin_amdgcn_raw_buffer_load_b32(rsR, offR, (unsigned)(top - ((top - sl) & 15)) * 4u, 0);
            Lcur = __builtin_amdgcn_raw_buffer_load_b32(rsL, offL, (unsigned)c16 * 4u, 0);
            Lnxt = __builtin_amdgcn_raw_buffer_load_b32(rsL, BACK ? offL - 64u : offL + 64u, (unsigned)c16 * 4u, 0);
        }
        PMX_LOOP_ENTRY_DRAIN();
        if constexpr (BACK) costs_of(std::integral_constant<int, 15>{}, Lcur);
        else costs_of(std::integral_constant<int, 0>{}, Lcur);
        for (int n = 0; n < nblk; ++n) {
            const int b = b0 + db * n;
            const int c16 = 16 * b;
            blk_ok = row_ok && c16 >= o && c16 + 15 < W - o && c16 + a.d0 >= o && c16 + 15 + a.d0 + D - 1 < W - o;
            if (b == (BACK ? hb - 1 : hb)) {
                // the meeting point: everybody's first half is in memory before anybody's second half reads it; from here on the
                // read-back is real (this block's 16 columns now, then 16 columns ahead in every step)
                __builtin_amdgcn_s_waitcnt(0x0F70);
                __syncthreads();
                second = true;
#pragma unroll
                for (int sl = 0; sl < 16; ++sl) {
                    load_prev(c16 + sl, prev[sl]);
                    if constexpr (EXTRA) prevE[sl] = __builtin_amdgcn_raw_buffer_load_b8(rsO, offPE, (unsigned)(c16 + sl) * (unsigned)a.Dp, 16);
                }
                PMX_LOOP_ENTRY_DRAIN();
            }
#define PMX_HROW_STEP(UV) step(c16 + UV, std::integral_constant<int, UV>{}, dir_tag)
            if (BACK) {
                PMX_HROW_STEP(15); PMX_HROW_STEP(14); PMX_HROW_STEP(13); PMX_HROW_STEP(12); PMX_HROW_STEP(11); PMX_HROW_STEP(10);
                PMX_HROW_STEP(9); PMX_HROW_STEP(8); PMX_HROW_STEP(7); PMX_HROW_STEP(6); PMX_HROW_STEP(5); PMX_HROW_STEP(4);
                PMX_HROW_STEP(3); PMX_HROW_STEP(2); PMX_HROW_STEP(1); PMX_HROW_STEP(0);
            } else {
                PMX_HROW_STEP(0); PMX_HROW_STEP(1); PMX_HROW_STEP(2); PMX_HROW_STEP(3); PMX_HROW_STEP(4); PMX_HROW_STEP(5);
                PMX_HROW_STEP(6); PMX_HROW_STEP(7); PMX_HROW_STEP(8); PMX_HROW_STEP(9); PMX_HROW_STEP(10); PMX_HROW_STEP(11);
                PMX_HROW_STEP(12); PMX_HROW_STEP(13); PMX_HROW_STEP(14); PMX_HROW_STEP(15);
            }
#undef PMX_HROW_STEP
            // the left words: the next block's have had 16 steps to arrive, the one after's are requested
            Lcur = Lnxt;
            Lnxt = __builtin_amdgcn_raw_buffer_load_b32(rsL, BACK ? offL - 128u : offL + 128u, (unsigned)c16 * 4u, 0);
        }
    };
    if (backward) walk(std::true_type{});
    else walk(std::false_type{});
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
    int Dp = nact * kpl;  // multiple of 4
    size_t vol = pmx_dir_stride(H, W, Dp);
    const int nw = (cv->win * cv->win + 31) / 32;
    // five-bit costs when they fit (PMX_COST5=0 keeps bytes: test hook)
    const char* e5 = pmx_opt(ctx, "COST5");
    const bool five = invalid_cost <= 31 && (uint32_t)(cv->win * cv->win) <= 31 && !(e5 && e5[0] == '0');
    const int ndw = five ? (kpl + 5) / 6 : kpl / 4;
    const int Dc = nact * ndw * 4;
    const size_t cvol = (size_t)H * W * Dc;
    // Direction families (k_sgmfam8.hip) when a family's sum fits a byte: three volumes (horizontal pair, downward family, upward
    // family) instead of eight.  PMX_SGM8_FAM=0 keeps the eight path volumes, =1 takes the families whatever the size (test hooks).
    // By default for wide images: the marching kernels want one 32-column window per CU and family (round 3: 2560 columns x 2128 rows:
    // 6.8 ms against 7.6; 2048 x 2048 x 129: 3.7 against 2.9), and a few hundred rows to amortise the pipeline of windows
    // (4096 columns: 336 rows 2.7 ms against 2.8, 592 rows 3.7 against 4.1, 1104 rows 5.5 against 6.6, 2128 rows 8.9 against 12.5, 3072 rows
    // 12.3 against 16.7) - profiles/r03_b_shapes.txt, r03_e_shapes.txt.
    // Round 6, re-measured after the marching kernel's XCD-local hand-off (profiles/r06_fam_rule.txt): the eight-volume route costs
    // 4.7 - 6.8 ps per cell (SGM + its WTA), the family form ~(0.48 + 0.058 kpl) us per image ROW whatever the width plus 0.55 ps per
    // cell for its WTA - so the families win when a row holds enough cells.  Break-even measured at W D = 284 000 for kpl = 12
    // (2048 rows x 2200 x 129: 3.1 ms either way; 2048 x 129: 2.80 against 2.98) and 390 000 for kpl = 16 (2048 x 191), families
    // ahead at 195 000 for kpl = 8 (3000 x 3000 x 65: 3.8 against 4.4) and at 526 000 for kpl = 20 (4096 rows x 2048 x 257: 8.5 against
    // 11.1): a line just below the two break-evens, W D >= 26500 kpl - 50000 (2600 columns x 65 stay with the families, as in rounds
    // 3 - 5), given 32-column windows (W >= 2048; the 16-column ones
    // are slower than either route).  Four disparities per lane (D <= 64) keep round 3's bound, which is all that was measured there.
    // Short images (round 6, profiles/r06_fam_rows.txt): until then nothing below 480 rows took the families; on the round's kernels they
    // are ahead from ~200 rows when a row holds 1.8 times the cells of the tall images' bound - 128 / 200 / 300 / 400 rows x 4096 x 257:
    // 1.41 / 1.54 / 1.81 / 2.00 ms against 1.74 / 2.20 / 2.60 / 3.25, 300 / 400 rows x 4096 x 129: 1.44 / 1.55 against 1.61 / 1.75 (200
    // rows: equal), 300 x 6000 x 193: 2.40 against 3.02; 300 x 3000 x 129 (1.12 against 1.01) and 2600 x 65 stay with the eight volumes.
    const size_t row_cells = (size_t)W * cv->D;
    const bool wide_tall = kpl < 8 ? W >= 2560 : (W >= 2048 && row_cells + 50000 >= (size_t)26500 * kpl);
    const bool wide_short = kpl >= 8 && W >= 2048 && 5 * (row_cells + 50000) >= (size_t)9 * 26500 * kpl;
    bool fam = pmx_fam8_supported(kpl, H) && 3u * (invalid_cost + P2) <= 255u && (H >= 480 ? wide_tall : (H >= 192 && wide_short));
    if (const char* ef = pmx_opt(ctx, "SGM8_FAM")) {
        if (ef[0] == '0') fam = false;
        if (ef[0] == '1') fam = pmx_fam8_supported(kpl, H) && 3u * (invalid_cost + P2) <= 255u;
    }
    // The horizontal pair: one wavefront per four rows (mode 1, tall images), the two-sided walk of the same lane map (mode 2), or -
    // round 4, short images such as the row tiles of a multi-GPU run - one row per wavefront from the census words (mode 3:
    // sgm_u8_hrow_codes_kernel).  PMX_SGM8_HPAIR=1 / 2 / 3 forces one: A/B hook.
    // The family form WITHOUT a cost volume (round 4): with one census word per pixel (windows 3x3, 5x5) and the plain census
    // geometry (no valid intervals from masks / grids) the SGM kernels can make a cell's Hamming cost where they use it - the cost
    // kernel, its 0.8 B/cell of stores and the three reads of them go, 2.5 instead of 1 vector instruction per cell and pass
    // come.  At 4096 x 4096 x 257 the step is no faster that way (14.1 against 13.8 ms: 25 % fewer bytes, 13 % more
    // instructions), so tall images keep the cost volume; short ones, whose SIMDs are mostly idle, drop it together
    // with the row-per-wavefront walk.  PMX_SGM8_CODES=1 / 0: both kernels from the words wherever legal / never.
    // The code-word kernels reach the right image's words through per-lane offsets from a guard of 1024 zeroed dwords in front of
    // (and behind) each code image (k_matching.hip kCodePad; sgm_u8_hrow_codes_kernel's offR, sgm_u8_hpair_codes_kernel's raw
    // pointers): a range that starts or ends further than that from the pixel would wrap the unsigned offset (loads answer 0 for
    // cells that ARE numbers) or leave the allocation.  Such ranges keep the cost volume (census_cost_u8_kernel handles any d0).
    const bool codes_ok = nw == 1 && !cv->has_range && cv->D <= 512 && abs(cv->d0) + cv->D <= 1024 - 64;
    const char* ehp = pmx_opt(ctx, "SGM8_HPAIR");
    const char* ec = pmx_opt(ctx, "SGM8_CODES");
    const bool codes_never = ec && ec[0] == '0', codes_always = ec && ec[0] == '1';
    // (4096 columns x 257, one GPU, ms per step, row walk with the marching kernel from the words / row walk with the marching
    //  kernel on the cost volume / two-sided / one-sided: 592 rows 2.55 / 2.65 / 3.5 / 4.9, 1104 rows 4.35 / 4.6 / 5.0 / 6.7,
    //  2088 rows 8.85 / 8.3 / 8.55 / 9.7, 3072 rows - / - / 13.0 / 12.3, 4096 rows 17.2 / 15.9 / - / 14.6: profiles/r04_tiles.txt,
    //  r03_e_shapes.txt)
    // Round 6, re-measured on the round's kernels (profiles/r06_hpair_rule.txt: 16 shapes x 3 modes x the marching kernel from the words
    // / from the cost volume, a fresh context each): the row walk wins below ~1000 rows (480 ... 800 rows x 4096 x 257: 2.2 - 3.2 ms
    // against 2.7 - 3.7 two-sided), the two-sided walk from there to ~2000 rows (1104 rows 4.2 - 4.4 against 4.7 - 5.1, 1536 rows
    // 4.9 - 5.2 against 6.1 - 6.2, 1500 x 2600 x 65 1.8 against 2.2 - 2.4; 2000 - 2088 rows: one- and two-sided equal), the one-sided
    // walk from 2048 rows (2088 rows x 4096 x 257: 7.1 - 7.3 against 7.4 - 7.7, 2560 rows 8.1 - 8.5 against 10.3).  Until then: the row
    // walk below 2560 rows.
    int hp_mode = H < 1024 ? (W >= 32 && codes_ok && !codes_never ? 3 : 2) : (H < 2048 ? 2 : 1);
    if (ehp && ehp[0] >= '1' && ehp[0] <= '3') hp_mode = ehp[0] - '0';
    if (hp_mode == 3 && (!codes_ok || W < 32)) hp_mode = 2;
    const bool two_sided = hp_mode == 2;
    if (fam && hp_mode == 3) {  // the row walk stores 8 bytes per lane: a pixel stride of a multiple of 8 (at most 4 bytes of pad)
        Dp = (Dp + 7) & ~7;
        vol = pmx_dir_stride(H, W, Dp);
    }
    const bool hp_codes = fam && (hp_mode == 3 || (hp_mode == 1 && codes_ok && codes_always));
    // (the marching kernel from the words: until round 6 the default beside the row walk below 1536 rows; with round 6's cost kernel the
    //  cost volume is 3 - 6 % ahead on every short shape measured - 480 / 592 / 800 rows x 4096 x 257: 2.18 / 2.56 / 3.15 against 2.30 /
    //  2.64 / 3.30 ms, 1000 x 6000 x 193: 3.94 against 4.19 - so the words are taken on request only)
    bool fam_codes = fam && codes_ok && !codes_never && codes_always;
    if (const char* efc = pmx_opt(ctx, "SGM8_FAMCODES")) fam_codes = fam && codes_ok && efc[0] == '1';  // (A/B hook: the marching kernel alone)
    const bool from_codes = hp_codes && fam_codes;  // no cost volume at all
    if (!from_codes && cv->cost8_bytes < cvol) {
        pmx_pool_free(ctx, cv->cost8);
        cv->cost8 = nullptr;
        cv->cost8_bytes = 0;
        PMX_HIP(pmx_pool_alloc(ctx, (void**)&cv->cost8, cvol + 64));
        cv->cost8_bytes = cvol;
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
    auto launch_cost = [&](hipStream_t cst) -> int {
        pmx_stage_scope t(ctx, PMX_STAGE_CENSUS_COST, cst);
        cost8_args c;
        c.codeL = cv->codeL; c.codeR = cv->codeR; c.cost = cv->cost8;
        c.range = cv->has_range ? cv->range : nullptr;
        c.H = H; c.W = W; c.D = cv->D; c.Dp = Dc; c.d0 = cv->d0; c.o = cv->win / 2;
        c.invalid_cost = invalid_cost;
        c.nact = nact;
        const size_t want = ((size_t)H * W * nact + 256 * kUnitSteps - 1) / (256 * kUnitSteps);  // a unit per lane and step
        PMX_CHECK(want < (1u << 31), PMX_ERR_ARG, "census costs: %d x %d x %d units are more than one launch takes", H, W, nact);
        const dim3 grid((unsigned)want);
#define PMX_COST8(NWV, KPLV)                                                                                                  \
    if (five && NWV == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(census_cost_u8_kernel<NWV, KPLV, (NWV == 1 ? 5 : 8)>), grid, dim3(256), 0, cst, c); \
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(census_cost_u8_kernel<NWV, KPLV, 8>), grid, dim3(256), 0, cst, c)
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
        PMX_HIP(hipGetLastError());
        return PMX_OK;
    };
    if (!from_codes) {
        const int rcc = launch_cost(ctx->stream);
        if (rcc) return rcc;
    }
    sgm8_args a;
    a.cost = cv->cost8; a.ldir = cv->ldir; a.dstride = cv->dstride;
    a.H = H; a.W = W; a.D = cv->D; a.Dp = Dp; a.Dc = Dc; a.P1 = P1; a.P2 = P2;
    if (fam) {
        // The horizontal pair has one wavefront per four rows (1024 at 4096 rows: one per SIMD, latency-bound on its own) and is
        // independent of the vertical families: it runs on the context's second stream, beside the marching kernel
        // (PMX_SGM8_OVERLAP=0 keeps everything in line: A/B hook).
        const char* eo = pmx_opt(ctx, "SGM8_OVERLAP");
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
            const int ngroups = (H + kLines8 - 1) / kLines8;
            const dim3 hgrid(two_sided ? (ngroups + 1) / 2 : (ngroups + kWaves8 - 1) / kWaves8), hblock(kWaves8 * 64);
#define PMX_HP(KPLV)                                                                                                       \
    if (two_sided && five) hipLaunchKernelGGL(HIP_KERNEL_NAME(sgm_u8_hpair2_kernel<KPLV, 5>), hgrid, hblock, 0, hs, a);    \
    else if (two_sided) hipLaunchKernelGGL(HIP_KERNEL_NAME(sgm_u8_hpair2_kernel<KPLV, 8>), hgrid, hblock, 0, hs, a);       \
    else if (five) hipLaunchKernelGGL(HIP_KERNEL_NAME(sgm_u8_hpair_kernel<KPLV, 5>), hgrid, hblock, 0, hs, a);             \
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(sgm_u8_hpair_kernel<KPLV, 8>), hgrid, hblock, 0, hs, a)
            if (hp_codes) {
                hpc_args h;
                h.codeL = cv->codeL; h.codeR = cv->codeR; h.ldir = cv->ldir;
                h.H = H; h.W = W; h.D = cv->D; h.Dp = Dp; h.d0 = cv->d0; h.o = cv->win / 2;
                h.P1 = P1; h.P2 = P2; h.invalid_cost = invalid_cost;
#define PMX_HPC(KPLV) hipLaunchKernelGGL(HIP_KERNEL_NAME(sgm_u8_hpair_codes_kernel<KPLV>), hgrid, hblock, 0, hs, h)
                if (hp_mode == 3) {
                    const dim3 rgrid(H), rblock(128);  // one row per workgroup: a wavefront from each end
                    if (cv->D <= 256) hipLaunchKernelGGL(HIP_KERNEL_NAME(sgm_u8_hrow_codes_kernel<4, false>), rgrid, rblock, 0, hs, h);
                    else if (cv->D == 257) hipLaunchKernelGGL(HIP_KERNEL_NAME(sgm_u8_hrow_codes_kernel<4, true>), rgrid, rblock, 0, hs, h);
                    else hipLaunchKernelGGL(HIP_KERNEL_NAME(sgm_u8_hrow_codes_kernel<8, false>), rgrid, rblock, 0, hs, h);
                } else
                switch (kpl) {
                    case 4: PMX_HPC(4); break;
                    case 8: PMX_HPC(8); break;
                    case 12: PMX_HPC(12); break;
                    case 16: PMX_HPC(16); break;
                    default: PMX_HPC(20); break;
                }
#undef PMX_HPC
            } else
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
        const int rcf = pmx_launch_sgm_fam8(ctx, cv, kpl, five, Dc, cv->ldir + vol, vol, P1, P2, 3, fam_codes, invalid_cost);
        if (overlap) PMX_HIP(hipStreamWaitEvent(ctx->stream, ctx->aux_join, 0));  // whoever reads the volumes next waits for both
        return rcf;
    }
    const int nwaves = 2 * ((H + kLines8 - 1) / kLines8) + 6 * ((W + kLines8 - 1) / kLines8);
    const dim3 grid((nwaves + kWaves8 - 1) / kWaves8), block(kWaves8 * 64);
    {
        pmx_stage_scope t(ctx, PMX_STAGE_SGM_FUSED);
        // PMX_SGM8_HF=0: the minima on the integer pipe (v_pk_min_u16 chains), the round-2 form - an A/B hook
        const char* ehf = pmx_opt(ctx, "SGM8_HF");
        const bool hf = !(ehf && ehf[0] == '0');
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
