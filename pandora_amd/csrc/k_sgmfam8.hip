// k_sgmfam8.hip - the integer fast path's SGM as DIRECTION FAMILIES on packed 16-bit arithmetic.  gfx950.
//
// The one-launch-per-eight-paths kernel of k_sgm8.hip writes eight byte volumes and the WTA reads them back: 16 of the step's
// ~22 B/cell are path volumes, and both kernels run at the speed of that traffic (profiles/r02_e_northstar_pmc_hbm.csv;
// taking 12 % of the path kernel's instructions out moved nothing, profiles/r03_a_*).  The three paths that advance one image
// row per step - (+1,0) (+1,+1) (+1,-1), or their mirror images - can be summed IN REGISTERS when one kernel marches down the
// rows computing all three at each pixel: one byte per cell and family leaves the chip instead of three (every L_r <=
// invalid_cost + P2, so a family's sum fits a byte whenever 3 * (invalid_cost + P2) <= 255: census 5x5 with P2 = 32 gives 174).
// The horizontal pair is summed in k_sgm8.hip (the backward pass adds into the forward pass's volume).  Volumes: 8 -> 3.
//
// Decomposition: the float32 family kernel's (k_sgmfam.hip), restated for two disparities per 32-bit register.  Rows are
// sequential, columns are the parallel axis; a workgroup owns a window of CW columns that SLIDES LEFT by one column per row,
//     column(r, j) = base - r + j,   j = 0 .. CW-1,   base = s * CW,
// so that the (+1,-1) path stays in its lane group (registers), the vertical path comes from local column j-1 and the (+1,+1)
// path from j-2: every cross-window dependency points to the LEFT neighbour, workgroups form a one-directional pipeline, and the
// hand-off latency is paid once as pipeline lag.  Window indices come from the tickets of pmx_buf.h (a window's left neighbour has
// always been taken before it: no co-residency assumption); both families run in ONE launch, so that a CU
// holds waves of both and the chip sees 2 x W columns of parallel work.  Inside a workgroup the two shifting paths go through
// LDS as they are (packed u16 pairs, double-buffered by row parity, one barrier per row); the left neighbour's last two columns
// arrive through global memory as BYTES (every L_r < 256) in 16-byte blocks {value, tag, value, tag}, tag = the launch's scrambled
// count (pmx_fam_tag), each 8-byte half self-validating (cdna_hip_programming.md Guideline 16, form R2), read by a dedicated
// wave with bounded polling.  (Round 4 tried the float32 kernels' blocks of three values and a checked tag here: 82 instead of 122
// blocks per row and border, but three (A, B) register pairs per block instead of two is six instead of four LDS instructions per
// row in the hand-off wavefront, which every row's barrier waits for: the marching kernel alone 7.3 -> 9.0 ms.  Out again.)  Same lane map as the path kernel and the cost kernel: 16 lanes per pixel, KPL = 4 Q consecutive
// disparities per lane as A[q] = (L[4q], L[4q+2]), B[q] = (L[4q+1], L[4q+3]); 4 pixels per wave.
//
// Arithmetic: minima on the f16 pipe (positive halves order like integers; v_pk_minimum3_f16), additions as plain 32-bit adds of
// packed pairs (no half overflows), "- M + C" in one v_add3_u32 - see k_sgm8.hip.  Semantics = oracle.c orc_sgm on integers:
// L = C + min(Lp[d], min(Lp[d-1], Lp[d+1]) + P1, M + P2) - M, paths start from (Lp, M) = (0, 0) at the image border.
#include <cstdlib>
#include <type_traits>

#include "pmx_buf.h"
#include "pmx_internal.h"

namespace {

typedef __attribute__((address_space(1))) unsigned int gu32;

constexpr int kSc1 = 16;                    // aux bit of the buffer instructions: write-through store / L1-bypassing load
constexpr uint32_t kPad16 = 0x7000u;        // what a padded disparity (d >= D) carries: beyond every real value, not an f16 NaN
constexpr uint32_t kPadPk = 0x70007000u;
constexpr unsigned kSpinLimit = 1u << 21;   // polls before a hand-off gives up

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
__device__ __forceinline__ uint32_t add3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t d;
    asm("v_add3_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
template <int CTRL>
__device__ __forceinline__ uint32_t dppo(uint32_t oldv, uint32_t src) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)oldv, (int)src, CTRL, 0xf, 0xf, false);
}

struct fam8_args {
    const uint8_t* cost;  // [H][W][Dc]: a pixel's costs lane by lane (k_sgm8.hip census_cost_u8_kernel)
    uint8_t* out;         // family f's sums at out + f * dstride, [H][W][Dp] bytes in lane-map order
    size_t dstride;
    int H, W, D, Dp, Dc;
    uint32_t P1, P2;
    u32x4* halo;          // hand-off blocks [family][H][NB][NGP]
    size_t halo_fam;      // blocks per family
    int NB;               // window borders per row = ceil(W / CW)
    unsigned epoch;
    unsigned* ctl;        // [1] error word
    // Window tickets (pmx_buf.h pmx_take_window, as in k_sgmfam.hip): chunks of G consecutive windows of a family belong to XCD
    // (chunk mod 8); a workgroup prefers its own XCD's chunks and never takes a window whose left neighbour is not taken; a publisher
    // whose reader is known to sit on its own XCD writes plain stores instead of write-through ones.
    // xtab: [f * nchunk + c] windows taken of chunk c of family fam0 + f; started[f * nwin + w] = 1 + the XCD of window w (zeroed per launch).
    unsigned* xtab;
    unsigned* started;
    int G, nwin, nchunk;
    int fam0, nfam;       // families of this launch: fam0, fam0 + 1, ... (0 = downward, 1 = upward)
    int prio;             // wave priority of this launch's wavefronts (beside the horizontal-pair kernel on the second stream)
    // CODES form: no cost volume - the Hamming costs are computed here from the census words (one per pixel)
    const uint32_t* codes;  // start of the code allocation: [guard | left image | guard | right image | guard], zeroed guards
    unsigned code_bytes;    // ... its size
    unsigned codeL_off, codeR_off;  // dword index of pixel (0, 0) of the left / right code image in it
    int d0, o;              // first disparity, half census window (cells whose windows leave an image carry invalid_cost)
    uint32_t invalid_cost;
};

// One path, one pixel per 16-lane row: new path costs (nA, nB) from the predecessor's (A, B, M); returns the packed minimum of the
// lane's new costs (both halves still to be reduced).
template <int Q>
__device__ __forceinline__ uint32_t path_update(const uint32_t (&A)[Q], const uint32_t (&B)[Q], uint32_t M, const uint32_t (&ccA)[Q],
                                                const uint32_t (&ccB)[Q], uint32_t P1pk, uint32_t P2pk, uint32_t (&nA)[Q], uint32_t (&nB)[Q],
                                                uint32_t (&edge)[2]) {
    // row_shr:1 - the previous lane's (.., L[d_first - 1]); row_shl:1 - the next lane's (L[d_first + KPL], ..).  Lane 0 of a
    // pixel's 16 is never written by the first shift, lane 15 never by the second: `edge` (kPadPk before the first step, the
    // caller's for the kernel's life) keeps their +inf, and no constant is loaded per step
    const uint32_t belowB = edge[0] = dppo<0x111>(edge[0], B[Q - 1]);
    const uint32_t aboveA = edge[1] = dppo<0x101>(edge[1], A[0]);
    const uint32_t mp2 = M + P2pk;
    const uint32_t negM = 0u - M;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        // neighbours: A = (d, d+2) has lo = (d-1, d+1), hi = (d+1, d+3) = B;  B has lo = A, hi = (d+2, d+4)
        const uint32_t loA = __builtin_amdgcn_alignbit(B[q], q > 0 ? B[q > 0 ? q - 1 : 0] : belowB, 16);
        const uint32_t hiB = __builtin_amdgcn_alignbit(q < Q - 1 ? A[q < Q - 1 ? q + 1 : 0] : aboveA, A[q], 16);
        const uint32_t tA = hmin3(A[q], hmin(loA, B[q]) + P1pk, mp2);
        const uint32_t tB = hmin3(B[q], hmin(A[q], hiB) + P1pk, mp2);
        nA[q] = add3(tA, ccA[q], negM);
        nB[q] = add3(tB, ccB[q], negM);
    }
    uint32_t m = hmin(nA[0], nB[0]);
#pragma unroll
    for (int q = 1; q < Q; ++q) m = hmin3(m, nA[q], nB[q]);
    return m;
}

// minima of three paths over the 16 lanes of each pixel, every lane receives them as (m | m << 16)
__device__ __forceinline__ void group_min3(uint32_t& a, uint32_t& b, uint32_t& c) {
    auto halves = [](uint32_t m) { const uint32_t lo = m & 0xffffu, hi = m >> 16; return lo < hi ? lo : hi; };
    a = halves(a); b = halves(b); c = halves(c);
    auto mn = [](uint32_t x, uint32_t y) { return x < y ? x : y; };
    a = mn(a, dppo<0x128>(0xffffffffu, a)); b = mn(b, dppo<0x128>(0xffffffffu, b)); c = mn(c, dppo<0x128>(0xffffffffu, c));
    a = mn(a, dppo<0x124>(0xffffffffu, a)); b = mn(b, dppo<0x124>(0xffffffffu, b)); c = mn(c, dppo<0x124>(0xffffffffu, c));
    a = mn(a, dppo<0x122>(0xffffffffu, a)); b = mn(b, dppo<0x122>(0xffffffffu, b)); c = mn(c, dppo<0x122>(0xffffffffu, c));
    a = mn(a, dppo<0x121>(0xffffffffu, a)); b = mn(b, dppo<0x121>(0xffffffffu, b)); c = mn(c, dppo<0x121>(0xffffffffu, c));
    a |= a << 16; b |= b << 16; c |= c << 16;
}

// CODES: the costs of a row are not read from a volume but made from the census words of the row (census.cpp:132-172: Hamming
// distance of the two bit strings; cells whose window leaves the left or the right image stay NaN there = invalid_cost here).
// The compute wavefronts fetch the window's words two rows ahead (one dword per thread: CW left words, CW + 16 KPL + 3 right
// words - the window's pixels reach disparities d0 .. d0 + 16 KPL - 1), park them in LDS one row ahead - the right words four
// times, copy g shifted by g words, so that the 16-byte reads of a lane of pixel group g are aligned (an unaligned ds_read_b128
// is replayed at 64 cycles) - and every lane reads its KPL words and the pixel's left word where the old form read NDW cost
// dwords from memory: v_xor, v_bcnt (whose addend is the pad of a disparity >= D), v_lshl_or per pair of cells.
template <int KPL, int CBITS, int NW, int PF, bool CODES>
__global__ __launch_bounds__((NW + 2) * 64) void sgm_fam8_kernel(fam8_args a) {
    constexpr int Q = KPL / 4;                  // (A, B) register pairs per lane and path
    constexpr int NR = 2 * Q;                   // registers per lane and path
    constexpr int PER = CBITS == 8 ? 4 : 6;     // costs per dword of the cost volume
    constexpr int NDW = (KPL + PER - 1) / PER;  // cost dwords per lane
    constexpr int CW = NW * 4;                  // columns per window
    constexpr int KS = (NR + 3) & ~3;           // LDS dwords per lane slice: A0 B0 A1 B1 ... (16-byte aligned)
    constexpr int ES = 16 * KS + 4;             // LDS dwords per (path, column): 16 slices + the minimum
    constexpr int EDIR = (CW + 2) * ES;         // one path: column slots -2 .. CW-1
    constexpr int EBUF = 2 * EDIR;              // one row parity: vertical path, diagonal path
    constexpr int NVB = 8 * Q;                  // hand-off blocks per vector: 16 Q dwords of packed bytes, two per block
    constexpr int NG = 3 * NVB + 2;             // blocks per (row, border): V[CW-1], A[CW-1], A[CW-2], then their three minima
    constexpr int NQ = (NG + 63) / 64;
    constexpr int NGP = NQ * 64;
    constexpr int NT = NW * 64;                 // compute threads
    constexpr int CS = CW + 16 * KPL;           // CODES: right words per copy (a lane of column j reads words j + sub KPL ...)
    constexpr int NSRC = CS + 3;                // ... distinct right words of a row (copy g starts g words further)
    constexpr int NLOADS = NSRC + CW;           // ... + the left words
    constexpr int NLD = (NLOADS + NT - 1) / NT; // ... dwords fetched per thread and row
    constexpr int CPB = 4 * CS + CW;            // ... LDS dwords per row parity
    static_assert(KPL % 4 == 0 && KPL >= 4 && KPL <= 20, "whole dwords per lane");
    extern __shared__ __attribute__((aligned(16))) uint32_t lds8[];
    typedef __attribute__((address_space(3))) int lds_int;
    volatile lds_int* ctl = (volatile lds_int*)(lds_int*)(lds8 + 2 * EBUF);  // [0] ticket, [1], [2] abort flag by row parity
    uint32_t* const cbuf = lds8 + 2 * EBUF + 4;  // CODES: [row parity][4 copies of CS right words | CW left words]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // the marching pipeline is one long dependency chain (every row waits for the row before, every window for its neighbour):
    // its wavefronts go first, whatever shares the SIMD with them (the horizontal pair on the second stream) fills the gaps
    if (a.prio == 1) __builtin_amdgcn_s_setprio(1);
    else if (a.prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (a.prio == 3) __builtin_amdgcn_s_setprio(3);
    if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7u;
        // a window of this XCD's chunks (of either family, the workgroups of an XCD starting with them in turn) if one may be taken,
        // after a while any window that may: placement is for speed only, every window is taken exactly once and never ahead of its
        // left neighbour
        pmx_win_tickets tq;
        tq.cnt = a.xtab; tq.G = a.G; tq.nwin = a.nwin; tq.nchunk = a.nchunk; tq.nfam = a.nfam;
        const int got = pmx_take_window(tq, xcc, (blockIdx.x >> 3) & 1u);
        int tk = -1;
        if (got >= 0) {
            __hip_atomic_store(a.started + got, xcc + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tk = (got % a.nwin) * a.nfam + got / a.nwin;  // (window * nfam + family, as the rest of the kernel reads it)
        }
        ctl[0] = tk;
        ctl[1] = 0;
        ctl[2] = 0;
        ctl[3] = (int)xcc;
    }
    __syncthreads();
    const int ticket = __builtin_amdgcn_readfirstlane(ctl[0]);
    if (ticket < 0) return;  // (the launch is rounded up so that every XCD's share covers its sequence)
    const unsigned my_xcc = (unsigned)__builtin_amdgcn_readfirstlane(ctl[3]);
    const int fam = a.fam0 + ticket % a.nfam;  // 0: rows top -> bottom, paths (+1,0) (+1,+1) (+1,-1);  1: bottom -> top, mirrored
    const int s = ticket / a.nfam;
    const int H = a.H, W = a.W, D = a.D;
    const int base = s * CW;
    const int r_lo = base - W + 1 > 0 ? base - W + 1 : 0;
    const int r_hi = base + CW - 1 < H - 1 ? base + CW - 1 : H - 1;
    if (r_lo > r_hi) return;
    gu32* errw = (gu32*)(a.ctl + 1);
    u32x4* const halo = a.halo + (size_t)(fam - a.fam0) * a.halo_fam;
    constexpr unsigned kBlockBytes = (unsigned)NGP * 16u;

    if (wave >= NW) {
        // ---- hand-off wavefronts: wave NW publishes this window's last two columns for window s+1, wave NW+1 brings the left
        // neighbour's into column slots -2, -1 ---------------------------------------------------------------------------------------
        // A SIMD issues about one instruction per six cycles here, of whatever kind and from whichever of its wavefronts (round 5:
        // 6.1 alone, 6.6 with the horizontal pair beside it): one hand-off wavefront
        // of ~185 instructions per row (110 of them scalar: three descriptors rebuilt per row from 64-bit products) sat on the SIMD
        // of compute wavefronts 0 and 4 and was a fifth of its load.  Now two wavefronts (they land on different SIMDs), records
        // addressed by running pointers, and no lane-varying branch: every lane moves two (A, B) register pairs and two minima per
        // block - the kinds its block does not hold go to a spare dword pair of the column slot.
        const bool publisher = wave == NW;
        constexpr int SINK = 16 * KS + 2;  // spare dwords of a column slot (its minimum sits at 16 KS): 8-byte aligned
        int offP0[NQ], offP1[NQ];     // LDS dword offsets (within a row parity) of the block's two register pairs
        int offM0[NQ], offM1[NQ];     // ... of its two minima
        uint32_t pA0[NQ], pB0[NQ], pA1[NQ], pB1[NQ];  // pad masks of the two packed dwords
        bool is_min[NQ], one_min[NQ]; // the block holds minima; ... one of them (twice)
        unsigned boff[NQ];            // byte offset of the block in a (row, border) record: lanes without a block touch nothing
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int idx = q * 64 + lane;
            offP0[q] = offP1[q] = SINK; offM0[q] = offM1[q] = SINK;
            pA0[q] = pB0[q] = pA1[q] = pB1[q] = 0;
            is_min[q] = one_min[q] = false;
            boff[q] = idx < NG ? (unsigned)idx * 16u : kOob;
            if (idx < 3 * NVB) {
                const int vec = idx / NVB, rem = idx - vec * NVB;
                // vec 0: vertical path of column CW-1 -> slot -1;  1: diagonal of CW-1 -> slot -1;  2: diagonal of CW-2 -> slot -2
                const int slot = (vec == 0 ? 0 : EDIR) + (vec == 2 ? 0 : ES);
                auto place = [&](int m, int& off, uint32_t& pa, uint32_t& pb) {
                    const int l = m / Q, qq = m - l * Q, d = l * KPL + 4 * qq;
                    off = slot + l * KS + 2 * qq;
                    pa = ((d < D) ? 0u : kPad16) | (((d + 2 < D) ? 0u : kPad16) << 16);
                    pb = ((d + 1 < D) ? 0u : kPad16) | (((d + 3 < D) ? 0u : kPad16) << 16);
                };
                place(2 * rem, offP0[q], pA0[q], pB0[q]);
                place(2 * rem + 1, offP1[q], pA1[q], pB1[q]);
            } else if (idx == 3 * NVB) {      // minima of slot -1: vertical path, diagonal
                is_min[q] = true; offM0[q] = ES + 16 * KS; offM1[q] = EDIR + ES + 16 * KS;
            } else if (idx == 3 * NVB + 1) {  // minimum of slot -2: diagonal
                is_min[q] = one_min[q] = true; offM0[q] = EDIR + 16 * KS; offM1[q] = offM0[q];
            }
        }
        constexpr int QM0 = (3 * NVB) / 64, QM1 = (3 * NVB + 1) / 64;  // the blocks that hold minima
        // The record the neighbour wrote for row t sits at block row bi(t) = t NB + (base - t - 1) / CW = t NB + s - 1 - t / CW
        // (t >= 0), the one this window writes for row t one further (column block (base + CW - 1 - t) / CW).  From row to row:
        // bi(t + 1) = bi(t) + NB, one less when t + 1 is a multiple of CW - a running pointer and a counter.
        auto rec_of = [&](int t) { const int tt = t > 0 ? t : 0; return halo + ((size_t)tt * a.NB + s - 1 - tt / CW) * NGP; };
        const size_t row_blocks = (size_t)a.NB * NGP;
        auto rsrc_of = [&](const u32x4* rec, bool need) {
            return __builtin_amdgcn_make_buffer_rsrc((void*)rec, 0, need ? kBlockBytes : 0u, kRsrcWord3);
        };
        if (publisher) {
            // Row p of this window - the path costs of local columns CW-2, CW-1, left by the compute wavefronts in column slots CW,
            // CW+1 of row parity p & 1 - is complete once barrier p is passed and untouched until barrier p+1.  Window s+1 computes
            // row p+1 at image column cb = base+CW-1-p: it exists and needs the row iff cb < W and p < H-1.
            const int pA = r_lo > base + CW - W ? r_lo : base + CW - W;
            const int pB = r_hi < H - 2 ? r_hi : H - 2;
            const u32x4* rec = rec_of(r_lo) + NGP;
            int pmod = r_lo % CW;
            // Window s + 1 reads what is published here.  Once it has said (xtab) that it runs on THIS XCD - by construction G - 1
            // of G neighbours do - the blocks go out as plain stores: they stay in the XCD's L2, where the reader's L1-bypassing
            // loads find them, and cost a fraction of a write-through store (k_sgmfam.hip, round 6: 51.7 -> 43.8 ms per
            // 4096^2 x 257 float32 step).  Until then, and for a reader elsewhere: sc1, which every reader sees.  A block is taken by
            // its tags whichever way it was written.
            bool peer_known = (s + 1) % a.G == 0 || s + 1 >= a.nwin, peer_local = false;
            unsigned peer_probe = 0;  // the reader's entry as read one row ago
            const unsigned* peer_entry = a.started + (size_t)(fam - a.fam0) * a.nwin + (s + 1 < a.nwin ? s + 1 : s);
            __syncthreads();  // the barrier before the first step
            for (int pr = r_lo; pr <= r_hi; ++pr) {
                __syncthreads();  // barrier pr
                if (!peer_known) {
                    if (peer_probe != 0u) {
                        peer_known = true;
                        peer_local = __builtin_amdgcn_readfirstlane(peer_probe) == my_xcc + 1u;
                    } else {
                        peer_probe = __hip_atomic_load(peer_entry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                const __amdgpu_buffer_rsrc_t rs = rsrc_of(rec, pr >= pA && pr <= pB);
                const uint32_t* Eb = lds8 + (pr & 1) * EBUF + CW * ES;
                const uint32_t gave_up = (uint32_t)ctl[1];
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const u32x2 v0 = *(const u32x2*)(Eb + offP0[q]);
                    const u32x2 v1 = *(const u32x2*)(Eb + offP1[q]);
                    u32x4 b;
                    b.x = v0.x | (v0.y << 8);  // (pads spill upwards only: into bytes that are pads themselves)
                    b.z = v1.x | (v1.y << 8);
                    if (q == QM0 || q == QM1) {  // (compile time)
                        const uint32_t m0 = Eb[offM0[q]], m1 = Eb[offM1[q]];
                        b.x = is_min[q] ? m0 : b.x;
                        b.z = is_min[q] ? m1 : b.z;
                    }
                    b.y = a.epoch;
                    b.w = a.epoch;
                    if (peer_local) __builtin_amdgcn_raw_buffer_store_b128(b, rs, boff[q], 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b128(b, rs, boff[q], 0, kSc1);
                }
                if (__builtin_amdgcn_readfirstlane(gave_up) != 0u) return;  // (the consumer's: this launch has failed)
                rec += row_blocks;
                if (++pmod == CW) { pmod = 0; rec -= NGP; }
            }
            return;
        }
        // Rows tA .. tB of the neighbour are needed (row t feeds this window's row t+1: 1 <= t+1 <= base, r_lo <= t+1 <= r_hi).
        const int tA = r_lo - 1 > 0 ? r_lo - 1 : 0;
        const int tB = (r_hi < base ? r_hi : base) - 1;
        u32x4 x[NQ];
        auto issue = [&](int t, const u32x4* rec) {
            const __amdgpu_buffer_rsrc_t rs = rsrc_of(rec, t >= tA && t <= tB);
#pragma unroll
            for (int q = 0; q < NQ; ++q) x[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, boff[q], 0, kSc1);
        };
        auto consume = [&](int t, const u32x4* rec) -> bool {
            if (t < tA || t > tB) return true;
            const __amdgpu_buffer_rsrc_t rs = rsrc_of(rec, true);
            for (unsigned spins = 0;; ++spins) {
                bool ok = true;
#pragma unroll
                for (int q = 0; q < NQ; ++q) ok &= boff[q] == kOob || (x[q].y == a.epoch && x[q].w == a.epoch);
                if (__all(ok)) break;
                if ((spins & 31) == 31 && __hip_atomic_load(errw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
                if (spins > kSpinLimit) {
                    if (lane == 0) __hip_atomic_store(errw, 0x80000000u + (unsigned)ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    return false;
                }
                __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int q = 0; q < NQ; ++q) x[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, boff[q], 0, kSc1);
            }
            uint32_t* Eb = lds8 + (t & 1) * EBUF;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                u32x2 v0, v1;  // bytes d, d+1, d+2, d+3 -> A = (d, d+2), B = (d+1, d+3), pads restored
                v0.x = (x[q].x & 0x00ff00ffu) | pA0[q]; v0.y = ((x[q].x >> 8) & 0x00ff00ffu) | pB0[q];
                v1.x = (x[q].z & 0x00ff00ffu) | pA1[q]; v1.y = ((x[q].z >> 8) & 0x00ff00ffu) | pB1[q];
                *(u32x2*)(Eb + offP0[q]) = v0;
                *(u32x2*)(Eb + offP1[q]) = v1;
                if (q == QM0 || q == QM1) {  // (compile time)
                    Eb[offM0[q]] = x[q].x;
                    Eb[offM1[q]] = one_min[q] ? x[q].x : x[q].z;
                }
            }
            return true;
        };
        // barrier index t runs from r_lo-1 (the barrier before the first step) to r_hi.  Row t of the neighbour is asked for as
        // early as it can exist (one look-ahead load per barrier), and polled for when it is due.  (Cycle stamps show this wavefront
        // waiting 2500 of a row's 4250 cycles for that load - but the wait is slack: a window cannot run ahead of its left neighbour,
        // the chain moves at the pace of its head, and the head's row is what two compute wavefronts per SIMD need to issue.  Three
        // rows of look-ahead, alternated on one box: 6.90 / 6.98 / 7.16 ms alone against 6.89 / 6.76 / 6.61 with one; two publishers
        // and two consumers with half the blocks each (commit 1d1420f): 6.79 / 6.79 / 7.10 against 6.44 / 6.87 / 6.76; and the whole
        // march without a barrier per row - flags between the wavefronts, rings of four rows, commit a538c47: 6.6 - 7.1 against
        // 6.3 - 6.7.  DESIGN 7.27.)
        int t = r_lo - 1;
        const u32x4* rec_cur = rec_of(t);
        const u32x4* rec_next = rec_of(t + 1);
        int tmod = (t + 1 > 0 ? t + 1 : 0) % CW;  // (t + 1) mod CW: the block index loses one when t + 2 reaches a multiple of CW
        issue(t, rec_cur);
        for (; t <= r_hi; ++t) {
            const bool got = consume(t, rec_cur);
            if (!got) ctl[1] = 1;  // (sticky: the other wavefronts read it with their next row's LDS reads)
            issue(t + 1, rec_next);
            __syncthreads();
            if (!got) return;
            // bi(t + 2) from bi(t + 1): one row further, one column block back at every multiple of CW
            rec_cur = rec_next;
            if (t + 1 >= 0) {
                ++tmod;
                rec_next += row_blocks;
                if (tmod == CW) { tmod = 0; rec_next -= NGP; }
            }
        }
        return;
    }

    // ---- compute waves ---------------------------------------------------------------------------------------------------
    const int g = lane >> 4, sub = lane & 15;
    const int j = wave * 4 + g;
    const int d_first = sub * KPL;
    const bool lane_active = d_first < D;
    const unsigned cost_lane = lane_active ? (unsigned)sub * NDW * 4u : kOob;  // lanes without a disparity read zeros
    const unsigned out_lane = (unsigned)sub * KPL;
    const unsigned cost_row_bytes = (unsigned)W * (unsigned)a.Dc, out_row_bytes = (unsigned)W * (unsigned)a.Dp;
    uint8_t* const outv = a.out + (size_t)fam * a.dstride;

    uint32_t padA[Q], padB[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int d = d_first + 4 * q;
        padA[q] = ((d < D) ? 0u : kPad16) | (((d + 2 < D) ? 0u : kPad16) << 16);
        padB[q] = ((d + 1 < D) ? 0u : kPad16) | (((d + 3 < D) ? 0u : kPad16) << 16);
    }
    const uint32_t P1pk = a.P1 | (a.P1 << 16), P2pk = a.P2 | (a.P2 << 16);

    // Row pointers and lane offsets advance with the march (one row, one column to the left per step) instead of being rebuilt:
    // the scalar unit's multiplications for a 64-bit row address were a fifth of this kernel's instructions.
    const ptrdiff_t cost_step = fam ? -(ptrdiff_t)cost_row_bytes : (ptrdiff_t)cost_row_bytes;
    const ptrdiff_t out_step = fam ? -(ptrdiff_t)out_row_bytes : (ptrdiff_t)out_row_bytes;
    const int rimg_lo = fam ? H - 1 - r_lo : r_lo;
    const uint8_t* cost_row = a.cost + (size_t)rimg_lo * cost_row_bytes;   // row `pr` of the prefetch
    uint8_t* out_row = outv + (size_t)rimg_lo * out_row_bytes;             // row `r` of the step
    int pc = base - r_lo + j;                                              // column of the prefetched row
    unsigned pcoff = (unsigned)pc * (unsigned)a.Dc + cost_lane;            // (wraps for negative columns: masked below)

    // loads of the costs run PF rows ahead in a register ring
    int pr = r_lo;
    // (costs from the volume: the ring is hring<slot> - registers above the compiler's, waited for with the ring's own count, pmx_buf.h;
    //  CODES: the census words' ring stays in compiler-managed registers, its takers are LDS stores one row later)
    struct slot_t { uint32_t x[CODES ? NLD : 1]; };
    slot_t ring[PF];
    constexpr int kRingCnt = (PF - 1) * (hring_loads(NDW) + (Q == 5 ? 2 : 1));  // memory instructions of PF - 1 steps
    // (slot S: v[96 + 6 S ..]: 114 registers end the allocation at 120 - what a workgroup's wavefronts hold decides who else fits
    //  on their SIMDs: with the ring at v128.. the marching kernel took 152 registers and the horizontal pair beside it no longer
    //  found room, 22.5 instead of 8.4 ms.  The compiler's own: 50 .. 90; `make` runs tools/check_hring.py k_sgmfam8.o 96)
    constexpr int kRing0 = 96;
    static_assert(PF <= 3 && NDW <= 6, "hring slots");
    if constexpr (!CODES) PMX_HRING_RESERVE("v113");
    // CODES: thread tid fetches word m = tid + i NT of the row's list: right words of image columns cb + d0 + m (m < NSRC),
    // then the left words of columns cb + m - NSRC, cb = column of the window's first pixel.  One descriptor over the whole
    // code allocation: a column outside the image reads a neighbouring row's words or a zeroed guard (such cells are invalid
    // and never look at their words), an offset outside the allocation reads 0.
    const int tid = wave * 64 + lane;
    const __amdgpu_buffer_rsrc_t crs = __builtin_amdgcn_make_buffer_rsrc((void*)a.codes, 0, a.code_bytes, kRsrcWord3);
    int coff[NLD];  // dword index of this thread's words in the row `pr`, relative to the row's first pixel
    if constexpr (CODES) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int m = tid + i * NT;
            coff[i] = m < NSRC ? (int)a.codeR_off + a.d0 + m : (m < NLOADS ? (int)a.codeL_off + m - NSRC : -(1 << 28));
        }
    }
    int code_row = rimg_lo * W + (base - r_lo);  // dword index of the prefetched row's window start, within an image
    auto prefetch = [&](auto slot_tag, slot_t& sl) __attribute__((always_inline)) {
        constexpr int SL = decltype(slot_tag)::value;
        if constexpr (CODES) {
#pragma unroll
            for (int i = 0; i < NLD; ++i)
                sl.x[i] = __builtin_amdgcn_raw_buffer_load_b32(crs, (unsigned)(code_row + coff[i]) * 4u, 0, 0);
            if (pr < r_hi) {
                ++pr;
                code_row += (fam ? -W : W) - 1;
            }
        } else {
            hring_load<kRing0 + 6 * SL, NDW>(rsrc_words(cost_row, cost_row_bytes), ((unsigned)pc < (unsigned)W && lane_active) ? pcoff : kOob);
            // (uniform selects, no branch: past the last row the last one is read again.  A branch here, or between the steps
            // of the unrolled loop, splits the body into blocks whose wait counts the compiler derives from merged states -
            // vmcnt(0) in one step of three, i.e. a wait for the stores just issued: DESIGN 7.27)
            const bool adv = pr < r_hi;
            pr += adv ? 1 : 0;
            cost_row += adv ? cost_step : (ptrdiff_t)0;
            pc -= adv ? 1 : 0;
            pcoff -= adv ? (unsigned)a.Dc : 0u;
        }
    };
    // CODES: the fetched words of a row go to LDS one row before they are used (row parity of the row they belong to)
    auto park = [&](int row, const slot_t& sl) {
        uint32_t* Cn = cbuf + (row & 1) * CPB;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int m = tid + i * NT;
            if (m < NSRC) {
#pragma unroll
                for (int gg = 0; gg < 4; ++gg)
                    if (m - gg >= 0 && m - gg < CS) Cn[gg * CS + m - gg] = sl.x[i];
            } else if (m < NLOADS) {
                Cn[4 * CS + m - NSRC] = sl.x[i];
            }
        }
    };
    auto for_slots = [&](auto&& f) __attribute__((always_inline)) {  // f(slot tag) for every ring slot
        f(std::integral_constant<int, 0>{});
        if constexpr (PF > 1) f(std::integral_constant<int, 1>{});
        if constexpr (PF > 2) f(std::integral_constant<int, 2>{});
        if constexpr (PF > 3) f(std::integral_constant<int, 3>{});
    };
    for_slots([&](auto tag) __attribute__((always_inline)) { prefetch(tag, ring[decltype(tag)::value]); });
    if constexpr (CODES) {  // the first row's words are parked before the first barrier, its slot refilled (row r_lo + PF)
        park(r_lo, ring[0]);
        prefetch(std::integral_constant<int, 0>{}, ring[0]);
    }

    uint32_t LBa[Q], LBb[Q];  // the path that stays in its lane group (predecessor column c+1)
    uint32_t edgeV[2] = {kPadPk, kPadPk}, edgeA[2] = {kPadPk, kPadPk}, edgeB[2] = {kPadPk, kPadPk};  // (path_update)
    uint32_t MB = 0u;
#pragma unroll
    for (int q = 0; q < Q; ++q) { LBa[q] = padA[q]; LBb[q] = padB[q]; }

    __syncthreads();

    int c = base - r_lo + j;                                       // column of the step's row
    unsigned ooff = (unsigned)c * (unsigned)a.Dp + out_lane;
    uint32_t abort_seen = 0u;
    auto step = [&](int r, auto slot_tag, slot_t& sl) __attribute__((always_inline)) {
        constexpr int SL = decltype(slot_tag)::value;
        const uint32_t* Ep = lds8 + ((r - 1) & 1) * EBUF;
        uint32_t* En = lds8 + (r & 1) * EBUF;
        // predecessors: vertical path (r-1, c) = local column j-1, diagonal (r-1, c-1) = local column j-2 (LDS, previous row
        // parity), diagonal (r-1, c+1) = this lane group (registers)
        uint32_t LVa[Q], LVb[Q], LAa[Q], LAb[Q];
        {
            const uint32_t* srcV = Ep + (j + 1) * ES + sub * KS;
            const uint32_t* srcA = Ep + EDIR + j * ES + sub * KS;
#pragma unroll
            for (int i = 0; i < KS / 4; ++i) {
                if (4 * i + 2 < NR) {
                    const u32x4 t = *(const u32x4*)(srcV + 4 * i);
                    const u32x4 u = *(const u32x4*)(srcA + 4 * i);
                    LVa[2 * i] = t.x; LVb[2 * i] = t.y; LVa[2 * i + 1] = t.z; LVb[2 * i + 1] = t.w;
                    LAa[2 * i] = u.x; LAb[2 * i] = u.y; LAa[2 * i + 1] = u.z; LAb[2 * i + 1] = u.w;
                } else if (4 * i < NR) {
                    const u32x2 t = *(const u32x2*)(srcV + 4 * i);
                    const u32x2 u = *(const u32x2*)(srcA + 4 * i);
                    LVa[2 * i] = t.x; LVb[2 * i] = t.y;
                    LAa[2 * i] = u.x; LAb[2 * i] = u.y;
                }
            }
        }
        uint32_t MV = Ep[(j + 1) * ES + 16 * KS];
        uint32_t MA = Ep[EDIR + j * ES + 16 * KS];
        // the hand-off wavefront's "gave up" word rides with the predecessors' reads (a read of its own behind the barrier was a
        // second LDS round trip in every row's chain: ~300 of the row's ~3300 cycles); it is looked at once per PF rows
        abort_seen |= (uint32_t)ctl[1];
        // costs of the pixel: (d, d+2) and (d+1, d+3) pairs, padded disparities carry kPad16
        uint32_t ccA[Q], ccB[Q];
        if constexpr (CODES) {
            const uint32_t* Cc = cbuf + (r & 1) * CPB;
            const uint32_t* src = Cc + g * CS + wave * 4 + sub * KPL;  // copy g: word i = right word of local column i + g
            uint32_t rw[KPL];
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const u32x4 t = *(const u32x4*)(src + 4 * q);
                rw[4 * q] = t.x; rw[4 * q + 1] = t.y; rw[4 * q + 2] = t.z; rw[4 * q + 3] = t.w;
            }
            const uint32_t lw = Cc[4 * CS + j];
            // Which cells are numbers (census.cpp:132-172): the pixel's window inside the left image, the window at column
            // c + d inside the right one.  Wave-uniform test first: away from the borders every cell of the four pixels is.
            const int rimg = fam ? H - 1 - r : r;
            const int c0 = base - r + wave * 4;  // (uniform: column of the wavefront's first pixel)
            const int o = a.o;
            const bool all_ok = rimg >= o && rimg < H - o && c0 >= o && c0 + 3 < W - o && c0 + a.d0 >= o && c0 + 3 + a.d0 + D - 1 < W - o;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const uint32_t a0 = __builtin_popcount(lw ^ rw[4 * q]) + (padA[q] & 0xffffu);
                const uint32_t a2 = __builtin_popcount(lw ^ rw[4 * q + 2]) + (padA[q] >> 16);
                const uint32_t b1 = __builtin_popcount(lw ^ rw[4 * q + 1]) + (padB[q] & 0xffffu);
                const uint32_t b3 = __builtin_popcount(lw ^ rw[4 * q + 3]) + (padB[q] >> 16);
                ccA[q] = a0 | (a2 << 16);
                ccB[q] = b1 | (b3 << 16);
            }
            if (!all_ok) {  // (uniform: the image's borders)
                asm volatile("; cells that are not numbers" ::);
                // cell k of the lane is a number iff 0 <= u0 + k < W - 2 o, u0 = c + d0 + d_first - o (and the pixel's own window fits)
                const bool pix_ok = rimg >= o && rimg < H - o && c >= o && c < W - o;
                const int u0 = c + a.d0 + d_first - o;
                int klo = -u0, khi = W - 2 * o - u0;
                klo = klo < 0 ? 0 : klo;
                khi = khi > KPL ? KPL : khi;
                const uint32_t vm = (khi > klo && pix_ok) ? (((1u << khi) - 1u) & ~((1u << klo) - 1u)) : 0u;
                const uint32_t invpk = a.invalid_cost | (a.invalid_cost << 16);
                auto keep = [&](uint32_t cc, int k) {
                    const uint32_t m0 = (uint32_t)__builtin_amdgcn_sbfe((int)vm, k, 1), m2 = (uint32_t)__builtin_amdgcn_sbfe((int)vm, k + 2, 1);
                    const uint32_t mask = __builtin_amdgcn_perm(m2, m0, 0x07060100u);  // low half from m0, high half from m2
                    return (cc & mask) | (invpk & ~mask);
                };
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    ccA[q] = keep(ccA[q], 4 * q) | padA[q];
                    ccB[q] = keep(ccB[q], 4 * q + 1) | padB[q];
                }
            }
        } else {
            uint32_t cx[NDW];
            hring_take<kRing0 + 6 * SL, NDW, kRingCnt>(cx);
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                if (CBITS == 8) {
                    ccA[q] = (cx[q] & 0x00ff00ffu) | padA[q];
                    ccB[q] = ((cx[q] >> 8) & 0x00ff00ffu) | padB[q];
                } else {  // pair jj of the lane: bits 5 * (jj % 3) of both halves of dword jj / 3 (census_cost_u8_kernel)
                    constexpr uint32_t m5 = 0x001f001fu;
                    ccA[q] = ((cx[(2 * q) / 3] >> (5 * ((2 * q) % 3))) & m5) | padA[q];
                    ccB[q] = ((cx[(2 * q + 1) / 3] >> (5 * ((2 * q + 1) % 3))) & m5) | padB[q];
                }
            }
        }
        // paths that start at this pixel (first row, image border): (Lp, M) = (0, 0).  Rare: a wave-uniform branch
        const bool r0 = (r == 0);
        const bool rsA = r0 || c == 0, rsB = r0 || c == W - 1;
        if (__builtin_amdgcn_ballot_w64(rsA || rsB) != 0ull) {
            asm volatile("; path starts" ::);
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                LVa[q] = r0 ? padA[q] : LVa[q]; LVb[q] = r0 ? padB[q] : LVb[q];
                LAa[q] = rsA ? padA[q] : LAa[q]; LAb[q] = rsA ? padB[q] : LAb[q];
                LBa[q] = rsB ? padA[q] : LBa[q]; LBb[q] = rsB ? padB[q] : LBb[q];
            }
            MV = r0 ? 0u : MV; MA = rsA ? 0u : MA; MB = rsB ? 0u : MB;
        }
        uint32_t nVa[Q], nVb[Q], nAa[Q], nAb[Q], nBa[Q], nBb[Q];
        uint32_t mV = path_update<Q>(LVa, LVb, MV, ccA, ccB, P1pk, P2pk, nVa, nVb, edgeV);
        uint32_t mA = path_update<Q>(LAa, LAb, MA, ccA, ccB, P1pk, P2pk, nAa, nAb, edgeA);
        uint32_t mB = path_update<Q>(LBa, LBb, MB, ccA, ccB, P1pk, P2pk, nBa, nBb, edgeB);
        group_min3(mV, mA, mB);
        MB = mB;
        // next row's predecessors
        {
            uint32_t* dstV = En + (j + 2) * ES + sub * KS;
            uint32_t* dstA = En + EDIR + (j + 2) * ES + sub * KS;
#pragma unroll
            for (int i = 0; i < KS / 4; ++i) {
                if (4 * i + 2 < NR) {
                    u32x4 t, u;
                    t.x = nVa[2 * i]; t.y = nVb[2 * i]; t.z = nVa[2 * i + 1]; t.w = nVb[2 * i + 1];
                    u.x = nAa[2 * i]; u.y = nAb[2 * i]; u.z = nAa[2 * i + 1]; u.w = nAb[2 * i + 1];
                    *(u32x4*)(dstV + 4 * i) = t;
                    *(u32x4*)(dstA + 4 * i) = u;
                } else if (4 * i < NR) {
                    u32x2 t, u;
                    t.x = nVa[2 * i]; t.y = nVb[2 * i];
                    u.x = nAa[2 * i]; u.y = nAb[2 * i];
                    *(u32x2*)(dstV + 4 * i) = t;
                    *(u32x2*)(dstA + 4 * i) = u;
                }
            }
            if (sub == 0) {
                En[(j + 2) * ES + 16 * KS] = mV;
                En[EDIR + (j + 2) * ES + 16 * KS] = mA;
            }
        }
        // the family's sum leaves as bytes d, d+1, d+2, d+3 per dword (pads spill upwards only, into bytes that are pads)
        uint32_t packed[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const uint32_t sa = add3(nVa[q], nAa[q], nBa[q]), sb = add3(nVb[q], nAb[q], nBb[q]);
            packed[q] = sa | (sb << 8);
            LBa[q] = nBa[q];
            LBb[q] = nBb[q];
        }
        {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)out_row, 0, out_row_bytes, kRsrcWord3);
            store_dwords<Q>(rs, ((unsigned)c < (unsigned)W && lane_active) ? ooff : kOob, packed);
        }
        out_row += out_step;
        --c;
        ooff -= (unsigned)a.Dp;
        if constexpr (CODES) park(r + 1, sl);  // (sl: the ring slot that holds row r + 1's words; refilled with row r + 1 + PF)
        prefetch(slot_tag, sl);
        __syncthreads();
    };

    // ring slot of step u: the step's own costs - or, CODES, the words of the row after it (parked in LDS during the step)
    constexpr int SH = CODES ? 1 : 0;
    // The unrolled body has no branch between its steps and every memory instruction in it is unconditional: the compiler's wait
    // counts are then the ring's (pmx_buf.h).  After a failed hand-off the rows up to the next look at `abort_seen` are computed from
    // whatever the column slots hold and stored: the launch has failed by then (the next call that synchronises reports it: pmx_check_async_error), and a wavefront
    // that has ended no longer counts at the barrier.
    int r = r_lo;
    PMX_LOOP_ENTRY_DRAIN();
    for (; r + PF <= r_hi + 1; r += PF) {
        for_slots([&](auto tag) __attribute__((always_inline)) {
            constexpr int U = decltype(tag)::value, SL = (U + SH) % PF;
            step(r + U, std::integral_constant<int, SL>{}, ring[SL]);
        });
        if (__builtin_amdgcn_readfirstlane(abort_seen) != 0u) return;
    }
    for_slots([&](auto tag) __attribute__((always_inline)) {
        constexpr int U = decltype(tag)::value, SL = (U + SH) % PF;
        if (U < PF - 1 && r + U <= r_hi) step(r + U, std::integral_constant<int, SL>{}, ring[SL]);
    });
}

template <int KPL, int CBITS, int NW, bool CODES = false>
int launch_fam8(pmx_ctx* ctx, const fam8_args& a, int nwg) {
    constexpr int PF = 3;
    constexpr int Q = KPL / 4, NR = 2 * Q, KS = (NR + 3) & ~3, ES = 16 * KS + 4, CW = NW * 4;
    const size_t lds_bytes = (size_t)(2 * 2 * (CW + 2) * ES + 4 + (CODES ? 2 * (4 * (CW + 16 * KPL) + CW) : 0)) * sizeof(uint32_t);
    auto kern = sgm_fam8_kernel<KPL, CBITS, NW, PF, CODES>;
    PMX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(kern, dim3(nwg), dim3((NW + 2) * 64), lds_bytes, ctx->stream, a);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

}  // namespace

// compute wavefronts per window for an image width: the smallest of 4 / 8 that keeps the chain of windows short (every window
// border costs one hand-off per row and one step of pipeline lag)
int pmx_fam8_waves(const pmx_ctx* ctx, int W) {
    if (const char* e = pmx_opt(ctx, "SGM8_FAM_NW")) {
        const int v = atoi(e);
        if (v == 4 || v == 8) return v;
    }
    return W >= 2048 ? 8 : 4;
}

bool pmx_fam8_supported(int kpl, int H) { return (kpl % 4) == 0 && kpl >= 4 && kpl <= 20 && H >= 2; }

// Both vertical families (fams: bit 0 downward, bit 1 upward) into out + f * dstride
int pmx_launch_sgm_fam8(pmx_ctx* ctx, pmx_cv* cv, int kpl, bool five, int Dc, uint8_t* out, size_t dstride, uint32_t P1, uint32_t P2,
                        int fams, bool from_codes, uint32_t invalid_cost) {
    const int nw = pmx_fam8_waves(ctx, cv->W), CW = nw * 4;
    const int Q = kpl / 4;
    const int NB = (cv->W + CW - 1) / CW;
    const int NG = 3 * 8 * Q + 2, NGP = (NG + 63) / 64 * 64;
    const int fam0 = (fams & 1) ? 0 : 1, nfam = (fams == 3) ? 2 : 1;
    const size_t halo_fam = (size_t)cv->H * NB * NGP;
    int rc = pmx_fam_prepare(ctx, halo_fam * nfam * 16);
    if (rc) return rc;
    fam8_args a;
    a.cost = cv->cost8; a.out = out; a.dstride = dstride;
    a.H = cv->H; a.W = cv->W; a.D = cv->D; a.Dp = cv->Dp; a.Dc = Dc;
    a.P1 = P1; a.P2 = P2;
    a.halo = (u32x4*)ctx->fam_halo; a.halo_fam = halo_fam; a.NB = NB;
    a.epoch = pmx_fam_tag(++ctx->fam_epoch);
    a.ctl = ctx->fam_ctl;
    a.fam0 = fam0; a.nfam = nfam;
    const int nwin = (cv->W + cv->H - 2) / CW + 1;
    // windows of a family per chunk of an XCD's sequence: an XCD's CUs shared by the launch's families (SGM8_FAM_XCD=<G>: A/B hook)
    int G = pmx_cus_per_xcd() / nfam;
    if (const char* eg = pmx_opt(ctx, "SGM8_FAM_XCD")) {
        const int g = atoi(eg);
        if (g >= 1 && g <= 64) G = g;
    }
    if (G < 1) G = 1;
    const int nwg = (nwin + 8 * G - 1) / (8 * G) * (8 * G) * nfam;
    const int nchunk = (nwin + G - 1) / G;
    const size_t ncnt = ((size_t)nchunk * nfam + 7) / 8 * 8;
    const size_t xtab_words = ncnt + (size_t)nwin * nfam;
    if (ctx->fam_xtab_words < xtab_words) {
        if (ctx->fam_xtab) PMX_HIP(hipFree(ctx->fam_xtab));
        ctx->fam_xtab = nullptr;
        ctx->fam_xtab_words = 0;
        PMX_HIP(hipMalloc((void**)&ctx->fam_xtab, (xtab_words + 1024) * sizeof(unsigned)));
        ctx->fam_xtab_words = xtab_words + 1024;
    }
    ctx->fam_xtab_flags = ncnt;
    a.xtab = ctx->fam_xtab; a.started = ctx->fam_xtab + ncnt; a.G = G; a.nwin = nwin; a.nchunk = nchunk;
    const char* eprio = pmx_opt(ctx, "SGM8_FAM_PRIO");
    // (round 4, one hand-off wavefront: 0: 14.6 ms per 4096^2 x 257 step, 3: 13.9.  Round 5, two hand-off wavefronts, alternated on two boxes:
    //  3: 13.60 - 13.72 / 14.29 - 14.31, 2: 14.26 - 14.36, 1: 13.39 - 13.44 / 13.99 - 14.09, 0: 14.47 - 14.54)
    a.prio = eprio ? atoi(eprio) : 1;
    a.codes = cv->codes; a.code_bytes = (unsigned)cv->codes_bytes;
    a.codeL_off = (unsigned)(cv->codeL - cv->codes); a.codeR_off = (unsigned)(cv->codeR - cv->codes);
    a.d0 = cv->d0; a.o = cv->win / 2; a.invalid_cost = invalid_cost;
    if (from_codes) PMX_CHECK(cv->codes && cv->codes_bytes < 0xfffff000ull, PMX_ERR_STATE, "pmx_sgm (family form): census codes missing or too large");
    PMX_HIP(hipMemsetAsync(ctx->fam_xtab, 0, xtab_words * sizeof(unsigned), ctx->stream));  // tickets and the windows' XCDs (the error word is sticky)
    {
        pmx_stage_scope t(ctx, PMX_STAGE_SGM_FAMILY);
#define PMX_FAM8(KPLV, CB)                                                                              \
    (nw == 8 ? launch_fam8<KPLV, CB, 8>(ctx, a, nwg) : launch_fam8<KPLV, CB, 4>(ctx, a, nwg))
#define PMX_FAM8C(KPLV) (nw == 8 ? launch_fam8<KPLV, 8, 8, true>(ctx, a, nwg) : launch_fam8<KPLV, 8, 4, true>(ctx, a, nwg))
#define PMX_FAM8_KPL(KPLV) rc = from_codes ? PMX_FAM8C(KPLV) : (five ? PMX_FAM8(KPLV, 5) : PMX_FAM8(KPLV, 8))
        switch (kpl) {
            case 4: PMX_FAM8_KPL(4); break;
            case 8: PMX_FAM8_KPL(8); break;
            case 12: PMX_FAM8_KPL(12); break;
            case 16: PMX_FAM8_KPL(16); break;
            default: PMX_FAM8_KPL(20); break;
        }
#undef PMX_FAM8_KPL
#undef PMX_FAM8C
#undef PMX_FAM8
        if (rc) return rc;
    }
    PMX_HIP(hipMemcpyAsync(ctx->fam_err_host, ctx->fam_ctl + 1, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream));
    return PMX_OK;
}
