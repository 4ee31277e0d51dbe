// k_sgmfam.hip - float32 SGM, "family" schedule: three path directions fused into one marching pass.  gfx950.
//
// The per-direction schedule of k_sgm.hip reads the cost volume C and rewrites the accumulator S once per direction
// (8 x 12 B/cell).  The three paths that advance one image row per step - (+1,0), (+1,+1), (+1,-1), or their mirror
// images - share every read of C and every read-modify-write of S when one kernel marches down the rows computing all
// three at each pixel:  R C + R S + W S = 12 B/cell for three directions instead of 36.  The definition (oracle.c
// orc_sgm, k_sgm.hip header) sums family by family: a pass makes its family's sum F = (L_vertical + L_(pred col-1)) +
// L_(pred col+1) and writes out = F, in1 + F or (in1 + in2) + F - so the downward family, which reads no earlier sum, can run
// BESIDE the horizontal pair (round 6), and the upward family adds the two: S = (S_H + S_D) + S_U.
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
// s = 0 .. (W+H-2)/CW and is active on the rows where its window meets the image.  The window index comes from the
// tickets of pmx_buf.h (pmx_take_window): a workgroup prefers windows whose neighbours run on its own XCD and never takes a
// window whose left neighbour has not been taken - the taken windows are a prefix, every wait is a wait for a resident workgroup,
// the leftmost unfinished window waits for nobody: no co-residency assumption, no deadlock, also not beside other contexts or
// processes on the same device (every spin is bounded all the same).
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

#include "pmx_buf.h"
#include "pmx_internal.h"

namespace {

typedef __attribute__((address_space(1))) unsigned int gu32;

// This translation unit is compiled with -fno-honor-nans (v_min_f32 without the canonicalising v_max that IEEE mode otherwise
// asks for, DPP operands folded into the minimum): no floating-point instruction may see a NaN whose result matters.  The
// NaN costs of the input are recognised by their bits and replaced before any arithmetic.
__device__ __forceinline__ float f_inf() { return __int_as_float(0x7f800000); }
__device__ __forceinline__ bool is_nan_bits(float v) { return (__float_as_uint(v) & 0x7fffffffu) > 0x7f800000u; }
__device__ __forceinline__ float fmin2(float a, float b) { return __builtin_fminf(a, b); }

// minimum of a value with the same lane of the pixel's OTHER row of 16 lanes (32 lanes per pixel): gfx950's v_permlane16_swap
// exchanges the odd rows of one register with the even rows of another - given the value twice it leaves (row 0, row 0, row 2,
// row 2) and (row 1, row 1, row 3, row 3), whose minimum is the pair's in both rows.  One vector instruction where ds_swizzle was a
// round trip through the LDS pipe inside the row-synchronous chain.
__device__ __forceinline__ float min_other_row(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __builtin_fminf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ unsigned umin_other_row(unsigned v) {
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return r[0] < r[1] ? r[0] : r[1];
}

template <int CTRL>
__device__ __forceinline__ float dpp(float src) {  // lanes without a source lane read 0
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(src), CTRL, 0xf, 0xf, true));
}

constexpr int kSc1 = 16;  // aux bit of the buffer instructions: write-through store / L1-bypassing load

struct fam_args {
    const float* C;  // raw cost volume [H][W][D] (NaN = invalid)
    const float* in1;  // sums of earlier families (nullptr: none; may be `out`: every cell is read and written by one lane)
    const float* in2;  // a second earlier family's sums (IN2 instantiations only)
    float* out;        // in1 + in2 + this family's sum (not written in WTA mode)
    int H, W, D;
    int flip;        // 0: rows top -> bottom, paths (+1,0) (+1,+1) (+1,-1);  1: bottom -> top, paths (-1,0) (-1,+1) (-1,-1)
    float P1, P2, invalid_cost;
    int is_max, overcounting;
    int epilogue;    // last pass: overcounting, sign, NaN restore
    int dmask;       // paths that enter the family's sum: bit 0 vertical, bit 1 predecessor column c-1, bit 2 predecessor column c+1
    u32x4* halo;     // hand-off blocks [H][NB][NGP] of 16 bytes {value, value, value, epoch ^ the three}
    int NB;          // window borders per row = ceil(W / CW)
    unsigned epoch;
    unsigned* ctl;   // [1] error word ([0] is the integer marching kernel's ticket)
    // Window tickets (pmx_buf.h pmx_take_window): chunks of G consecutive windows belong to XCD (chunk mod 8), a workgroup prefers the
    // chunks of the XCD it runs on - G - 1 of G window borders then have both sides on one XCD and their hand-off can stay in that
    // XCD's L2 (see publish) - and never takes a window whose left neighbour is not taken.
    // xtab: [0 .. nchunk) windows taken per chunk; started[w] = 1 + the XCD window w runs on, 0 until it has started (zeroed per launch).
    unsigned* xtab;
    unsigned* started;
    int G, nwin, nchunk;
    // WTA mode (last pass only, template flag): S is not written; the pass reduces over D and leaves, per pixel, the winner's
    // disparity and (S[k-1], S[k], S[k+1], k) for the refinement step
    float* disp;
    float* near;     // float4 [H][W]
    double d0;
    int subpix;
    float invalid_disparity;
};

constexpr unsigned kSpinLimit = 1u << 21;  // polls before a hand-off gives up (seconds; a healthy wait is microseconds)

// minimum over the GL lanes of a pixel for three independent values at once (the chains interleave, which fills the wait
// states a DPP operand needs behind the instruction that wrote it); every lane of the pixel receives the results
template <int GL>
__device__ __forceinline__ void group_min3(float& a, float& b, float& c) {
    a = fmin2(a, dpp<0x121>(a)); b = fmin2(b, dpp<0x121>(b)); c = fmin2(c, dpp<0x121>(c));  // row_ror:1
    a = fmin2(a, dpp<0x122>(a)); b = fmin2(b, dpp<0x122>(b)); c = fmin2(c, dpp<0x122>(c));  // row_ror:2
    a = fmin2(a, dpp<0x124>(a)); b = fmin2(b, dpp<0x124>(b)); c = fmin2(c, dpp<0x124>(c));  // row_ror:4
    a = fmin2(a, dpp<0x128>(a)); b = fmin2(b, dpp<0x128>(b)); c = fmin2(c, dpp<0x128>(c));  // row_ror:8: the 16-lane row is done
    if (GL == 32) {  // the pixel's other row: lane ^ 16
        a = min_other_row(a);
        b = min_other_row(b);
        c = min_other_row(c);
    }
}

// One path, one pixel: new path costs from the predecessor's (Lp, M); padded disparities carry cc = +inf and come out +inf.
// restart (predecessor outside the image): L = C'.  Returns the lane's minimum of the new costs.
template <int GL, int KPL>
__device__ __forceinline__ float path_costs(const float (&Lp)[KPL], float M, bool restart, int l, const float (&cc)[KPL], float P1,
                                            float P2, float (&Ln)[KPL]) {
    float below = dpp<0x138>(Lp[KPL - 1]);  // wave_shr:1  lane l <- lane l-1
    float above = dpp<0x130>(Lp[0]);        // wave_shl:1  lane l <- lane l+1
    if (l == 0) below = f_inf();            // the neighbouring lane belongs to another pixel (or does not exist)
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
        Ln[k] = cc[k] + (t - M);
        lmin = fmin2(lmin, Ln[k]);
    }
    // a path that starts here (first row, image border): L = C'.  Rare, and the selects would cost 9 instructions per path and
    // step in the row-synchronous core that bounds the kernel: a wave-uniform branch instead
    if (__builtin_amdgcn_ballot_w64(restart) != 0ull) {
        lmin = f_inf();
#pragma unroll
        for (int k = 0; k < KPL; ++k) {
            Ln[k] = restart ? cc[k] : Ln[k];
            lmin = fmin2(lmin, Ln[k]);
        }
    }
    return lmin;
}

template <int GL, int KPL, int NW, int PF, bool WTA, bool IN2>
__global__ __launch_bounds__((NW + 1) * 64) void sgm_family_kernel(fam_args a) {
    constexpr int NPW = 64 / GL;           // pixels per wave
    constexpr int CW = NW * NPW;           // columns per workgroup window
    constexpr int KS = (KPL + 3) & ~3;     // LDS floats per lane slice (16-byte aligned)
    constexpr int ES = GL * KS + 4;        // LDS floats per (path, column): slices + the minimum
    constexpr int EDIR = (CW + 2) * ES;    // one path: column slots -2 .. CW-1
    constexpr int EBUF = 2 * EDIR;         // one row parity: vertical path, diagonal path
    constexpr int K3 = (KPL + 2) / 3;      // 16-byte hand-off blocks per lane and vector: {value, value, value, tag} each
    constexpr int NVB = GL * K3;           // blocks per handed-off vector
    constexpr int NG = 3 * NVB + 1;        // blocks per (row, border): V[CW-1], A[CW-1], A[CW-2], then one block of their three minima
    constexpr int NQ = (NG + 63) / 64;
    constexpr int NGP = NQ * 64;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // [0] window index, [1], [2] abort flag by row parity.  An LDS pointer by TYPE (address space 3): through a generic volatile
    // pointer these reads were flat loads, and a flat load's wait (vmcnt(0) lgkmcnt(0)) also waits for every global load in
    // flight - after the barrier of EVERY row the wave stood until the rows it had just prefetched arrived, which made one
    // memory round trip the duration of a row.
    typedef __attribute__((address_space(3))) int lds_int;
    volatile lds_int* ctl = (volatile lds_int*)(lds_int*)(lds + 2 * EBUF);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform values live in scalar registers: buffer
    if (threadIdx.x == 0) {                                              // descriptors built from them need no waterfall loop
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7u;
        // a window of this XCD's chunks if one may be taken (its left neighbour is), after a while any window that may: placement
        // is for speed only, every window is taken exactly once whatever the dispatcher does, and never ahead of its left neighbour
        pmx_win_tickets tk;
        tk.cnt = a.xtab; tk.G = a.G; tk.nwin = a.nwin; tk.nchunk = a.nchunk; tk.nfam = 1;
        const int win = pmx_take_window(tk, xcc, 0u);
        if (win >= 0) __hip_atomic_store(a.started + win, xcc + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ctl[0] = win;
        ctl[1] = 0;
        ctl[2] = 0;
        ctl[3] = (int)xcc;
    }
    __syncthreads();
    const int s = __builtin_amdgcn_readfirstlane(ctl[0]);
    if (s < 0) return;  // (more workgroups than windows: the launch is rounded up so that every XCD gets its share)
    const unsigned my_xcc = (unsigned)__builtin_amdgcn_readfirstlane(ctl[3]);
    const int H = a.H, W = a.W, D = a.D;
    const int base = s * CW;
    const int r_lo = base - W + 1 > 0 ? base - W + 1 : 0;
    const int r_hi = base + CW - 1 < H - 1 ? base + CW - 1 : H - 1;
    if (r_lo > r_hi) return;
    gu32* errw = (gu32*)(a.ctl + 1);
    constexpr unsigned kBlockBytes = (unsigned)NGP * 16u;

    if (wave == NW) {
        // ---- hand-off wave: brings the left neighbour's columns CW-2, CW-1 of row t into column slots -2, -1 ----
        // (every row's barrier waits for this wavefront: it goes first on its SIMD - A/B on one box at 4096^2 x 257: 14.2-14.4 ms per
        //  family against 15.0-15.2; 10000^2 x 129: no difference)
        __builtin_amdgcn_s_setprio(3);
        // A block is three values and a tag: tag = epoch ^ the three words, so that a block is taken only whole (a stale block
        // carries another epoch, a block of which only a part had arrived does not add up), at 12 of 16 bytes payload where two
        // {value, epoch} halves had 8 (the hand-off was a third of this kernel's traffic at 4096^2 x 257: 482 blocks per row and
        // border, now 289)
        int ldsoff[NQ];
        int nreal[NQ];  // how many of the block's three values exist (0: padding block, 4: the block of the three minima)
        int pairoff[NQ], oneoff[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int idx = q * 64 + lane;
            int vec, off, n;
            if (idx < 3 * NVB) {
                vec = idx / NVB;
                const int rem = idx - vec * NVB;  // block rem = k3 * GL + l: the lanes of one store are contiguous
                const int k3 = rem / GL;
                off = (rem - k3 * GL) * KS + 3 * k3;
                n = KPL - 3 * k3 >= 3 ? 3 : KPL - 3 * k3;
            } else if (idx == 3 * NVB) {  // the minima: V and A of column slot -1, A of column slot -2
                vec = 3; off = GL * KS; n = 4;
            } else {
                vec = 0; off = 0; n = 0;
            }
            // vec 0: vertical path of column CW-1 -> slot -1;  1: diagonal of CW-1 -> slot -1;  2: diagonal of CW-2 -> slot -2
            ldsoff[q] = (vec == 0 || vec == 3 ? 0 : EDIR) + (vec == 2 ? 0 : ES) + off;
            nreal[q] = n;
            // a full block moves as an 8-byte pair and a single word (whichever of the three sits at an even index starts the pair)
            pairoff[q] = ldsoff[q] + (off & 1);
            oneoff[q] = ldsoff[q] + ((off & 1) ? 0 : 2);
        }
        // Rows tA .. tB of the neighbour are needed (row t feeds this window's row t+1: 1 <= t+1 <= base, r_lo <= t+1 <= r_hi).
        // A row is read when it is due and read again (bounded spin) until all of its blocks carry this launch's tag.  Until round
        // 5 the rows were also requested four barriers ahead into a register ring; with the publishing order of round 6 (below)
        // a window runs less than a row behind its neighbour, no look-ahead load ever found its row, and the counters showed
        // 2.2 bytes of hand-off fetched for every byte published: the ring is gone (4096^2 x 257 / 10000^2 x 129, alternated on one
        // box: 43.9 / 125.2 and 43.7 / 125.4 ms per pipeline step without it against 46.7 / 126.9 and 44.3 / 125.8 with it).
        const int tA = r_lo - 1 > 0 ? r_lo - 1 : 0;
        const int tB = (r_hi < base ? r_hi : base) - 1;
        // descriptor of the neighbour's block for row t
        auto in_rsrc = [&](int t) {
            const int cb = base - (t + 1);  // image column of the neighbour's last pixel on row t (0 <= cb < W when needed)
#if defined(PMX_EXP_HALO) && (PMX_EXP_HALO & 1)  // timing experiment (results wrong): every hand-off read hits one cached slot
            return __builtin_amdgcn_make_buffer_rsrc((void*)(a.halo + (size_t)(s & 255) * NGP), 0, kBlockBytes, kRsrcWord3);
#endif
            return __builtin_amdgcn_make_buffer_rsrc((void*)(a.halo + ((size_t)t * a.NB + cb / CW) * NGP), 0, kBlockBytes, kRsrcWord3);
        };
        auto consume = [&](int t) -> bool {
            if (t < tA || t > tB) return true;
            const __amdgpu_buffer_rsrc_t rs = in_rsrc(t);
            u32x4 slot[NQ];
            for (unsigned spins = 0;; ++spins) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) slot[q] = __builtin_amdgcn_raw_buffer_load_b128(rs, nreal[q] ? (unsigned)(q * 64 + lane) * 16u : kOob, 0, kSc1);
#ifdef PMX_FAM_STATS
                if (lane == 0) atomicAdd(a.started + a.nwin + (spins == 0 ? 2 : 3), 1u);  // rows consumed / extra reads of a row
#endif
                bool ok = true;
#pragma unroll
                for (int q = 0; q < NQ; ++q) ok &= nreal[q] == 0 || slot[q].w == (a.epoch ^ slot[q].x ^ slot[q].y ^ slot[q].z);
#if defined(PMX_EXP_HALO) && (PMX_EXP_HALO & 1)
                ok = true;
#endif
                if (__all(ok)) break;
                if ((spins & 31) == 31 && __hip_atomic_load(errw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
                if (spins > kSpinLimit) {
                    if (lane == 0) __hip_atomic_store(errw, 1u + (unsigned)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    return false;
                }
                // (Round 6 tried waiting on the row's LAST block alone - one 16-byte request per poll - before reading the row again:
                //  fewer bytes, but a second dependent round trip in almost every row, since a window runs right behind its
                //  neighbour: 4096^2 x 257 48.5 / 47.3 ms with it against 46.7 / 46.3 without, alternated on one box.)
                __builtin_amdgcn_s_sleep(1);
            }
            float* Eb = lds + (t & 1) * EBUF;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if (nreal[q] == 4) {  // (ldsoff: the V minimum of slot -1)
                    Eb[ldsoff[q]] = __uint_as_float(slot[q].x);
                    Eb[ldsoff[q] + EDIR] = __uint_as_float(slot[q].y);
                    Eb[ldsoff[q] + EDIR - ES] = __uint_as_float(slot[q].z);
                } else if (nreal[q] == 3) {
                    float2 v;
                    v.x = __uint_as_float(slot[q].x);
                    v.y = __uint_as_float(slot[q].y);
                    *(float2*)(Eb + pairoff[q]) = v;
                    Eb[oneoff[q]] = __uint_as_float(slot[q].z);
                } else if (nreal[q] > 0) {
                    Eb[ldsoff[q]] = __uint_as_float(slot[q].x);
                    if (nreal[q] > 1) Eb[ldsoff[q] + 1] = __uint_as_float(slot[q].y);
                }
            }
            return true;
        };
        // Who reads what this window publishes: window s + 1.  When it runs on THIS XCD (it says so in xtab once it has started;
        // by construction G - 1 of G neighbours do), the blocks are written with plain stores: they stay in the XCD's L2, where the
        // neighbour's L1-bypassing loads find them - no trip over the fabric for the reader, and the 2.2 bytes fetched per byte
        // published (look-ahead loads that came too early, polls) become L2 hits.  Until it is known to be here - not started
        // yet, another XCD's chunk, a window taken out of turn - write-through (sc1) stores, which every reader sees.  Only ever
        // a question of speed: a block is taken by its tag whichever way it was written.
#ifdef PMX_FAM_ALLSC1  // experiment: write-through stores whoever reads them
        bool peer_known = true, peer_local = false;
#else
        bool peer_known = (s + 1) % a.G == 0 || s + 1 >= a.nwin, peer_local = false;
#endif
        unsigned peer_probe = 0;  // xtab[8 + s + 1] as read one row ago (in flight behind this row's other loads)
        auto publish = [&](int t) {
            if (!peer_known) {
                if (peer_probe != 0u) {
                    peer_known = true;
                    peer_local = peer_probe == my_xcc + 1u;
                } else {
                    peer_probe = __hip_atomic_load(a.started + s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
#ifdef PMX_FAM_STATS
            if (lane == 0) atomicAdd(a.started + a.nwin + (peer_local ? 0 : 1), 1u);  // rows published with plain / write-through stores
#endif
            const int cb = base + CW - 1 - t;
#if defined(PMX_EXP_HALO) && (PMX_EXP_HALO & 2)  // timing experiment (results wrong): nothing is published
            const bool need = false;
#else
            const bool need = t >= r_lo && cb < W && t < H - 1;
#endif
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(a.halo + ((size_t)(need ? t : 0) * a.NB + (need ? cb / CW : 0)) * NGP), 0, need ? kBlockBytes : 0u, kRsrcWord3);
            const float* Eb = lds + (t & 1) * EBUF + CW * ES;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                u32x4 b;
                if (nreal[q] == 3) {
                    const float2 v = *(const float2*)(Eb + pairoff[q]);
                    b.x = __float_as_uint(v.x);
                    b.y = __float_as_uint(v.y);
                    b.z = __float_as_uint(Eb[oneoff[q]]);
                } else if (nreal[q] == 4) {
                    b.x = __float_as_uint(Eb[ldsoff[q]]);
                    b.y = __float_as_uint(Eb[ldsoff[q] + EDIR]);
                    b.z = __float_as_uint(Eb[ldsoff[q] + EDIR - ES]);
                } else {
                    b.x = __float_as_uint(Eb[ldsoff[q]]);
                    b.y = nreal[q] > 1 ? __float_as_uint(Eb[ldsoff[q] + 1]) : b.x;
                    b.z = b.x;
                }
                b.w = a.epoch ^ b.x ^ b.y ^ b.z;
                if (peer_local) __builtin_amdgcn_raw_buffer_store_b128(b, rs, nreal[q] ? (unsigned)(q * 64 + lane) * 16u : kOob, 0, 0);
                else __builtin_amdgcn_raw_buffer_store_b128(b, rs, nreal[q] ? (unsigned)(q * 64 + lane) * 16u : kOob, 0, kSc1);
            }
        };
        // barrier index t runs from r_lo-1 (the barrier before the first step) to r_hi
        for (int tt = r_lo - 1; tt <= r_hi; ++tt) {
            // publish first: row tt-1 has been complete since barrier tt-1, and whatever this window still has to wait for from
            // ITS left neighbour must not hold up the window on its right (round 6: with the stores behind the wait every window
            // ran one row + one hand-off latency behind its neighbour, now one latency: 4096^2 x 257 46.0 - 46.3 -> 44.3 - 44.6 ms
            // per pipeline step, 10000^2 x 129 128.9 - 129.7 -> 126.2 - 126.4, alternated on one box)
            publish(tt - 1);
            if (!consume(tt)) ctl[1 + (tt & 1)] = 1;
            __syncthreads();
            if (__builtin_amdgcn_readfirstlane(ctl[1 + (tt & 1)])) return;
        }
        publish(r_hi);
        return;
    }

    // ---- compute waves -------------------------------------------------------------------------------------------
    const int g = lane / GL;
    const int l = lane - g * GL;
    const int j = wave * NPW + g;
    const int d0 = l * KPL;
    const int nv = d0 >= D ? 0 : (D - d0 < KPL ? D - d0 : KPL);  // disparities this lane owns
    const bool is_tail = nv > 0 && nv < KPL;
    const int tailn = D % KPL;  // uniform: what the pixel's last lane owns (0: every lane is full)
    using P = pieces<KPL>;
    int cov = 0;                // uniform: what the wide pieces cover of the tail lane
#pragma unroll
    for (int i = 0; i < P::N; ++i)
        if (P::start(i) + P::width(i) <= tailn) cov = P::start(i) + P::width(i);
    const int rem = tailn - cov;
    const unsigned lane_off = (unsigned)(d0 < D ? d0 : 0) * 4u;  // lanes without a disparity read the pixel's d = 0
    const unsigned row_bytes = (unsigned)W * (unsigned)D * 4u;
    const unsigned pix_bytes = (unsigned)D * 4u;
    float padv[KPL];  // -inf on owned disparities, +inf on padding: cc = max(cc, padv) leaves the former alone
#pragma unroll
    for (int k = 0; k < KPL; ++k) padv[k] = k < nv ? -f_inf() : f_inf();

    auto row_rsrc = [&](const float* vol, int r) {
        const int rimg = a.flip ? H - 1 - r : r;
        return __builtin_amdgcn_make_buffer_rsrc((void*)(vol + (size_t)rimg * W * D), 0, row_bytes, kRsrcWord3);
    };
    // byte offset of this lane's pixel inside row r: outside the image -> kOob (loads return 0, stores are dropped)
    auto pix_off = [&](int r) -> unsigned {
        const int c = base - r + j;
        return (c >= 0 && c < W) ? (unsigned)c * pix_bytes : kOob;
    };

    // loads of C and S run PF rows ahead in a register ring
    int pr = r_lo;
    float cbuf[PF][KPL], sbuf[PF][KPL];
    float tbuf[IN2 ? PF : 1][KPL];
    auto prefetch = [&](float (&cslot)[KPL], float (&sslot)[KPL], float (&tslot)[KPL]) {
        const unsigned off = pix_off(pr) + lane_off;
        buf_load<KPL>(row_rsrc(a.C, pr), off, cslot);
        buf_load<KPL>(row_rsrc(a.in1, pr), a.in1 ? off : kOob, sslot);  // first family of a sum: out of range = zeros, no traffic
        if (IN2) buf_load<KPL>(row_rsrc(a.in2, pr), off, tslot);
        if (pr < r_hi) ++pr;
    };
#pragma unroll
    for (int i = 0; i < PF; ++i) prefetch(cbuf[i], sbuf[i], tbuf[IN2 ? i : 0]);

    float LB[KPL];  // the path that stays in its lane group (predecessor column c+1)
    float MB = 0.f;
#pragma unroll
    for (int k = 0; k < KPL; ++k) LB[k] = f_inf();

    __syncthreads();
    if (__builtin_amdgcn_readfirstlane(ctl[1 + ((r_lo - 1) & 1)])) return;

    auto step = [&](int r, float (&cslot)[KPL], float (&sslot)[KPL], float (&tslot)[KPL]) {
        const int c = base - r + j;
        const float* Ep = lds + ((r - 1) & 1) * EBUF;
        float* En = lds + (r & 1) * EBUF;
        // predecessors: vertical path (r-1, c) = local column j-1, diagonal (r-1, c-1) = local column j-2 (LDS, previous row
        // parity), diagonal (r-1, c+1) = this lane group (registers)
        float LV[KPL], LA[KPL];
        const float* srcV = Ep + (j + 1) * ES + l * KS;
        const float* srcA = Ep + EDIR + j * ES + l * KS;
#pragma unroll
        for (int q = 0; q < KS / 4; ++q) {
            const float4 t = *(const float4*)(srcV + 4 * q);
            const float4 u = *(const float4*)(srcA + 4 * q);
            if (4 * q + 0 < KPL) { LV[4 * q + 0] = t.x; LA[4 * q + 0] = u.x; }
            if (4 * q + 1 < KPL) { LV[4 * q + 1] = t.y; LA[4 * q + 1] = u.y; }
            if (4 * q + 2 < KPL) { LV[4 * q + 2] = t.z; LA[4 * q + 2] = u.z; }
            if (4 * q + 3 < KPL) { LV[4 * q + 3] = t.w; LA[4 * q + 3] = u.w; }
        }
        const float MV = Ep[(j + 1) * ES + GL * KS];
        const float MA = Ep[EDIR + j * ES + GL * KS];
        // costs: NaN -> invalid_cost, sign for "max" measures, +inf on padded disparities
        float cc[KPL];
        // (by bits: this file is compiled with -fno-honor-nans, under which the compiler folds a floating-point class test to
        //  "never".  Which values were NaN is asked again - from the same registers - only by the last pass's epilogue, instead
        //  of building a bit mask on every pass)
        auto was_nan = [&](int k) { return is_nan_bits(cslot[k]); };
        if (a.is_max) {  // (uniform)
#pragma unroll
            for (int k = 0; k < KPL; ++k) {
                const float sg = __uint_as_float(__float_as_uint(cslot[k]) ^ 0x80000000u);
                cc[k] = __builtin_fmaxf(was_nan(k) ? a.invalid_cost : sg, padv[k]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < KPL; ++k) cc[k] = __builtin_fmaxf(was_nan(k) ? a.invalid_cost : cslot[k], padv[k]);
        }
        const bool r0 = (r == 0);
        float nV[KPL], nA[KPL], nB[KPL];
        float mV = path_costs<GL, KPL>(LV, MV, r0, l, cc, a.P1, a.P2, nV);
        float mA = path_costs<GL, KPL>(LA, MA, r0 || c == 0, l, cc, a.P1, a.P2, nA);
        float mB = path_costs<GL, KPL>(LB, MB, r0 || c == W - 1, l, cc, a.P1, a.P2, nB);
        group_min3<GL>(mV, mA, mB);
        MB = mB;
        // next row's predecessors
        float* dstV = En + (j + 2) * ES + l * KS;
        float* dstA = En + EDIR + (j + 2) * ES + l * KS;
#pragma unroll
        for (int q = 0; q < KS / 4; ++q) {
            float4 t, u;
            t.x = 4 * q + 0 < KPL ? nV[4 * q + 0] : 0.f; u.x = 4 * q + 0 < KPL ? nA[4 * q + 0] : 0.f;
            t.y = 4 * q + 1 < KPL ? nV[4 * q + 1] : 0.f; u.y = 4 * q + 1 < KPL ? nA[4 * q + 1] : 0.f;
            t.z = 4 * q + 2 < KPL ? nV[4 * q + 2] : 0.f; u.z = 4 * q + 2 < KPL ? nA[4 * q + 2] : 0.f;
            t.w = 4 * q + 3 < KPL ? nV[4 * q + 3] : 0.f; u.w = 4 * q + 3 < KPL ? nA[4 * q + 3] : 0.f;
            *(float4*)(dstV + 4 * q) = t;
            *(float4*)(dstA + 4 * q) = u;
        }
        if (l == 0) {
            En[(j + 2) * ES + GL * KS] = mV;
            En[EDIR + (j + 2) * ES + GL * KS] = mA;
        }
#pragma unroll
        for (int k = 0; k < KPL; ++k) LB[k] = nB[k];
        // the family's sum on an accumulator of its own (from +0, as the definition's), then the earlier families in front of it
        float acc[KPL];
#pragma unroll
        for (int k = 0; k < KPL; ++k) acc[k] = 0.f;
        if (a.dmask & 1) {
#pragma unroll
            for (int k = 0; k < KPL; ++k) acc[k] = acc[k] + nV[k];
        }
        if (a.dmask & 2) {
#pragma unroll
            for (int k = 0; k < KPL; ++k) acc[k] = acc[k] + nA[k];
        }
        if (a.dmask & 4) {
#pragma unroll
            for (int k = 0; k < KPL; ++k) acc[k] = acc[k] + nB[k];
        }
#pragma unroll
        for (int k = 0; k < KPL; ++k) acc[k] = (IN2 ? sslot[k] + tslot[k] : sslot[k]) + acc[k];  // (no in1: zeros; the family's sum is never -0)
        if (a.epilogue) {
#pragma unroll
            for (int k = 0; k < KPL; ++k) {
                float sv = acc[k];
                if (a.overcounting) sv = sv - 7.0f * cc[k];
                if (a.is_max) sv = __uint_as_float(__float_as_uint(sv) ^ 0x80000000u);
                acc[k] = was_nan(k) ? __uint_as_float(0x7fc00000u) : sv;
            }
        }
        if (!WTA) {
            buf_store<KPL>(row_rsrc(a.out, r), pix_off(r) + (unsigned)d0 * 4u, nv, is_tail, cov, rem, acc);
        } else {
            // winner-takes-all over the pixel's D values, as wta_kernel (k_disparity.hip) would do it on the stored volume: NaN
            // counts as the worst value, the FIRST extremum wins.  Two reductions over the pixel's lanes: the minimum value, then the
            // lowest index that holds it (float equality makes -0 and +0 tie); a third of the instructions of one reduction over
            // 64-bit (orderable value, index) keys, and this epilogue runs inside the row-synchronous core that bounds the kernel
            float vv[KPL];
            float vmin = f_inf();
#pragma unroll
            for (int k = 0; k < KPL; ++k) {
                float v = a.is_max ? __uint_as_float(__float_as_uint(acc[k]) ^ 0x80000000u) : acc[k];  // back to the "min" domain
                v = (was_nan(k) || k >= nv) ? f_inf() : v;  // NaN costs and padded disparities never win
                vv[k] = v;
                vmin = fmin2(vmin, v);
            }
            vmin = fmin2(vmin, dpp<0x121>(vmin));
            vmin = fmin2(vmin, dpp<0x122>(vmin));
            vmin = fmin2(vmin, dpp<0x124>(vmin));
            vmin = fmin2(vmin, dpp<0x128>(vmin));
            if (GL == 32) vmin = min_other_row(vmin);
            unsigned klo = 0xffffffffu;
#pragma unroll
            for (int k = KPL - 1; k >= 0; --k) klo = (vv[k] == vmin) ? (unsigned)(d0 + k) : klo;
            auto umin_dpp = [&](unsigned o) { klo = o < klo ? o : klo; };
            umin_dpp((unsigned)__builtin_amdgcn_mov_dpp((int)klo, 0x121, 0xf, 0xf, true));
            umin_dpp((unsigned)__builtin_amdgcn_mov_dpp((int)klo, 0x122, 0xf, 0xf, true));
            umin_dpp((unsigned)__builtin_amdgcn_mov_dpp((int)klo, 0x124, 0xf, 0xf, true));
            umin_dpp((unsigned)__builtin_amdgcn_mov_dpp((int)klo, 0x128, 0xf, 0xf, true));
            if (GL == 32) klo = umin_other_row(klo);
            const unsigned anyb = vmin != f_inf() ? 1u : 0u;  // some cost of the pixel is a number (sums of finite costs are finite)
            const int kw = (int)klo;            // the winner, the same in every lane of the pixel
            const int li = kw / KPL, kk = kw - li * KPL;
            // its neighbours in the output domain (what the stored volume would hold); NaN outside [0, D)
            const float qnan = __uint_as_float(0x7fc00000u);
            float prevv = dpp<0x138>(acc[KPL - 1]);  // lane l <- lane l-1
            float nextv = dpp<0x130>(acc[0]);        // lane l <- lane l+1
            float c0 = kk == 0 ? prevv : acc[0], c1 = acc[0], c2 = kk == KPL - 1 ? nextv : acc[0];
#pragma unroll
            for (int k = 0; k < KPL; ++k) {
                c1 = kk == k ? acc[k] : c1;
                if (k > 0) c0 = kk == k ? acc[k - 1] : c0;
                if (k < KPL - 1) c2 = kk == k ? acc[k + 1] : c2;
            }
            if (kw == 0) c0 = qnan;
            if (kw >= D - 1) c2 = qnan;
            const bool owner = (l == li);
            const unsigned poff = pix_off(r);  // kOob outside the image
            const unsigned pidx = poff == kOob ? kOob : poff / pix_bytes;
            const int rimg = a.flip ? H - 1 - r : r;
            const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void*)(a.disp + (size_t)rimg * W), 0, (unsigned)W * 4u, kRsrcWord3);
            const __amdgpu_buffer_rsrc_t rsN = __builtin_amdgcn_make_buffer_rsrc((void*)(a.near + (size_t)rimg * W * 4), 0, (unsigned)W * 16u, kRsrcWord3);
            const float dv = anyb ? (float)(a.d0 + (double)(unsigned)kw / (double)a.subpix) : a.invalid_disparity;
            u32x4 nb;
            nb.x = __float_as_uint(c0); nb.y = __float_as_uint(c1); nb.z = __float_as_uint(c2);
            nb.w = anyb ? (unsigned)kw : 0xffffffffu;  // -1: every cost of the pixel is NaN (the fix-up kernel applies the validity rule)
            const bool st = owner && pidx != kOob;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dv), rsD, st ? pidx * 4u : kOob, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(nb, rsN, st ? pidx * 16u : kOob, 0, 0);
        }
        // refill this ring slot with row r + PF (issued after the slot's last use: same registers, no copy)
        prefetch(cslot, sslot, tslot);
        __syncthreads();
    };

    int r = r_lo;
    bool dead = false;
    for (; r + PF <= r_hi + 1 && !dead; r += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            if (!dead) {
                step(r + u, cbuf[u], sbuf[u], tbuf[IN2 ? u : 0]);
                dead = __builtin_amdgcn_readfirstlane(ctl[1 + ((r + u) & 1)]) != 0;
            }
        }
    }
    if (dead) return;
#pragma unroll
    for (int u = 0; u < PF - 1; ++u) {
        if (r + u <= r_hi && !dead) {
            step(r + u, cbuf[u], sbuf[u], tbuf[IN2 ? u : 0]);
            dead = __builtin_amdgcn_readfirstlane(ctl[1 + ((r + u) & 1)]) != 0;
        }
    }
}

struct fam_shape {
    int gl, kpl, nw;
};

// Lane maps that are instantiated: 16 lanes per pixel up to 144 disparities, 32 up to 512.  The window width CW = nw * 64 / gl sets
// both the hand-off volume per cell (3 vectors per CW columns) and the number of windows in flight (W / CW): every CU wants a
// window, so narrow images take the 32-lane map (twice the waves per column) and 4 compute waves per workgroup.
bool pick_shape(const pmx_ctx* ctx, int D, int W, fam_shape* out) {
    static const int k16[] = {3, 5, 7, 9}, k32[] = {3, 5, 6, 9, 12, 16};
    fam_shape f{0, 0, 0};
    // (round 6, profiles/r06_fam_shape.txt: since the marching schedule is the default from 2400 columns, the 16-lane map is ahead there
    //  too when a lane holds 7 or 9 disparities - 2048 x 2600 x 129: 11.1 against 13.0 ms, 3000^2 x 129: 16.8 against 19.4, 1024 x 3500
    //  x 129: 7.8 against 8.7 - and not with 5: 2048 x 2600 x 65 9.7 against 9.0)
    if (D <= 144 && (W >= 16 * 224 || (W >= 2400 && D > 80))) {
        f.gl = 16;
        for (int k : k16)
            if (16 * k >= D) { f.kpl = k; break; }
    } else if (D <= 512) {
        f.gl = 32;
        for (int k : k32)
            if (32 * k >= D) { f.kpl = k; break; }
    }
    if (!f.kpl) return false;
    {
        // compute waves per workgroup: the smallest of 4 / 8 / 10 that gives every window of a row a CU of its own (a launch with
        // more windows than CUs runs them in two generations: 10000 columns in 32-column windows = 313 windows on 256 CUs took
        // 41.0 ms per family, in 40-column windows = 250 windows 37.2)
        static int cus = 0;
        if (!cus) {
            int dev = 0;
            hipDeviceProp_t prop;
            cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                      ? prop.multiProcessorCount : 256;
        }
        const int npw = 64 / f.gl;
        const int es = f.gl * ((f.kpl + 3) & ~3) + 4;  // floats per column slot of the exchange buffers (launch_family)
        auto fits = [&](int nw) { return (size_t)(2 * 2 * (nw * npw + 2) * es + 4) * sizeof(float) <= (size_t)160 * 1024; };
        f.nw = fits(10) ? 10 : 8;
        for (int nw : {4, 8, 10})
            if (fits(nw) && (W + nw * npw - 1) / (nw * npw) <= cus) { f.nw = nw; break; }
    }
    if (const char* e = pmx_opt(ctx, "SGM_FAM_SHAPE")) {  // test hook: "gl,kpl,nw" forces an instantiated lane map that fits D
        int gl = 0, kpl = 0, nw = 0;
        if (sscanf(e, "%d,%d,%d", &gl, &kpl, &nw) == 3 && gl * kpl >= D && (nw == 4 || nw == 8 || nw == 10)) {
            bool known = false;
            if (gl == 16) for (int k : k16) known |= k == kpl;
            if (gl == 32) for (int k : k32) known |= k == kpl;
            const size_t lds = (size_t)(2 * 2 * (nw * (64 / (gl ? gl : 64)) + 2) * (gl * ((kpl + 3) & ~3) + 4) + 4) * sizeof(float);
            if (known && lds <= (size_t)160 * 1024) f = fam_shape{gl, kpl, nw};  // (a map whose exchange buffers do not fit is ignored)
        }
    }
    if (out) *out = f;
    return true;
}

template <int GL, int KPL, int NW, bool WTA, bool IN2>
int launch_family(pmx_ctx* ctx, const fam_args& a, int nwg, hipStream_t st) {
    // rows of read-ahead: three; two where three rings of KPL registers would not fit (a second input, 16 disparities per lane)
    constexpr int PF = (KPL > 12 || IN2) ? 2 : 3;
    constexpr int NPW = 64 / GL, CW = NW * NPW, KS = (KPL + 3) & ~3, ES = GL * KS + 4;
    const size_t lds_bytes = (size_t)(2 * 2 * (CW + 2) * ES + 4) * sizeof(float);
    auto kern = sgm_family_kernel<GL, KPL, NW, PF, WTA, IN2>;
    PMX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(kern, dim3(nwg), dim3((NW + 1) * 64), lds_bytes, st, a);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

template <int GL, int KPL, int NW>
int launch_family_mode(pmx_ctx* ctx, const fam_args& a, int nwg, bool wta, hipStream_t st) {
    const bool in2 = a.in2 != nullptr;
    if (wta) return in2 ? launch_family<GL, KPL, NW, true, true>(ctx, a, nwg, st) : launch_family<GL, KPL, NW, true, false>(ctx, a, nwg, st);
    return in2 ? launch_family<GL, KPL, NW, false, true>(ctx, a, nwg, st) : launch_family<GL, KPL, NW, false, false>(ctx, a, nwg, st);
}

int dispatch_family(pmx_ctx* ctx, const fam_shape& f, const fam_args& a, int nwg, bool wta, hipStream_t st) {
#define PMX_FAM(GL, KPL)                                                                          \
    if (f.gl == GL && f.kpl == KPL) {                                                             \
        if (f.nw == 10) return launch_family_mode<GL, KPL, 10>(ctx, a, nwg, wta, st);             \
        return f.nw == 8 ? launch_family_mode<GL, KPL, 8>(ctx, a, nwg, wta, st) : launch_family_mode<GL, KPL, 4>(ctx, a, nwg, wta, st); \
    }
    PMX_FAM(16, 3)
    PMX_FAM(16, 5)
    PMX_FAM(16, 7)
    PMX_FAM(16, 9)
    PMX_FAM(32, 3)
    PMX_FAM(32, 5)
    PMX_FAM(32, 6)
    PMX_FAM(32, 9)
    PMX_FAM(32, 12)
    PMX_FAM(32, 16)
#undef PMX_FAM
    pmx_set_error("pmx_sgm (family schedule): no kernel for lane map %dx%d", f.gl, f.kpl);
    return PMX_ERR_STATE;
}

}  // namespace

// CUs of one XCD (the windows of one chunk of an XCD's ticket sequence, this file and k_sgmfam8.hip)
int pmx_cus_per_xcd() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t prop;
        const int cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                            ? prop.multiProcessorCount : 256;
        n = cus >= 8 ? cus / 8 : 1;
        if (n > 64) n = 64;
    }
    return n;
}

// The strip-to-strip hand-off buffer of the marching kernels (this file and k_sgmfam8.hip): at least `halo_bytes`, zeroed when
// (re)allocated so that a stale tag can never equal a future epoch (epochs count launches from 1), plus the ticket / error words.
int pmx_fam_prepare(pmx_ctx* ctx, size_t halo_bytes) {
    if (ctx->fam_halo_bytes < halo_bytes) {
        if (ctx->fam_halo) PMX_HIP(hipFree(ctx->fam_halo));
        ctx->fam_halo = nullptr;
        ctx->fam_halo_bytes = 0;
        PMX_HIP(hipMalloc((void**)&ctx->fam_halo, halo_bytes));
        ctx->fam_halo_bytes = halo_bytes;
        PMX_HIP(hipMemsetAsync(ctx->fam_halo, 0, halo_bytes, ctx->stream));
        ctx->fam_epoch = 0;
    }
    if (!ctx->fam_ctl) {
        PMX_HIP(hipMalloc((void**)&ctx->fam_ctl, 2 * sizeof(unsigned)));
        PMX_HIP(hipMemsetAsync(ctx->fam_ctl, 0, 2 * sizeof(unsigned), ctx->stream));
        PMX_HIP(hipHostMalloc((void**)&ctx->fam_err_host, sizeof(unsigned), hipHostMallocDefault));
        *ctx->fam_err_host = 0;
    }
    if (ctx->fam_epoch >= 0x7ffffff0u) {  // tags would repeat (pmx_fam_tag): start over on a clean buffer
        PMX_HIP(hipMemsetAsync(ctx->fam_halo, 0, ctx->fam_halo_bytes, ctx->stream));
        ctx->fam_epoch = 0;
    }
    return PMX_OK;
}

bool pmx_sgm_family_supported(const pmx_ctx* ctx, const pmx_cv* cv) { return cv->H >= 2 && pick_shape(ctx, cv->D, cv->W, nullptr); }

// the hand-off buffer for the marching passes of `cv`, made (and zeroed) on the context's own stream: the scheduler calls this
// before it forks a pass onto the second stream, the launches below then find it ready
int pmx_sgm_family_prepare(pmx_ctx* ctx, const pmx_cv* cv) {
    fam_shape f;
    PMX_CHECK(pick_shape(ctx, cv->D, cv->W, &f), PMX_ERR_UNSUPPORTED, "pmx_sgm (family schedule): D = %d not supported", cv->D);
    const int CW = f.nw * (64 / f.gl), NB = (cv->W + CW - 1) / CW;
    const int NG = 3 * f.gl * ((f.kpl + 2) / 3) + 1, NGP = (NG + 63) / 64 * 64;
    return pmx_fam_prepare(ctx, (size_t)cv->H * NB * NGP * 16);
}

int pmx_launch_sgm_family(pmx_ctx* ctx, pmx_cv* cv, int fam, const float* in1, const float* in2, float* out, float P1, float P2,
                          int is_max, float invalid_cost, int overcounting, int bits, bool epilogue, const pmx_fam_wta* wta,
                          hipStream_t st) {
    fam_shape f;
    PMX_CHECK(pick_shape(ctx, cv->D, cv->W, &f), PMX_ERR_UNSUPPORTED, "pmx_sgm (family schedule): D = %d not supported", cv->D);
    PMX_CHECK(bits && (in1 || !in2) && (out || wta), PMX_ERR_STATE, "pmx_sgm (family schedule): bad pass");
    if (!st) st = ctx->stream;
    const int npw = 64 / f.gl, CW = f.nw * npw;
    const int NB = (cv->W + CW - 1) / CW;
    const int NG = 3 * f.gl * ((f.kpl + 2) / 3) + 1, NGP = (NG + 63) / 64 * 64;  // 16-byte blocks per (row, border)
    const size_t halo_bytes = (size_t)cv->H * NB * NGP * 16;
    // (the hand-off buffer and its control words are (re)made on the context's own stream: a pass on the second stream was forked
    //  from it behind them)
    if (int rcp = pmx_fam_prepare(ctx, halo_bytes)) return rcp;
    const int nwin = (cv->W + cv->H - 2) / CW + 1;
    // windows per chunk of one XCD's sequence (fam_args::xtab): as many as an XCD has CUs, one workgroup each (SGM_FAM_XCD=<G>: A/B
    // hook; 1 = every neighbour on another XCD).  Measured at 4096^2 x 257 / 10000^2 x 129, ms per pipeline step on one box:
    // G = 1: 53.0 / 140, 8: 50.4 / 138, 16: 55.5 / 150, 32: 47.6 / 126 (the global ticket of round 5: 48.4 / 129).
    int G = pmx_cus_per_xcd();
    if (const char* eg = pmx_opt(ctx, "SGM_FAM_XCD")) {
        const int g = atoi(eg);
        if (g >= 1 && g <= 64) G = g;
    }
    const int nwg = (nwin + 8 * G - 1) / (8 * G) * (8 * G);  // every XCD's share of the launch covers its chunks
    const int nchunk = (nwin + G - 1) / G;
    const size_t ncnt = ((size_t)nchunk + 7) / 8 * 8;
    const size_t xtab_words = ncnt + (size_t)nwin + 8;  // (+ 8 statistics words of the PMX_FAM_STATS build)
    if (ctx->fam_xtab_words < xtab_words) {
        if (ctx->fam_xtab) PMX_HIP(hipFree(ctx->fam_xtab));
        ctx->fam_xtab = nullptr;
        ctx->fam_xtab_words = 0;
        PMX_HIP(hipMalloc((void**)&ctx->fam_xtab, (xtab_words + 1024) * sizeof(unsigned)));
        ctx->fam_xtab_words = xtab_words + 1024;
    }
    ctx->fam_xtab_flags = ncnt;
    fam_args a;
    a.C = cv->data;
    a.in1 = in1; a.in2 = in2; a.out = out;
    a.H = cv->H; a.W = cv->W; a.D = cv->D;
    a.flip = fam;
    a.P1 = P1; a.P2 = P2; a.invalid_cost = invalid_cost;
    a.is_max = is_max; a.overcounting = overcounting;
    a.epilogue = epilogue;
    a.dmask = bits;
    a.halo = (u32x4*)ctx->fam_halo;
    a.NB = NB;
    a.epoch = pmx_fam_tag(++ctx->fam_epoch);
    a.ctl = ctx->fam_ctl;
    a.xtab = ctx->fam_xtab;
    a.started = ctx->fam_xtab + ncnt;
    a.G = G;
    a.nwin = nwin;
    a.nchunk = nchunk;
    a.disp = wta ? wta->disp : nullptr;
    a.near = wta ? wta->near : nullptr;
    a.d0 = wta ? wta->d0 : 0.0;
    a.subpix = wta ? wta->subpix : 1;
    a.invalid_disparity = wta ? wta->invalid_disparity : 0.f;
    PMX_HIP(hipMemsetAsync(ctx->fam_xtab, 0, xtab_words * sizeof(unsigned), st));  // tickets and the windows' XCDs (the error word is sticky)
    {
        pmx_stage_scope t(ctx, PMX_STAGE_SGM_FAMILY, st);
        int rc = dispatch_family(ctx, f, a, nwg, wta != nullptr, st);
        if (rc) return rc;
    }
    // the error word travels to pinned host memory behind the launch; pmx_check_async_error reads it after a sync
    PMX_HIP(hipMemcpyAsync(ctx->fam_err_host, ctx->fam_ctl + 1, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    return PMX_OK;
}

// debug: the last marching launch's window table: [0..7] the first eight chunk counters (windows taken per chunk), [8 + w] = 1 + XCD of
// window w (of family w / nwin for the integer kernel)
extern "C" int pmx_debug_fam_windows(pmx_ctx* ctx, unsigned* host_out, int max_words) {
    PMX_CHECK(ctx && host_out && max_words > 8, PMX_ERR_ARG, "pmx_debug_fam_windows: null argument");
    PMX_CHECK(ctx->fam_xtab, PMX_ERR_STATE, "pmx_debug_fam_windows: no marching pass has run");
    PMX_HIP(hipStreamSynchronize(ctx->stream));
    const size_t flags = ctx->fam_xtab_flags, have = ctx->fam_xtab_words - flags;
    const size_t n = (size_t)(max_words - 8) < have ? (size_t)(max_words - 8) : have;
    PMX_HIP(hipMemcpy(host_out, ctx->fam_xtab, 8 * sizeof(unsigned), hipMemcpyDeviceToHost));
    PMX_HIP(hipMemcpy(host_out + 8, ctx->fam_xtab + flags, n * sizeof(unsigned), hipMemcpyDeviceToHost));
    return (int)n + 8;
}
