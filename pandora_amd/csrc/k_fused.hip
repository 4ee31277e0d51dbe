// k_fused.hip - the integer fast path: Census -> 8-path SGM -> WTA / refinement without ever
// materialising a float32 cost volume.  gfx950.
//
// Legal when every quantity is a small integer (census Hamming costs, integer P1 < P2, integer
// invalid cost, invalid_cost + P2 <= 255, subpix 1, no masks / disparity grids): then every
// L_r(p,d) is an exact integer <= invalid_cost + P2, so
//   * the matching cost C(p,d) = popcount(codeL(p) ^ codeR(p+d)) is RECOMPUTED inside the path
//     kernel from the census codes (4*NW bytes per pixel, L2 / Infinity-Cache resident) instead of
//     being read from HBM,
//   * each direction stores its L_r as ONE BYTE per cell into its own volume (no read-modify-write
//     of a float accumulator), and
//   * the 8-direction sum is formed where it is consumed: in the WTA kernel (sum8_wta_kernel), in
//     the refinement kernel, or in the float32 materialisation kernel when the caller really asks
//     for cv["cost_volume"].data.
// float32 arithmetic on these integers is exact, so the results are bit-identical to the general
// path (k_sgm.hip) and to the oracle.  HBM traffic: 8 B/cell written + 8 B/cell read for SGM+WTA
// against 24 B/cell algorithmic.
//
// All 8 directions run in ONE launch: a group of GL lanes (GL = 4, 8 or 16) walks one scanline of one
// direction, lanes over disparities (KPL per lane), DPP neighbour exchange and min-reduce inside the group, a
// register prefetch ring of census codes.  A wavefront therefore carries 64/GL scanlines.
//
// Measured limits on MI355X (tools/ubench, profiles/r01_c_pmc_sq.csv): a SIMD issues about one instruction
// per 4 cycles whatever the mix (VALU, SALU, 1/2/4 waves) and this kernel sits exactly on that bound (1.55e9
// wave-instructions per launch at C3 = 2.6 ms), and the texture addresser spends ~20 cycles per vector-memory
// instruction whatever its width.  So the step is written for FEW instructions and FEW, WIDE memory
// operations, and the lane map is chosen per D so that almost every lane-slot is a real disparity:
// D = 129 -> 8 lanes x 17 (136 slots, 8 scanlines per wave) instead of 16 x 12 (192 slots, 4 scanlines).
//
// Byte layout of a pixel in the per-direction volumes ([8][H][W][Dp], internal): lane s of the group owns
// disparities [s*KPL, (s+1)*KPL); its first M4 = KPL & ~3 bytes sit at s*M4 (one aligned wide store), the
// (KPL & 3 <= 1) remaining byte at nact*M4 + s (stored four lanes at a time as one dword), nact = ceil(D/KPL)
// = lanes that own a disparity.  Dp = nact*KPL rounded up to 4: no holes between pixels (a 192-byte pixel
// stride for 132 bytes of payload cost +35 %).  fused_pos() maps a disparity to its byte.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "pmx_internal.h"
#include "pmx_buf.h"

static constexpr int kWavesPerBlock = 4;
static constexpr uint32_t kInf = 0x7fffu;

// census-code read-ahead (pixels): as deep as the register budget of the lane map allows
__host__ __device__ constexpr int fused_ring(int nw, int kpl) { return nw * kpl <= 17 ? 4 : (nw * kpl <= 26 ? 3 : 2); }

__device__ __forceinline__ float g_inf() { return __int_as_float(0x7f800000); }
__device__ __forceinline__ float g_nan() { return __int_as_float(0x7fc00000); }

template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ uint32_t dppu(uint32_t oldv, uint32_t src) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)oldv, (int)src, CTRL, ROW_MASK, BANK_MASK, false);
}

__device__ __forceinline__ uint32_t umin2(uint32_t a, uint32_t b) { return a < b ? a : b; }

// min over the 64 lanes, wave-uniform result (used by the WTA kernel)
__device__ __forceinline__ uint32_t wave_min_u(uint32_t v) {
    v = umin2(v, dppu<0x111>(v, v));
    v = umin2(v, dppu<0x112>(v, v));
    v = umin2(v, dppu<0x114>(v, v));
    v = umin2(v, dppu<0x118>(v, v));
    v = umin2(v, dppu<0x142, 0xa>(v, v));
    v = umin2(v, dppu<0x143, 0xc>(v, v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// min over each group of GL lanes, result in EVERY lane of the group (rotate / permute butterfly, no readlane;
// old = identity of min lets each DPP move fold into the v_min_u32)
template <int GL>
__device__ __forceinline__ uint32_t group_allmin_u(uint32_t v) {
    static_assert(GL == 4 || GL == 8 || GL == 16, "group size");
    if (GL == 16) {
        v = umin2(v, dppu<0x128>(0xffffffffu, v));  // row_ror:8
        v = umin2(v, dppu<0x124>(0xffffffffu, v));  // row_ror:4
        v = umin2(v, dppu<0x122>(0xffffffffu, v));  // row_ror:2
        v = umin2(v, dppu<0x121>(0xffffffffu, v));  // row_ror:1
    } else {
        if (GL == 8) v = umin2(v, dppu<0x141>(0xffffffffu, v));  // row_half_mirror: lane i <-> 7-i
        v = umin2(v, dppu<0x4e>(0xffffffffu, v));                // quad_perm [2,3,0,1]
        v = umin2(v, dppu<0xb1>(0xffffffffu, v));                // quad_perm [1,0,3,2]
    }
    return v;
}
__device__ __forceinline__ uint32_t row_allmin_u(uint32_t v) { return group_allmin_u<16>(v); }

// byte offset of disparity index d inside a pixel of the per-direction volumes
__host__ __device__ __forceinline__ int fused_pos(int d, int nact, int kpl) {
    const int s = d / kpl, k = d - s * kpl, m4 = kpl & ~3;
    return k < m4 ? s * m4 + k : nact * m4 + s;
}

struct fused_args {
    const uint32_t* codeL;  // [H][W][NW]
    const uint32_t* codeR;  // [H][W][NW], readable 1024 dwords before / after
    uint8_t* ldir;          // [8][H][W][Dp], direction volumes dstride bytes apart
    size_t dstride;
    int H, W, D, Dp, d0, o;
    int nact;               // lanes of a group that own at least one disparity (ceil(D / KPL))
    uint32_t P1, P2, invalid_cost;
};

template <int NW, int KPL>
struct code_slot {
    uint32_t w[KPL * NW];  // right codes of the lane's KPL disparities
    uint32_t l[NW];        // left code of the pixel (same for the 16 lanes of a line)
};

// 64/GL scanlines of one direction per wavefront: lane group g walks line l0+g; lane `sub` of a group owns
// disparities [sub*KPL, (sub+1)*KPL).  All arithmetic is uint32 on small integers.  For GL < 16 the DPP
// row shifts cross group borders; that is harmless because GL*KPL > D is required, so the last slot of every
// group is a pad that reads as +infinity from below and whose own value is never stored or used.
template <int NW, int GL, int KPL>
__global__ __launch_bounds__(kWavesPerBlock * 64, 2) void sgm_census_fused_kernel(fused_args a) {
    constexpr int LPW = 64 / GL;        // scanlines per wavefront
    constexpr int M4 = KPL & ~3;        // bytes of the lane's wide store
    constexpr int T = KPL & 3;          // 0 or 1 trailing byte
    constexpr int kRing = fused_ring(NW, KPL);
    static_assert(T <= 1, "KPL must be 0 or 1 mod 4");
    const int lane = threadIdx.x & 63;
    const int sub = lane & (GL - 1), grp = lane / GL;
    const int gwave = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
    const int H = a.H, W = a.W, D = a.D;
    const int wavesH = (H + LPW - 1) / LPW, wavesW = (W + LPW - 1) / LPW;
    // directions in the order of k_sgm.hip / the oracle; 0,1 walk rows, 2..7 walk columns / diagonals
    int dir, l0;
    if (gwave < 2 * wavesH) {
        dir = gwave / wavesH;
        l0 = (gwave - dir * wavesH) * LPW;
    } else {
        const int t = gwave - 2 * wavesH;
        dir = 2 + t / wavesW;
        if (dir >= 8) return;
        l0 = (t - (dir - 2) * wavesW) * LPW;
    }
    const int dr = (dir < 2) ? 0 : ((dir & 1) ? -1 : 1);                        // 0 0 +1 -1 +1 -1 +1 -1
    const int dc = (dir == 0) ? 1 : (dir == 1) ? -1 : (dir < 4) ? 0 : ((dir == 4 || dir == 7) ? 1 : -1);  // +1 -1 0 0 +1 -1 -1 +1
    const bool horizontal = (dr == 0);
    const bool diagonal = (dr != 0) && (dc != 0);
    const int nlines = horizontal ? H : W;
    const int nsteps = horizontal ? W : H;
    const int line = min(l0 + grp, nlines - 1);  // surplus groups of the last wave repeat the last line (same bytes)
    const int d_first = sub * KPL;
    const bool lane_active = d_first < D;
    const int d_load = lane_active ? d_first : 0;

    // pixel being computed (per lane group) and the read-ahead cursor
    int r = horizontal ? line : (dr > 0 ? 0 : H - 1);
    int c = horizontal ? (dc > 0 ? 0 : W - 1) : line;
    int pc = c;
    int pleft = nsteps - 1;
    const int stride = dr * W + dc;  // pixel stride of one step (before wrapping)
    const uint32_t* pR = a.codeR + ((ptrdiff_t)r * W + c + a.d0 + d_load) * NW;
    const uint32_t* pL = a.codeL + ((ptrdiff_t)r * W + c) * NW;
    uint8_t* pO = a.ldir + (size_t)dir * a.dstride + ((size_t)r * W + c) * a.Dp + sub * M4;
    const int tail_delta = a.nact * M4 + sub - sub * M4;  // from the lane's wide store to its trailing byte

    code_slot<NW, KPL> ring[kRing];
    auto prefetch = [&](code_slot<NW, KPL>& slot) {
        __builtin_memcpy(&slot.w[0], pR, sizeof(uint32_t) * KPL * NW);
        __builtin_memcpy(&slot.l[0], pL, sizeof(uint32_t) * NW);
        if (pleft > 0) {  // wave-uniform; past the end the last pixel is re-read
            --pleft;
            pR += stride * NW;
            pL += stride * NW;
            if (diagonal) {
                pc += dc;
                const bool hi = pc >= W, lo = pc < 0;
                const int fix = hi ? -W : (lo ? W : 0);
                pc += fix;
                pR += fix * NW;
                pL += fix * NW;
            }
        }
    };
#pragma unroll
    for (int i = 0; i < kRing; ++i) prefetch(ring[i]);

    uint32_t Lp[KPL];
#pragma unroll
    for (int k = 0; k < KPL; ++k) Lp[k] = (d_first + k < D) ? 0u : kInf;
    uint32_t M = 0u;
    const uint32_t wvalid = (uint32_t)(W - 2 * a.o);  // number of valid right columns
    const int qbase = a.d0 + d_first - a.o;

    // per-lane pad mask: disparities >= D (tail of the last active lane, every inactive lane) must
    // look infinitely expensive to their neighbours and to the group minimum
    uint32_t padm[KPL];
#pragma unroll
    for (int k = 0; k < KPL; ++k) padm[k] = (d_first + k < D) ? 0u : kInf;

    // one pixel of each of the LPW lines.  ALL_OK = every cell touched is a valid census cell
    // (wave-uniform, true away from the image borders): no per-cell validity select.
    auto body = [&](code_slot<NW, KPL>& slot, auto all_ok_tag) {
        constexpr bool ALL_OK = decltype(all_ok_tag)::value;
        const uint32_t below = dppu<0x111>(kInf, Lp[KPL - 1]);  // row_shr:1 - disparity d_first-1 of the same line
        const uint32_t above = dppu<0x101>(kInf, Lp[0]);        // row_shl:1 - disparity d_first+KPL
        const uint32_t mp2 = M + a.P2;
        uint32_t negM = 0u - M;
        asm volatile("" : "+v"(negM));  // keep cc + t + negM a single three-operand add
        // per-lane validity, only on the slow path (image borders, disparities that leave the right image)
        bool pix_ok = true;
        uint32_t u = 0;
        if (!ALL_OK) {
            pix_ok = (r >= a.o) & (r < H - a.o) & (c >= a.o) & (c < W - a.o);
            u = (uint32_t)(c + qbase);  // element k is inside the right image iff u + k < wvalid (unsigned)
        }
        uint32_t Ln[KPL];
        uint32_t bytes[KPL];
#pragma unroll
        for (int k = 0; k < KPL; ++k) {
            // the pad mask rides on v_bcnt's add operand: a pad slot carries ~kInf in its cost, so it stays out of
            // every minimum without a separate OR; its (garbage) byte sits above the real ones of the lane
            uint32_t cc = padm[k];
#pragma unroll
            for (int w = 0; w < NW; ++w) cc += __popc(slot.l[w] ^ slot.w[k * NW + w]);
            if (!ALL_OK) cc = (pix_ok && (u + (uint32_t)k < wvalid)) ? cc : a.invalid_cost + padm[k];
            const uint32_t lo = (k > 0) ? Lp[k - 1] : below;
            const uint32_t hi = (k < KPL - 1) ? Lp[k + 1] : above;
            const uint32_t t = umin2(umin2(Lp[k], umin2(lo, hi) + a.P1), mp2);
            const uint32_t l = cc + t + negM;
            Ln[k] = l;
            bytes[k] = l;
        }
        uint32_t packed[M4 / 4];
#pragma unroll
        for (int q = 0; q < M4 / 4; ++q)  // three shift-or per dword
            packed[q] = (((bytes[4 * q + 3] << 8) | bytes[4 * q + 2]) << 16) | ((bytes[4 * q + 1] << 8) | bytes[4 * q]);
        if (lane_active) __builtin_memcpy(pO, packed, M4);
        if (T) {
            // byte stores cost about as much as the rest of the step (measured: 16x9 with a byte store 3.0 ms
            // against 16x8 2.1 ms at 2048^2 x 127), so four neighbouring lanes' trailing bytes are gathered with
            // two DPP shifts and leave as one dword from every fourth lane (pad lanes contribute don't-cares)
            const uint32_t tailv = bytes[KPL - 1] & 0xffu;
            uint32_t y = tailv | (dppu<0x101>(0u, tailv) << 8);  // row_shl:1 - lane i sees lane i+1
            y = y | (dppu<0x102>(0u, y) << 16);                  // row_shl:2
            if ((sub & 3) == 0 && lane_active) *reinterpret_cast<uint32_t*>(pO + tail_delta) = y;  // stays inside Dp
        }
        uint32_t lmin = Ln[0];
#pragma unroll
        for (int k = 1; k < KPL; ++k) lmin = umin2(lmin, Ln[k]);
#pragma unroll
        for (int k = 0; k < KPL; ++k) Lp[k] = Ln[k];
        return lmin;
    };

    // Wave-uniform test "every cell of this step is a valid census cell", on scalars only: s_r / s_c follow the pixel of
    // lane group 0; the groups of a row walk share the column and span LPW rows, those of a column / diagonal walk share
    // the row and span LPW columns (a group that wrapped away from the others fails the column test, which is all it
    // needs: the slow path is always correct).
    const int nact = a.nact;
    const int col_lo = max(a.o, a.o - a.d0);                                    // first column whose whole window is valid
    const int col_hi = min(W - a.o - 1, W - a.o - a.d0 - nact * KPL);           // last one
    const int rspan = horizontal ? LPW - 1 : 0, cspan = horizontal ? 0 : LPW - 1;
    int s_r = horizontal ? l0 : (dr > 0 ? 0 : H - 1);
    int s_c = horizontal ? (dc > 0 ? 0 : W - 1) : l0;

    auto step = [&](code_slot<NW, KPL>& slot) {
        const bool all_ok = (s_r >= a.o) & (s_r + rspan < H - a.o) & (s_c >= col_lo) & (s_c + cspan <= col_hi);
        uint32_t lmin;
        if (all_ok) lmin = body(slot, std::true_type{});
        else lmin = body(slot, std::false_type{});
        prefetch(slot);
        M = group_allmin_u<GL>(lmin);
        // advance; a diagonal line that leaves the image re-enters on the other side and the path restarts
        r += dr;
        c += dc;
        s_r += dr;
        s_c += dc;
        pO += (ptrdiff_t)stride * a.Dp;
        if (diagonal) {
            const bool hi = c >= W, lo = c < 0;
            const int fix = hi ? -W : (lo ? W : 0);
            c += fix;
            pO += (ptrdiff_t)fix * a.Dp;
            const bool wrapped = hi || lo;
#pragma unroll
            for (int k = 0; k < KPL; ++k) Lp[k] = wrapped ? padm[k] : Lp[k];
            M = wrapped ? 0u : M;
            s_c += (s_c >= W) ? -W : ((s_c < 0) ? W : 0);
        }
    };

    int i = 0;
    for (; i + kRing <= nsteps; i += kRing) {
#pragma unroll
        for (int j = 0; j < kRing; ++j) step(ring[j]);
    }
#pragma unroll
    for (int j = 0; j < kRing - 1; ++j)
        if (i + j < nsteps) step(ring[j]);
}

// ---- consumers of the 8 byte volumes ---------------------------------------------------------------
struct sum8_args {
    const uint8_t* ldir;  // [8][H][W][Dp], bytes of a pixel in fused_pos() order
    const uint32_t* range;  // [H][W] lo | hi << 16: the disparity indices that are numbers (cv_masked ran), or nullptr
    size_t dstride;         // bytes between the direction volumes
    int H, W, D, Dp, d0, o;
    int gl, kpl, nact;    // lane map the volumes were written with (nact = ceil(D / kpl) lanes own a disparity)
    int nvol;             // volumes that add up to S: eight paths, or three direction families (k_sgmfam8.hip)
};

// is cell (r, c, k) a NaN of the census volume?  Geometry alone, or the snapshot cv_masked took (grids, left mask)
__device__ __forceinline__ bool cell_is_nan(const sum8_args& a, int r, int c, int k) {
    if (a.range) {
        const uint32_t rg = a.range[(size_t)r * a.W + c];
        return !(k >= (int)(rg & 0xffffu) && k < (int)(rg >> 16));
    }
    const int q = c + a.d0 + k;
    return !((r >= a.o) && (r < a.H - a.o) && (c >= a.o) && (c < a.W - a.o) && (q >= a.o) && (q < a.W - a.o));
}

#define FMSK_INVALID 0x3C3LL
#define FMSK_STOPPED 0x8LL

// WTA over the summed volume (disparity.py:399-516 on S = sum of the eight byte volumes).
// Same lane map as the path kernel: 64/GL pixels per wavefront, lane `sub` of a group owns KPL consecutive
// disparities of its pixel = ONE wide load (+ one byte when KPL = 1 mod 4) per direction volume (the texture
// addresser charges per instruction).  The bytes are summed SWAR-style (even/odd bytes in 16-bit fields),
// (sum, index) is packed into one uint32 key per disparity so a single group min-reduce gives the FIRST
// minimum, and the winner's lane also writes (S[k-1], S[k], S[k+1], k) for the refinement step.
// As in the path kernel, GL*KPL > D makes the last slot of a group a pad, so the row shifts that fetch the
// neighbour lane's edge value never leak a value of the next pixel into a valid result.
template <int GL, int KPL, int NV>
__global__ __launch_bounds__(256) void sum8_wta_kernel(sum8_args a, size_t npix, double d0, float invalid_disparity,
                                                       float* __restrict__ disp, int64_t* __restrict__ validity,
                                                       float4* __restrict__ near) {
    constexpr int PPW = 64 / GL;   // pixels per wavefront
    constexpr int M4 = KPL & ~3, NB = M4 / 4, T = KPL & 3;
    // sums of one pixel in LDS (wave-private), slots d = -1 .. GL*KPL.  A lane's KPL sums start KST = KPL | 1 words after its
    // neighbour's and a pixel's row SROW = 16 (mod 32) words after the previous pixel's: with the natural strides (20 words per lane,
    // 324 per pixel) the 64 lanes of a store hit each bank up to eight times (73 % of this kernel's LDS cycles were conflicts);
    // now twice, the least 64 lanes on 32 banks can do
    constexpr int KST = KPL | 1;
    constexpr int SROW = ((GL * KST + 2 + 15) / 32) * 32 + 16;
    static_assert(T <= 1, "KPL must be 0 or 1 mod 4");
    __shared__ uint32_t sbuf[4][PPW][SROW];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int sub = lane & (GL - 1), grp = lane / GL;
    const size_t wave = (size_t)blockIdx.x * 4 + wv;
    const size_t nwaves = (size_t)gridDim.x * 4;
    const int d_first = sub * KPL;
    const bool lane_active = d_first < a.D;
    const size_t vol = a.dstride;
    const int wvalid = a.W - 2 * a.o;
    const int nown = min(KPL, a.D - d_first);  // real disparities of this lane (<= 0 for idle lanes)
    // candidate index of slot e with the pad mask folded in: pads get all-ones so that (sum << 16) | idx is never a minimum
    uint32_t idx[KPL];
#pragma unroll
    for (int e = 0; e < KPL; ++e) idx[e] = (e < nown) ? (uint32_t)(d_first + e) : 0xffffffffu;
    uint32_t* const srow = &sbuf[wv][grp][1];  // srow[d] = S(d), d = -1 .. GL*KPL
    for (size_t quad = wave; quad * PPW < npix; quad += nwaves) {
        const size_t pix = min(quad * PPW + grp, npix - 1);  // surplus groups repeat the last pixel (same values)
        const int r = (int)(pix / a.W), c = (int)(pix - (size_t)r * a.W);
        // ---- sum of the 8 directions: s[e] for the lane's KPL disparities (byte-select adds)
        uint32_t s[KPL];
#pragma unroll
        for (int e = 0; e < KPL; ++e) s[e] = 0;
        const uint8_t* base = a.ldir + pix * a.Dp + (lane_active ? sub * M4 : 0);  // idle lanes re-read lane 0
        const uint8_t* tbase = a.ldir + pix * a.Dp + a.nact * M4 + (lane_active ? sub : 0);
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            uint32_t x[NB];
            __builtin_memcpy(x, base + k * vol, 4 * NB);
            if (T) s[KPL - 1] += tbase[k * vol];
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                s[4 * q] += x[q] & 0xffu;
                s[4 * q + 1] += (x[q] >> 8) & 0xffu;
                s[4 * q + 2] += (x[q] >> 16) & 0xffu;
                s[4 * q + 3] += x[q] >> 24;
            }
        }
        // the sums go to LDS (wave-private row of the pixel) so that the winner's neighbours can be fetched by index
#pragma unroll
        for (int e = 0; e < KPL; ++e) srow[sub * KST + e] = s[e];
        // ---- validity of the lane's cells (geometry only on this path): slot e is a number iff elo <= e < ehi
        const bool pix_ok = (r >= a.o) & (r < a.H - a.o) & (c >= a.o) & (c < a.W - a.o);
        const int us = c + a.d0 + d_first - a.o;  // right column of slot 0, relative to the first valid one
        const bool interior = ((pix_ok & (us >= 0) & (us + nown <= wvalid)) | (nown <= 0)) & (a.range == nullptr);
        const uint32_t rg = a.range ? a.range[pix] : 0u;
        const int rlo = (int)(rg & 0xffffu), rhi = (int)(rg >> 16);
        uint32_t key = 0xffffffffu;
        if (__all(interior)) {
#pragma unroll
            for (int e = 0; e < KPL; ++e) key = umin2(key, (s[e] << 16) | idx[e]);
        } else {
            const int elo = a.range ? max(0, rlo - d_first) : max(0, -us);
            const int ehi = a.range ? min(nown, rhi - d_first) : (pix_ok ? min(nown, wvalid - us) : 0);
#pragma unroll
            for (int e = 0; e < KPL; ++e) {
                const bool ok = (e >= elo) & (e < ehi);
                key = umin2(key, ok ? ((s[e] << 16) | idx[e]) : 0xffffffffu);
            }
        }
        key = group_allmin_u<GL>(key);  // every lane of the group now holds the pixel's (min sum, first index)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- lane 0 of the group writes the result: winner, its neighbours S(k-1), S(k+1) (NaN where they are not numbers)
        if (sub == 0) {
            if (key == 0xffffffffu) {
                near[pix] = make_float4(g_nan(), g_nan(), g_nan(), __int_as_float(-8));
                disp[pix] = invalid_disparity;  // disparity.py:452-455
                int64_t m = validity[pix];
                if ((m & FMSK_INVALID) == 0) validity[pix] = FMSK_INVALID;  // disparity.py:471-474
            } else {
                const int kb = (int)(key & 0xffffu);
                const int q0 = c + a.d0 + kb - a.o;  // right column of the winner, relative
                const bool v0 = a.range ? (kb - 1 >= rlo) : ((kb - 1 >= 0) & (q0 - 1 >= 0));
                const bool v2 = a.range ? (kb + 1 < rhi) : ((kb + 1 < a.D) & (q0 + 1 < wvalid));
                const int k0 = kb - 1, k2 = kb + 1;  // slot of disparity d: (d / KPL) * KST + d % KPL (d = -1: the word before the row)
                const uint32_t c0 = k0 < 0 ? 0u : srow[(k0 / KPL) * KST + k0 % KPL], c2 = srow[(k2 / KPL) * KST + k2 % KPL];
                near[pix] = make_float4(v0 ? (float)c0 : g_nan(), (float)(key >> 16), v2 ? (float)c2 : g_nan(), __int_as_float(kb));
                disp[pix] = (float)(d0 + (double)kb);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // the next iteration overwrites the row
    }
}

// The WTA of the direction-family form (three byte volumes, 16 lanes x KPL disparities, KPL a multiple of 4, no per-pixel ranges):
// the kernel above spends about 300 instructions per four pixels and is bound by them (2.8 ms for 13.1 GB at 4096 x 4096 x 257 =
// 4.7 TB/s).  Same results here with half of them: a wavefront stays in one image row (no 64-bit divisions; the three volumes'
// rows behind buffer descriptors, a pixel four of them ahead already requested), the bytes are summed in pairs - v_perm spreads
// the even / odd bytes of a dword into two 16-bit fields, v_add3 adds the three volumes' - the candidate keys (sum << 16 | index)
// come straight from the pairs (v_lshl_or / v_and_or), the validity tests of the image's borders are one scalar test per four
// pixels, and the sums go to LDS as pairs (three wide writes instead of 20) for the winner's neighbours.
template <int KPL>
__global__ __launch_bounds__(256) void sum3_wta_kernel(sum8_args a, int qpw, double d0, float invalid_disparity, float* __restrict__ disp,
                                                       int64_t* __restrict__ validity, float4* __restrict__ near) {
    constexpr int Q = KPL / 4;
    constexpr int LST = 2 * Q;            // dwords of LDS per lane: the pairs (E0, O0, E1, O1, ..)
    constexpr int SROW = 16 * LST + 2;    // ... per pixel; S(d) is the 16-bit unit (d & ~3) | (d & 1) << 1 | (d >> 1) & 1 of the row
    static_assert(KPL % 4 == 0, "whole dwords per lane");
    __shared__ __attribute__((aligned(8))) uint32_t sbuf[4][4][SROW];
    const int lane = threadIdx.x & 63, wv = (int)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sub = lane & 15, grp = lane >> 4;
    const int H = a.H, W = a.W, D = a.D, o = a.o;
    const int r = blockIdx.y;
    const int quads_row = (W + 3) / 4;
    const int q0 = (blockIdx.x * 4 + wv) * qpw;  // (qpw = 16: two quads per turn of the loop, one result per lane of a group)
    if (q0 >= quads_row) return;
    const int q1 = min(quads_row, q0 + qpw);
    const int d_first = sub * KPL;
    const bool lane_active = d_first < D;
    const int nown = min(KPL, D - d_first);  // real disparities of this lane (<= 0 for idle lanes)
    const int wvalid = W - 2 * o;
    uint32_t idx[KPL];  // pads get all-ones: (sum << 16) | idx is then never a minimum
#pragma unroll
    for (int e = 0; e < KPL; ++e) idx[e] = (e < nown) ? (uint32_t)(d_first + e) : 0xffffffffu;
    const unsigned row_bytes = (unsigned)W * (unsigned)a.Dp;
    __amdgpu_buffer_rsrc_t rs[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
        rs[k] = __builtin_amdgcn_make_buffer_rsrc((void*)(a.ldir + k * a.dstride + (size_t)r * row_bytes), 0, row_bytes, kRsrcWord3);
    // (columns past the row's end - the last quad of a width that is no multiple of 4 - and idle lanes read zeros; nothing of them is stored)
    const unsigned voff = lane_active ? (unsigned)(grp * a.Dp + d_first) : kOob;
    // quads [qa, qb): every cell of their four pixels is a number (one scalar test per quad instead of the per-cell ones)
    const bool row_ok = r >= o && r < H - o;
    const int ca = max(o, o - a.d0), cb = min(W - o, W - o - a.d0 - (D - 1));  // columns [ca, cb)
    const int qa = row_ok ? (ca + 3) / 4 : 0, qb = row_ok ? cb / 4 : 0;
    uint32_t* const mine = &sbuf[wv][grp][sub * LST];
    const uint16_t* const halves = (const uint16_t*)&sbuf[wv][grp][0];
    auto half_of = [](int d) { return (d & ~3) | ((d & 1) << 1) | ((d >> 1) & 1); };

    auto fetch = [&](int q, uint32_t (&x)[3][Q]) {
#pragma unroll
        for (int k = 0; k < 3; ++k) load_dwords<Q>(rs[k], voff, x[k], (unsigned)(q * 4) * (unsigned)a.Dp);
    };
    uint32_t my_key = 0xffffffffu, my_nb = 0u;  // the result of quad q0 + sub (process)
    auto process = [&](int q, const uint32_t (&x)[3][Q]) {
        const int c = q * 4 + grp;
        // ---- sums of the three volumes: E[i] = (S(4i), S(4i+2)), O[i] = (S(4i+1), S(4i+3)) as 16-bit pairs
        uint32_t E[Q], O[Q];
#pragma unroll
        for (int i = 0; i < Q; ++i) {
            auto ev = [](uint32_t v) { return __builtin_amdgcn_perm(v, v, 0x0c020c00u); };
            auto od = [](uint32_t v) { return __builtin_amdgcn_perm(v, v, 0x0c030c01u); };
            E[i] = ev(x[0][i]) + ev(x[1][i]) + ev(x[2][i]);
            O[i] = od(x[0][i]) + od(x[1][i]) + od(x[2][i]);
            u32x2 t; t.x = E[i]; t.y = O[i];
            *(u32x2*)(mine + 2 * i) = t;
        }
        // ---- candidate keys; slot e of the lane is a number iff elo <= e < ehi
        uint32_t key = 0xffffffffu;
        if (q >= qa && q < qb) {  // (uniform)
#pragma unroll
            for (int i = 0; i < Q; ++i) {
                key = umin2(key, (E[i] << 16) | idx[4 * i]);
                key = umin2(key, (O[i] << 16) | idx[4 * i + 1]);
                key = umin2(key, (E[i] & 0xffff0000u) | idx[4 * i + 2]);
                key = umin2(key, (O[i] & 0xffff0000u) | idx[4 * i + 3]);
            }
        } else {
            const bool pix_ok = row_ok & (c >= o) & (c < W - o);
            const int us = c + a.d0 + d_first - o;  // right column of slot 0, relative to the first valid one
            const int elo = max(0, -us), ehi = pix_ok ? min(nown, wvalid - us) : 0;
#pragma unroll
            for (int i = 0; i < Q; ++i) {
                auto cand = [&](int e, uint32_t k) { return ((e >= elo) & (e < ehi)) ? k : 0xffffffffu; };
                key = umin2(key, cand(4 * i, (E[i] << 16) | idx[4 * i]));
                key = umin2(key, cand(4 * i + 1, (O[i] << 16) | idx[4 * i + 1]));
                key = umin2(key, cand(4 * i + 2, (E[i] & 0xffff0000u) | idx[4 * i + 2]));
                key = umin2(key, cand(4 * i + 3, (O[i] & 0xffff0000u) | idx[4 * i + 3]));
            }
        }
        key = group_allmin_u<16>(key);  // every lane of the group now holds the pixel's (min sum, first index)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- the winner's neighbours S(k-1), S(k+1) while the pixel's sums are in LDS; lane (q - q0) of the group keeps the
        // pixel's result: the maps are written once per 16 quads, by all 64 lanes (the result code, about 60 instructions, once
        // per wavefront instead of once per quad with four lanes active)
        {
            const int kb = (int)(key & 0xffffu);  // (a pixel without a number: 0xffff - the reads stay inside the row, unused)
            const uint32_t c0 = halves[half_of(min(max(kb - 1, 0), 16 * KPL - 1))], c2 = halves[half_of(min(kb + 1, 16 * KPL - 1))];
            const bool keep = sub == q - q0;
            my_key = keep ? key : my_key;
            my_nb = keep ? (c0 | (c2 << 16)) : my_nb;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // the next quad overwrites the row
    };
    // two quads per turn, each requested while the one before it is worked on (past the wavefront's last quad the last one is
    // requested again - lines that are in the cache, not a neighbour's bytes from memory - and worked on again: the same result
    // into the same lane)
    uint32_t xa[3][Q], xb[3][Q];
    fetch(q0, xa);
    for (int q = q0; q < q1; q += 2) {
        fetch(min(q + 1, q1 - 1), xb);
        process(q, xa);
        fetch(min(q + 2, q1 - 1), xa);
        process(min(q + 1, q1 - 1), xb);
    }
    // ---- the maps: lane `sub` of group `grp` holds pixel (q0 + sub) * 4 + grp
    const int c = (q0 + sub) * 4 + grp;
    if (q0 + sub < q1 && c < W) {
        const size_t pix = (size_t)r * W + c;
        const uint32_t key = my_key;
        if (key == 0xffffffffu) {
            near[pix] = make_float4(g_nan(), g_nan(), g_nan(), __int_as_float(-8));
            disp[pix] = invalid_disparity;  // disparity.py:452-455
            int64_t m = validity[pix];
            if ((m & FMSK_INVALID) == 0) validity[pix] = FMSK_INVALID;  // disparity.py:471-474
        } else {
            const int kb = (int)(key & 0xffffu);
            const int qr = c + a.d0 + kb - o;  // right column of the winner, relative
            const bool v0 = (kb - 1 >= 0) & (qr - 1 >= 0);
            const bool v2 = (kb + 1 < D) & (qr + 1 < wvalid);
            near[pix] = make_float4(v0 ? (float)(my_nb & 0xffffu) : g_nan(), (float)(key >> 16), v2 ? (float)(my_nb >> 16) : g_nan(),
                                    __int_as_float(kb));
            disp[pix] = (float)(d0 + (double)kb);
        }
    }
}

__device__ __forceinline__ float sum8_cell(const sum8_args& a, size_t pix, int r, int c, int k) {
    if (cell_is_nan(a, r, c, k)) return g_nan();
    const size_t vol = a.dstride;
    uint32_t s = 0;
    const int pos = fused_pos(k, a.nact, a.kpl);
    for (int j = 0; j < a.nvol; ++j) s += a.ldir[j * vol + pix * a.Dp + pos];
    return (float)s;
}

// refinement.cpp:28-99 + vfit / quadratic on the summed byte volumes ("min" measure only)
__global__ __launch_bounds__(256) void sum8_refine_kernel(sum8_args a, size_t npix, double d_min, double d_max, int method,
                                                          float* __restrict__ disp, int64_t* __restrict__ validity,
                                                          float* __restrict__ itp, const float4* __restrict__ near) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    int64_t m = validity[i];
    if ((m & FMSK_INVALID) != 0) { itp[i] = g_nan(); return; }
    const int r = (int)(i / a.W), c = (int)(i - (size_t)r * a.W);
    float raw = disp[i];
    if (!((double)raw >= d_min && (double)raw <= d_max)) { itp[i] = g_nan(); return; }  // as refine_kernel: never index outside the volume
    int k = (int)(((double)raw - d_min) * 1.0);
    // the WTA step left (S[k-1], S[k], S[k+1], k) of its winner; if the disparity map was edited on
    // the host since (a filter), fall back to gathering from the eight volumes
    const float4 nb = near ? near[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const bool cached = near && __float_as_int(nb.w) == k;
    float c1 = cached ? nb.y : sum8_cell(a, i, r, c, k);
    if (c1 != c1) { itp[i] = c1; return; }
    if ((double)raw == d_min || (double)raw == d_max) { itp[i] = c1; validity[i] = m + FMSK_STOPPED; return; }
    float c0 = cached ? nb.x : sum8_cell(a, i, r, c, k - 1);
    float c2 = cached ? nb.z : sum8_cell(a, i, r, c, k + 1);
    float sd, sc;
    int64_t flag = 0;
    if (c0 != c0 || c2 != c2 || c1 > c0 || c1 > c2) {
        sd = 0.f; sc = c1; flag = FMSK_STOPPED;
    } else if (method == PMX_REFINE_VFIT) {
        float aa = c0 > c2 ? c0 - c1 : c2 - c1;
        if (fabs((double)aa) < 1.0e-15) { sd = 0.f; sc = c1; }
        else { sd = (c0 - c2) / (2 * aa); sc = aa * (sd - 1) + c2; }
    } else {
        float alpha = (c0 - 2.f * c1 + c2) / 2.f;
        float beta = (c2 - c0) / 2.f;
        float x = -beta / (2.f * alpha);
        float mx = (-1.f < x) ? x : -1.f;
        sd = (mx < 1.f) ? mx : 1.f;
        sc = (alpha * sd * sd) + (beta * sd) + c1;
    }
    disp[i] = raw + sd / 1.0f;
    itp[i] = sc;
    validity[i] = m + flag;
}

// float32 materialisation (only when the caller reads the volume or a float-only step follows)
__global__ __launch_bounds__(256) void sum8_to_float_kernel(sum8_args a, float* __restrict__ cv) {
    const int r = blockIdx.y;
    int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= a.W * a.D) return;
    int c = j / a.D, k = j - c * a.D;
    cv[(size_t)r * a.W * a.D + j] = sum8_cell(a, (size_t)r * a.W + c, r, c, k);
}

// pixels whose census cost is NaN for every disparity, from geometry alone
__global__ __launch_bounds__(256) void census_nan_pixels_kernel(sum8_args a, uint8_t* __restrict__ out) {
    int c = blockIdx.x * 256 + threadIdx.x;
    int r = blockIdx.y;
    if (c >= a.W) return;
    bool any = false;
    for (int k = 0; k < a.D && !any; ++k) any = !cell_is_nan(a, r, c, k);
    out[(size_t)r * a.W + c] = any ? 0 : 1;
}

// ---- cv_masked on the integer path: snapshot of the cells that are numbers --------------------------------------------------
// matching_cost.py:815-860 for subpix 1 without a right mask: a cell is a number iff the census geometry allows it, the
// left pixel is not masked (invalid or dilated no-data) and disp_min[r,c] <= d <= disp_max[r,c] (NaN grids compare false, as
// in numpy).  All three are intervals of the disparity index, so their intersection [lo, hi) describes the pixel.
__global__ __launch_bounds__(256) void build_range_kernel(int H, int W, int D, int d0, int o, const uint8_t* __restrict__ bad_left,
                                                          const double* __restrict__ gmin, const double* __restrict__ gmax,
                                                          uint32_t* __restrict__ range) {
    const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
    if (c >= W) return;
    const size_t i = (size_t)r * W + c;
    int lo = max(0, o - c - d0), hi = min(D, W - o - c - d0);
    if (!((r >= o) && (r < H - o) && (c >= o) && (c < W - o))) hi = lo;
    if (bad_left && bad_left[i]) hi = lo;
    if (gmin) {
        const double a = gmin[i] - (double)d0, b = gmax[i] - (double)d0;  // k >= a and k <= b
        if (a == a && a > (double)lo) lo = a >= (double)D ? D : (int)ceil(a);
        if (b == b && b < (double)(hi - 1)) hi = b < 0.0 ? 0 : (int)floor(b) + 1;
    }
    if (hi < lo) hi = lo;
    range[i] = (uint32_t)lo | ((uint32_t)hi << 16);
}

__global__ __launch_bounds__(256) void range_nan_kernel(int W, int D, const uint32_t* __restrict__ range, float* __restrict__ cv) {
    const int r = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= W * D) return;
    const int c = j / D, k = j - c * D;
    const uint32_t rg = range[(size_t)r * W + c];
    if (!(k >= (int)(rg & 0xffffu) && k < (int)(rg >> 16))) cv[(size_t)r * W * D + j] = g_nan();
}

int pmx_launch_build_range(pmx_ctx* ctx, pmx_cv* cv) {
    const size_t need = (size_t)cv->H * cv->W * sizeof(uint32_t);
    if (cv->range_bytes < need) {
        pmx_pool_free(ctx, cv->range);
        cv->range = nullptr;
        cv->range_bytes = 0;
        PMX_HIP(pmx_pool_alloc(ctx, (void**)&cv->range, need));
        cv->range_bytes = need;
    }
    dim3 grid((cv->W + 255) / 256, cv->H);
    hipLaunchKernelGGL(build_range_kernel, grid, dim3(256), 0, ctx->stream, cv->H, cv->W, cv->D, cv->d0, cv->win / 2,
                       (const uint8_t*)ctx->bad_left, (const double*)ctx->grid_min, (const double*)ctx->grid_max, cv->range);
    PMX_HIP(hipGetLastError());
    cv->has_range = true;
    return PMX_OK;
}

int pmx_launch_range_nan(pmx_ctx* ctx, pmx_cv* cv) {
    dim3 grid((cv->W * cv->D + 255) / 256, cv->H);
    hipLaunchKernelGGL(range_nan_kernel, grid, dim3(256), 0, ctx->stream, cv->W, cv->D, cv->range, cv->data);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

// ---- host side ---------------------------------------------------------------------------------------
static sum8_args make_sum8(const pmx_cv* cv) {
    sum8_args s;
    s.ldir = cv->ldir;
    s.range = cv->has_range ? cv->range : nullptr;
    s.dstride = cv->dstride;
    s.H = cv->H; s.W = cv->W; s.D = cv->D; s.Dp = cv->Dp; s.d0 = cv->d0; s.o = cv->win / 2;
    s.gl = cv->gl; s.kpl = cv->kpl; s.nact = cv->kpl ? (cv->D + cv->kpl - 1) / cv->kpl : 0;  // (no map before the SGM step)
    s.nvol = cv->nvol;
    return s;
}

bool pmx_fused_sgm_eligible(const pmx_ctx* ctx, const pmx_cv* cv, float P1, float P2, int is_max, float invalid_cost,
                            int overcounting) {
    if (cv->repr != PMX_REPR_CENSUS_DEFERRED) return false;
    if (is_max || overcounting) return false;
    if (cv->subpix != 1 || cv->D >= 16 * 20) return false;
    const int nw = (cv->win * cv->win + 31) / 32;
    if (nw > 2) {  // wider census codes exist for the packed kernels only (k_sgm8.hip)
        const char* e8 = pmx_opt(ctx, "SGM8");
        if (nw > 6 || nw == 5 || pmx_opt(ctx, "FUSED_MAP") || (e8 && e8[0] == '0')) return false;
    }
    auto is_int = [](float x) { return x == floorf(x); };
    if (!is_int(P1) || !is_int(P2) || !is_int(invalid_cost)) return false;
    if (invalid_cost < 0 || invalid_cost + P2 > 255.f) return false;
    if (cv->has_range) {  // the snapshot of cv_masked is honoured by the packed kernels only (k_sgm8.hip)
        const char* e8 = pmx_opt(ctx, "SGM8");
        if (pmx_opt(ctx, "FUSED_MAP") || (e8 && e8[0] == '0')) return false;
    }
    return true;
}

// Lane map for a volume: gl lanes per scanline, kpl disparities per lane with gl*kpl > D (the pad slot the
// DPP shifts rely on) and kpl = 0 or 1 mod 4 (one wide store + at most one byte).  Among the maps the kernels
// are instantiated for, take the cheapest by a model of one wave-step fitted on MI355X (tools/bench_fused.py with
// forced maps; ns per wave-step = kernel time / steps / waves per SIMD):
//   16x12 311, 16x8 252, 16x9 324 (2048^2, 4 waves/SIMD); 8x17 624, 8x20 683 (4096^2, 4 waves/SIMD);
//   lone or paired waves (latency-bound): 16x5 521, 16x8 661, 8x12 857, 8x16 1172, 8x17 1357.
//   issue   = 1.75 ns x (70 + 9.5 kpl + #vmem)          one instruction per ~4.2 cycles per SIMD
//   memory  = #vmem x tau(gl), tau = 55 / 85 / 120 ns   texture path; narrower groups touch more rows per instruction
//   latency = 250 + 60 kpl (+100 with a trailing byte)  what a step costs a wave that has the SIMD to itself
//   step    = max(latency, waves_per_SIMD x max(issue, memory));  +8 % for trailing-byte maps (measured, unexplained)
static void fused_choose_map(const pmx_ctx* ctx, int H, int W, int D, int nw, int* gl_out, int* kpl_out) {
    // test hook: PMX_FUSED_MAP=<gl>x<kpl> forces a map (used by the parity tests to reach every instantiation
    // on small volumes); ignored unless it is a legal map for this D
    if (const char* e = pmx_opt(ctx, "FUSED_MAP")) {
        int gl = 0, kpl = 0;
        if (sscanf(e, "%dx%d", &gl, &kpl) == 2 && (gl == 4 || gl == 8 || gl == 16) && gl * kpl > D && (kpl & 3) <= 1 &&
            kpl >= 4 && kpl <= 20) {
            *gl_out = gl;
            *kpl_out = kpl;
            return;
        }
    }
    double best = 1e30;
    *gl_out = 16;
    *kpl_out = 20;
    for (int gl = 16; gl >= 4; gl /= 2) {  // widest group first: a narrower one must win by 5 % (model accuracy)
        for (int t = 0; t < 2; ++t) {
            int kpl = D / gl + 1;  // gl*kpl > D
            if (t == 0) kpl = (kpl + 3) & ~3;                                   // whole dwords only
            else if ((kpl & 3) != 1) continue;                                  // dwords + one trailing byte
            if (kpl < 4) kpl = 4;
            if (kpl > 20) continue;
            if (gl == 4 && kpl < 16) continue;  // (maps not instantiated: small D is served by wider groups)
            if (gl == 8 && kpl < 8) continue;
            const int lpw = 64 / gl, tb = kpl & 1, m4 = kpl & ~3;
            const double waves = 2.0 * ((H + lpw - 1) / lpw) + 6.0 * ((W + lpw - 1) / lpw);
            const double w = waves / 1024.0 > 1.0 ? waves / 1024.0 : 1.0;
            const int nstore = (m4 == 20 ? 2 : 1) + tb;                         // 20 bytes leave as 16 + 4
            const int nvmem = (kpl * nw + 3) / 4 + 1 + nstore;                  // code loads + left code + stores
            const double issue = 1.75 * (70.0 + 9.5 * kpl * (nw > 1 ? 1.25 : 1.0) + nvmem + 10.0 * tb);
            const double memory = nvmem * (gl == 16 ? 55.0 : gl == 8 ? 85.0 : 120.0);
            const double latency = 250.0 + 60.0 * kpl + 100.0 * tb;
            double step = w * (issue > memory ? issue : memory);
            if (step < latency) step = latency;
            if (tb) step *= 1.08;
            if (step < best * (gl == 16 ? 1.0 : 0.95)) { best = step; *gl_out = gl; *kpl_out = kpl; }
        }
    }
}

// one switch for every (gl, kpl) pair the kernels exist for
#define PMX_FUSED_MAPS(X)                                                                          \
    X(16, 4) X(16, 5) X(16, 8) X(16, 9) X(16, 12) X(16, 13) X(16, 16) X(16, 17) X(16, 20)          \
    X(8, 8) X(8, 9) X(8, 12) X(8, 13) X(8, 16) X(8, 17) X(8, 20)                                   \
    X(4, 16) X(4, 17) X(4, 20)

int pmx_launch_sgm_fused(pmx_ctx* ctx, pmx_cv* cv, float P1, float P2, float invalid_cost) {
    const int H = cv->H, W = cv->W;
    int gl, kpl;
    fused_choose_map(ctx, H, W, cv->D, (cv->win * cv->win + 31) / 32, &gl, &kpl);
    {
        // the packed-arithmetic kernels (k_sgm8.hip) are ~25 % faster than any map of the kernel below and exist for
        // 16 lanes x whole dwords: unless a map is forced, take that one
        const char* e8 = pmx_opt(ctx, "SGM8");
        if (!pmx_opt(ctx, "FUSED_MAP") && !(e8 && e8[0] == '0')) {
            gl = 16;
            kpl = ((cv->D / 16 + 1) + 3) & ~3;
        }
    }
    // the packed-arithmetic path (k_sgm8.hip: byte costs + two disparities per register) whenever the map allows it;
    // PMX_SGM8=0 keeps the popcount-fused kernel below (test hook: both stay covered by the parity suite)
    {
        const char* e8 = pmx_opt(ctx, "SGM8");
        const int nw8 = (cv->win * cv->win + 31) / 32;
        if (!(e8 && e8[0] == '0') && pmx_sgm8_supported(gl, kpl, nw8)) {
            int rc8 = pmx_launch_sgm8(ctx, cv, kpl, (uint32_t)P1, (uint32_t)P2, (uint32_t)invalid_cost);
            if (rc8) return rc8;
            cv->repr = PMX_REPR_SGM_U8X8;
            pmx_near_forget(ctx, cv);
            return PMX_OK;
        }
    }
    const int nact = (cv->D + kpl - 1) / kpl;
    const int Dp = (nact * kpl + 3) & ~3;
    const size_t dstride = pmx_dir_stride(H, W, Dp);
    size_t need = 8 * dstride;
    if (cv->ldir_bytes < need) {
        PMX_HIP(hipStreamSynchronize(ctx->stream));
        pmx_pool_free(ctx, cv->ldir);
        cv->ldir = nullptr;
        cv->ldir_bytes = 0;
        PMX_HIP(pmx_pool_alloc(ctx, (void**)&cv->ldir, need + 64));
        cv->ldir_bytes = need;
    }
    cv->Dp = Dp; cv->gl = gl; cv->kpl = kpl; cv->dstride = dstride; cv->nvol = 8;
    const int nw = (cv->win * cv->win + 31) / 32;
    fused_args a;
    a.codeL = cv->codeL;
    a.codeR = cv->codeR;
    a.ldir = cv->ldir;
    a.dstride = dstride;
    a.H = H; a.W = W; a.D = cv->D; a.Dp = Dp; a.d0 = cv->d0; a.o = cv->win / 2;
    a.P1 = (uint32_t)P1; a.P2 = (uint32_t)P2; a.invalid_cost = (uint32_t)invalid_cost;
    a.nact = nact;
    const int lpw = 64 / gl;
    const int nwaves = 2 * ((H + lpw - 1) / lpw) + 6 * ((W + lpw - 1) / lpw);
    dim3 grid((nwaves + kWavesPerBlock - 1) / kWavesPerBlock), block(kWavesPerBlock * 64);
    bool launched = false;
    {
        pmx_stage_scope t(ctx, PMX_STAGE_SGM_FUSED);
#define PMX_FUSED_CASE(GLV, KPLV)                                                                                      \
    if (!launched && gl == GLV && kpl == KPLV) {                                                                       \
        if (nw == 1) hipLaunchKernelGGL(HIP_KERNEL_NAME(sgm_census_fused_kernel<1, GLV, KPLV>), grid, block, 0, ctx->stream, a); \
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(sgm_census_fused_kernel<2, GLV, KPLV>), grid, block, 0, ctx->stream, a);         \
        launched = true;                                                                                               \
    }
        PMX_FUSED_MAPS(PMX_FUSED_CASE)
#undef PMX_FUSED_CASE
    }
    PMX_CHECK(launched, PMX_ERR_STATE, "pmx_sgm (fused): no kernel for lane map %dx%d", gl, kpl);
    PMX_HIP(hipGetLastError());
    cv->repr = PMX_REPR_SGM_U8X8;
    pmx_near_forget(ctx, cv);  // the volume changed under the cache
    return PMX_OK;
}

int pmx_launch_sum8_wta(pmx_ctx* ctx, const pmx_cv* cv, float invalid_disparity) {
    size_t npix = (size_t)cv->H * cv->W;
    const int ppw = 64 / cv->gl;
    size_t want = (npix + 4 * ppw - 1) / (4 * ppw);  // ppw pixels per wave, 4 waves per block
    int grid = (int)(want < 65536 ? want : 65536);
    bool launched = false;
    {
        pmx_stage_scope t(ctx, PMX_STAGE_WTA);
#define PMX_WTA_CASE(GLV, KPLV)                                                                                         \
    if (!launched && cv->nvol == 8 && cv->gl == GLV && cv->kpl == KPLV) {                                              \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(sum8_wta_kernel<GLV, KPLV, 8>), dim3(grid), dim3(256), 0, ctx->stream, make_sum8(cv), \
                           npix, (double)cv->d0, invalid_disparity, ctx->disp, ctx->validity, (float4*)ctx->near);     \
        launched = true;                                                                                                \
    }
        PMX_FUSED_MAPS(PMX_WTA_CASE)
#undef PMX_WTA_CASE
        // three volumes, no per-pixel ranges: the leaner kernel (PMX_WTA3=0: the general one, A/B hook)
        const char* e3 = pmx_opt(ctx, "WTA3");
        if (!launched && cv->nvol == 3 && cv->gl == 16 && !cv->has_range && cv->kpl % 4 == 0 && cv->kpl <= 20 && !(e3 && e3[0] == '0')) {
            const int qpw = 16;
            const int quads_row = (cv->W + 3) / 4;
            const dim3 g3((unsigned)(((quads_row + qpw - 1) / qpw + 3) / 4), (unsigned)cv->H);
#define PMX_WTA3L(KPLV)                                                                                                       \
    case KPLV:                                                                                                               \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(sum3_wta_kernel<KPLV>), g3, dim3(256), 0, ctx->stream, make_sum8(cv), qpw, (double)cv->d0, \
                           invalid_disparity, ctx->disp, ctx->validity, (float4*)ctx->near);                                   \
        break
            switch (cv->kpl) { PMX_WTA3L(4); PMX_WTA3L(8); PMX_WTA3L(12); PMX_WTA3L(16); PMX_WTA3L(20); }
#undef PMX_WTA3L
            launched = true;
        }
#define PMX_WTA3_CASE(KPLV)                                                                                             \
    if (!launched && cv->nvol == 3 && cv->gl == 16 && cv->kpl == KPLV) {                                               \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(sum8_wta_kernel<16, KPLV, 3>), dim3(grid), dim3(256), 0, ctx->stream, make_sum8(cv), \
                           npix, (double)cv->d0, invalid_disparity, ctx->disp, ctx->validity, (float4*)ctx->near);     \
        launched = true;                                                                                                \
    }
        PMX_WTA3_CASE(4) PMX_WTA3_CASE(8) PMX_WTA3_CASE(12) PMX_WTA3_CASE(16) PMX_WTA3_CASE(20)
#undef PMX_WTA3_CASE
    }
    PMX_CHECK(launched, PMX_ERR_STATE, "pmx_wta (fused): no kernel for lane map %dx%d", cv->gl, cv->kpl);
    PMX_HIP(hipGetLastError());
    ctx->near_owner = cv;
    return PMX_OK;
}

int pmx_launch_sum8_refine(pmx_ctx* ctx, const pmx_cv* cv, int method) {
    size_t npix = (size_t)cv->H * cv->W;
    pmx_stage_scope t(ctx, PMX_STAGE_REFINE);
    hipLaunchKernelGGL(sum8_refine_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, ctx->stream, make_sum8(cv), npix,
                       (double)cv->d0, (double)cv->d0 + (double)(cv->D - 1), method, ctx->disp, ctx->validity, ctx->itp,
                       ctx->near_owner == cv ? (const float4*)ctx->near : (const float4*)nullptr);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

int pmx_launch_sum8_to_float(pmx_ctx* ctx, pmx_cv* cv) {
    dim3 grid((cv->W * cv->D + 255) / 256, cv->H);
    hipLaunchKernelGGL(sum8_to_float_kernel, grid, dim3(256), 0, ctx->stream, make_sum8(cv), cv->data);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

int pmx_launch_census_nan_pixels(pmx_ctx* ctx, const pmx_cv* cv, uint8_t* dev_out) {
    dim3 grid((cv->W + 255) / 256, cv->H);
    hipLaunchKernelGGL(census_nan_pixels_kernel, grid, dim3(256), 0, ctx->stream, make_sum8(cv), dev_out);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}
