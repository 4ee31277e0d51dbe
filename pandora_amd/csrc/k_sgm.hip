// k_sgm.hip - 8-path semi-global matching on the device-resident float32 cost volume.  gfx950.
//
// Definition (this build's own; the reference delegates to pandora_plugin_libsgm, see DESIGN.md
// and oracle/oracle.c orc_sgm, which this file matches operation for operation):
//   C'(p,d) = C(p,d) (negated for "max" measures), NaN -> invalid_cost
//   L_r(p,d) = C' + ( min( L_r(p-r,d), min(L_r(p-r,d-1), L_r(p-r,d+1)) + P1, M + P2 ) - M ),
//              M = min_k L_r(p-r,k);  L_r = C' on the first pixel of a path
//   S = (S_H + S_D) + S_U, family by family (round 6; one running sum over the eight paths until then):
//       S_H = L(0,+1) + L(0,-1),  S_D = (L(+1,0) + L(+1,+1)) + L(+1,-1),  S_U = (L(-1,0) + L(-1,+1)) + L(-1,-1)
//       - a family without a path (direction masks) is left out.  Family-wise sums are what lets k_sgmfam.hip run the downward
//       family beside the horizontal pair instead of behind it.
//
// Three schedules compute that definition bit for bit (pmx_launch_sgm picks one by size, PMX_SGM_SCHED=seq|par|fam forces):
//   seq  one launch per direction (this file): the first family accumulates into S, a later one into a second volume T whose
//        last pass writes S = S + (T + L): 92 B/cell of HBM traffic as before
//   par  the eight directions side by side into eight path volumes + the family-wise sum (this file): small volumes
//   fam  two horizontal passes (this file) + two fused three-direction marching passes (k_sgmfam.hip): ~46 B/cell
//
// Execution model: ONE WAVEFRONT PER SCANLINE.  The 64 lanes of a wave hold the D path costs of the
// current pixel, KPL consecutive disparities per lane (blocked layout -> the d-1/d+1 neighbours are
// in-register except at the lane edges, fetched with DPP wave shifts); min over D is a DPP
// row_shr/row_bcast reduction, no LDS.  A wave walks its line pixel by pixel; the cost-volume
// reads of the next PF pixels are already in flight (register ring), so the recurrence latency
// and the HBM latency overlap.  Diagonal lines wrap around the image width (the path restarts at
// the wrap), which gives exactly W equally long lines per diagonal direction.
//
// Per pass the kernel reads C (4 B/cell), reads S (4) and writes S (4); the first pass skips the
// S read, the last pass also restores NaN / un-negates.  No MFMA: this is an HBM-bound scan.
#include <cstdlib>

#include "pmx_buf.h"
#include "pmx_internal.h"

static constexpr int kWavesPerBlock = 4;
static constexpr int kPF = 8;  // pixels of read-ahead per wave (register ring)

__device__ __forceinline__ float f_inf() { return __int_as_float(0x7f800000); }
__device__ __forceinline__ float f_nan() { return __int_as_float(0x7fc00000); }
__device__ __forceinline__ float fmin2(float a, float b) { return a < b ? a : b; }
// This file is compiled with -fno-honor-nans (Makefile): no NaN ever reaches a floating-point instruction of the recurrences - an
// invalid cost is replaced by `invalid_cost` before it is used, lanes without a disparity carry +inf - so the minima are single
// v_min_f32 / v_min3_f32 (with NaNs honoured every `a < b ? a : b` is a compare and a select, and the line kernels are bound by
// their vector instructions: 86 per step of the checkpoint pass).  What IS a NaN is recognised by its bits.
__device__ __forceinline__ bool is_nan_bits(float v) { return (__float_as_uint(v) & 0x7fffffffu) > 0x7f800000u; }
// NaN where `c`.  Under -fno-honor-nans a select that may yield a NaN CONSTANT is fair game for the optimiser (it was folded away,
// bit casts or not), so the NaN's bits come in as a kernel argument (`nan_bits` = 0x7fc00000, set by the launchers).
__device__ __forceinline__ float nan_where(bool c, float v, uint32_t nan_bits) { return __uint_as_float(c ? nan_bits : __float_as_uint(v)); }

template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ float dpp_mov(float oldv, float src) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(oldv), __float_as_int(src), CTRL, ROW_MASK,
                                                      BANK_MASK, false));
}

// lane l receives the value of lane l-1 (lane 0 keeps `fill`)
__device__ __forceinline__ float from_lane_below(float v, float fill) { return dpp_mov<0x138>(fill, v); }  // wave_shr:1
// lane l receives the value of lane l+1 (lane 63 keeps `fill`)
__device__ __forceinline__ float from_lane_above(float v, float fill) { return dpp_mov<0x130>(fill, v); }  // wave_shl:1

// minimum over the 64 lanes, returned wave-uniform.  Six in-place v_min_f32_dpp (a lane whose source lane does not exist keeps its
// value): written as assembly because the compiler's form of `min(v, dpp(v))` is a copy, a v_mov_dpp and a v_min per stage, and
// the line kernels are bound by their vector instructions.  (s_nop 1: the two wait states a DPP source needs after a VALU write -
// the hazard recogniser does not look into inline assembly.)
__device__ __forceinline__ float wave_min(float v) {
    asm("s_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"  // lane 15 of every row holds the row's minimum
        "s_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"  // into rows 1, 3
        "s_nop 1\n\t"
        "v_min_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"  // into rows 2, 3: lane 63 holds the minimum
        "s_nop 1"
        : "+v"(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// MODE bits of a pass: it adds to what S already holds / it is the last pass and finishes S (overcounting, sign, NaN) / it is the
// last pass of a family that is not the first one: S = S + (T + L), T the sum of the family's earlier paths
enum { SGM_READS_S = 1, SGM_EPILOGUE = 2, SGM_READS_T = 4 };

struct sgm_args {
    const float* C;  // raw cost volume (NaN = invalid)
    float* S;        // accumulator / output
    const float* T;  // SGM_READS_T: the family's earlier paths
    int H, W, D;
    int dr, dc;      // step from p-r to p
    float P1, P2, invalid_cost;
    int is_max, overcounting;
    // all eight directions in ONE launch (blockIdx.y = direction), each writing its own path-cost volume S + y * vol:
    // for volumes too small to fill the GPU one direction at a time (see pmx_launch_sgm)
    int multi;
    size_t vol;
    int mask;  // multi: directions (bits, definition order) that are wanted; the others' blocks exit at once
    // penalty methods that follow the image (plugin_libsgm.rst:20-27): P2 of every pixel for the direction of this launch
    // ([H][W] float32; multi: [8][H][W]); nullptr = the constant a.P2
    const float* p2map;
    uint32_t nan_bits;  // 0x7fc00000 (see nan_where)
};

// Walks one line.  Every access of the line's pixel is a raw buffer instruction on a descriptor of the pixel's image ROW (the
// wave is on one row per step, so the descriptor lives in scalar registers): the lane's KPL floats move as 16 / 8 / 4-byte
// pieces, the last lane of a pixel (fewer than KPL disparities when D is not a multiple of KPL) and the lanes beyond it are
// handled by the range check of the instruction (pmx_buf.h), reads past the row return 0.
template <int KPL, int MODE>
__global__ __launch_bounds__(kWavesPerBlock * 64) void sgm_path_kernel(sgm_args a) {
    if (a.multi) {  // direction order of the definition above
        const int k = blockIdx.y;
        a.dr = (k < 2) ? 0 : (k < 5 ? 1 : -1);
        a.dc = (k == 0) ? 1 : (k == 1) ? -1 : (k == 2 || k == 5) ? 0 : ((k == 3 || k == 6) ? 1 : -1);
        a.S += (size_t)k * a.vol;
        if (a.p2map) a.p2map += (size_t)k * a.H * a.W;
        if (!(a.mask >> k & 1)) return;
    }
    const int lane = threadIdx.x & 63;
    const int line = blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool horizontal = (a.dr == 0);
    const int nlines = horizontal ? a.H : a.W;
    if (line >= nlines) return;
    const int nsteps = horizontal ? a.W : a.H;
    const int D = a.D;
    const int d_first = lane * KPL;
    const int nv = d_first >= D ? 0 : (D - d_first < KPL ? D - d_first : KPL);  // disparities this lane owns
    const bool is_tail = nv > 0 && nv < KPL;
    const int tailn = D % KPL;  // uniform
    using P = pieces<KPL>;
    int cov = 0;  // uniform: what the wide pieces cover of the tail lane
#pragma unroll
    for (int i = 0; i < P::N; ++i)
        if (P::start(i) + P::width(i) <= tailn) cov = P::start(i) + P::width(i);
    const int rem = tailn - cov;
    const unsigned row_bytes = (unsigned)a.W * (unsigned)D * 4u, pix_bytes = (unsigned)D * 4u;
    const unsigned lane_load = (unsigned)(nv ? d_first : 0) * 4u;  // lanes without a disparity read the pixel's d = 0
    const unsigned lane_store = nv ? (unsigned)d_first * 4u : kOob;
    // (row pointers advance with the cursors: a descriptor from `vol + r * W * D` at every step was twenty scalar instructions per
    //  memory instruction, and on small volumes - a few wavefronts per SIMD, every one a chain of dependent steps - the kernel is
    //  bound by its instruction count: cones, eight directions side by side, 0.29 ms of the 0.52 of BASELINE configs[1])
    auto rsrc_of = [&](const float* row) { return __builtin_amdgcn_make_buffer_rsrc((void*)row, 0, row_bytes, kRsrcWord3); };
    const ptrdiff_t row_step = (ptrdiff_t)a.dr * (ptrdiff_t)a.W * (ptrdiff_t)D;  // floats from a line's pixel to its next one's row

    // cursor of the pixel being computed and of the pixel being prefetched (wave-uniform)
    int r = horizontal ? line : (a.dr > 0 ? 0 : a.H - 1);
    int c = horizontal ? (a.dc > 0 ? 0 : a.W - 1) : line;
    int pr = r, pc = c;
    const float* c_pre = a.C + (size_t)r * a.W * D;  // rows of the prefetch cursor ...
    const float* s_pre = a.S + (size_t)r * a.W * D;
    const float* t_pre = (MODE & SGM_READS_T) ? a.T + (size_t)r * a.W * D : nullptr;
    float* s_row = a.S + (size_t)r * a.W * D;        // ... and of the pixel being computed

    float cbuf[kPF][KPL], sbuf[kPF][KPL];
    float tbuf[(MODE & SGM_READS_T) ? kPF : 1][KPL];
    float p2buf[kPF];  // the pixel's P2 when it varies (rides in the same ring as its costs)
    const bool var_p2 = a.p2map != nullptr;  // (uniform)
    int pleft = nsteps - 1;  // steps the prefetch cursor may still advance (read-ahead past the end re-reads the last pixel)
    auto prefetch = [&](float (&cslot)[KPL], float (&sslot)[KPL], float (&tslot)[KPL], float& p2slot) {
        const unsigned off = (unsigned)pc * pix_bytes + lane_load;
        buf_load<KPL>(rsrc_of(c_pre), off, cslot);
        if (MODE & SGM_READS_S) buf_load<KPL>(rsrc_of(s_pre), off, sslot);
        if (MODE & SGM_READS_T) buf_load<KPL>(rsrc_of(t_pre), off, tslot);
        if (var_p2) p2slot = a.p2map[(size_t)pr * a.W + pc];
        if (pleft > 0) {
            --pleft;
            pr += a.dr;
            pc += a.dc;
            c_pre += row_step;
            s_pre += row_step;
            if (MODE & SGM_READS_T) t_pre += row_step;
            if (!horizontal) { if (pc >= a.W) pc = 0; else if (pc < 0) pc = a.W - 1; }
        }
    };
#pragma unroll
    for (int i = 0; i < kPF; ++i) prefetch(cbuf[i], sbuf[i], tbuf[(MODE & SGM_READS_T) ? i : 0], p2buf[i]);

    float Lp[KPL];  // path costs of the previous pixel (+inf on padded disparities)
#pragma unroll
    for (int k = 0; k < KPL; ++k) Lp[k] = k < nv ? 0.f : f_inf();
    float M = 0.f;  // min_k Lp; (Lp = 0, M = 0) reproduces L = C' on the first pixel of a path

    auto step = [&](float (&cslot)[KPL], float (&sslot)[KPL], float (&tslot)[KPL], float& p2slot) {
        // neighbours across the lane boundary
        const float below = from_lane_below(Lp[KPL - 1], f_inf());
        const float above = from_lane_above(Lp[0], f_inf());
        const float mp2 = M + (var_p2 ? p2slot : a.P2);
        float Ln[KPL], out[KPL];
        float lmin = f_inf();
#pragma unroll
        for (int k = 0; k < KPL; ++k) {
            float cr = cslot[k];
            float cc = is_nan_bits(cr) ? a.invalid_cost : (a.is_max ? -cr : cr);
            float lo = (k > 0) ? Lp[k - 1] : below;
            float hi = (k < KPL - 1) ? Lp[k + 1] : above;
            float nb = fmin2(lo, hi) + a.P1;
            float t = fmin2(Lp[k], nb);
            t = fmin2(t, mp2);
            float l = cc + (t - M);
            Ln[k] = k < nv ? l : f_inf();
            lmin = fmin2(lmin, Ln[k]);
            float s = (MODE & SGM_READS_T) ? (sslot[k] + (tslot[k] + l)) : (MODE & SGM_READS_S) ? (sslot[k] + l) : l;
            if (MODE & SGM_EPILOGUE) {
                if (a.overcounting) s = s - 7.0f * cc;
                if (a.is_max) s = -s;
                s = nan_where(is_nan_bits(cr), s, a.nan_bits);
            }
            out[k] = s;
        }
        buf_store<KPL>(rsrc_of(s_row), (unsigned)c * pix_bytes + lane_store, nv, is_tail, cov, rem, out);
        // refill this ring slot with pixel i + kPF.  Issued AFTER the slot's last use so the new data
        // lands in the same registers (no copy, hence no wait, at the loop back-edge).
        prefetch(cslot, sslot, tslot, p2slot);
        M = wave_min(lmin);
#pragma unroll
        for (int k = 0; k < KPL; ++k) Lp[k] = Ln[k];
        // advance; a diagonal line that leaves the image re-enters on the other side and the path
        // restarts there (border initialisation)
        r += a.dr;
        c += a.dc;
        s_row += row_step;
        if (!horizontal && (c >= a.W || c < 0)) {
            c = (c >= a.W) ? 0 : a.W - 1;
#pragma unroll
            for (int k = 0; k < KPL; ++k) Lp[k] = k < nv ? 0.f : f_inf();
            M = 0.f;
        }
    };

    int i = 0;
    for (; i + kPF <= nsteps; i += kPF) {
#pragma unroll
        for (int j = 0; j < kPF; ++j) step(cbuf[j], sbuf[j], tbuf[(MODE & SGM_READS_T) ? j : 0], p2buf[j]);
    }
#pragma unroll
    for (int j = 0; j < kPF - 1; ++j)
        if (i + j < nsteps) step(cbuf[j], sbuf[j], tbuf[(MODE & SGM_READS_T) ? j : 0], p2buf[j]);
}

// (drow, dcol) of the step from p-r to p, in the definition's order
static const int kSgmDirs[8][2] = {{0, 1}, {0, -1}, {1, 0}, {1, 1}, {1, -1}, {-1, 0}, {-1, 1}, {-1, -1}};

// one launch of the line kernel: direction k, MODE bits chosen by the caller
template <int KPL>
static void sgm_launch_direction(pmx_ctx* ctx, const sgm_args& base, int k, int mode) {
    sgm_args a = base;
    a.dr = kSgmDirs[k][0];
    a.dc = kSgmDirs[k][1];
    if (a.p2map) a.p2map += (size_t)k * a.H * a.W;
    int nlines = a.dr == 0 ? a.H : a.W;
    dim3 grid((nlines + kWavesPerBlock - 1) / kWavesPerBlock);
    dim3 block(kWavesPerBlock * 64);
    pmx_stage_scope t(ctx, PMX_STAGE_SGM_PATH);
#define PMX_SGM_LAUNCH(MODE) hipLaunchKernelGGL(HIP_KERNEL_NAME(sgm_path_kernel<KPL, MODE>), grid, block, 0, ctx->stream, a)
    switch (mode) {
        case 0: PMX_SGM_LAUNCH(0); break;
        case SGM_READS_S: PMX_SGM_LAUNCH(SGM_READS_S); break;
        case SGM_EPILOGUE: PMX_SGM_LAUNCH(SGM_EPILOGUE); break;
        case SGM_READS_S | SGM_EPILOGUE: PMX_SGM_LAUNCH(SGM_READS_S | SGM_EPILOGUE); break;
        case SGM_READS_S | SGM_READS_T: PMX_SGM_LAUNCH(SGM_READS_S | SGM_READS_T); break;
        default: PMX_SGM_LAUNCH(SGM_READS_S | SGM_READS_T | SGM_EPILOGUE); break;
    }
#undef PMX_SGM_LAUNCH
}

// families of the definition's sum: paths [kFamFirst[f], kFamFirst[f + 1])
static const int kFamFirst[4] = {0, 2, 5, 8};
static inline int sgm_family_of(int k) { return k < 2 ? 0 : (k < 5 ? 1 : 2); }
static inline int sgm_family_bits(int mask, int f) { return mask & (((1 << kFamFirst[f + 1]) - 1) & ~((1 << kFamFirst[f]) - 1)); }

// Where the pass of direction k accumulates when the directions of `mask` run one after the other, and its MODE bits.  The first
// family of the mask sums in S itself; a later family with one path adds it to S; a later family with several sums its earlier
// paths in T and its last pass writes S = S + (T + L).  *into_t: the pass reads and writes T in S's place.
static int sgm_pass_mode(int mask, int k, bool* into_t) {
    const int f = sgm_family_of(k);
    const int fam = sgm_family_bits(mask, f);
    const bool first_family = (mask & ((1 << kFamFirst[f]) - 1)) == 0;
    const int before = fam & ((1 << k) - 1), after_in_family = fam >> (k + 1), after = mask >> (k + 1);
    *into_t = false;
    const int epi = after ? 0 : SGM_EPILOGUE;
    if (first_family) return (before ? SGM_READS_S : 0) | epi;
    if (after_in_family) {  // an earlier path of a later family: T
        *into_t = true;
        return before ? SGM_READS_S : 0;
    }
    return SGM_READS_S | (before ? SGM_READS_T : 0) | epi;
}

// does a sequential run of `mask` need the second accumulator?
static bool sgm_seq_needs_t(int mask) {
    for (int f = 1; f < 3; ++f) {
        const int fam = sgm_family_bits(mask, f);
        if ((fam & (fam - 1)) != 0 && (mask & ((1 << kFamFirst[f]) - 1)) != 0) return true;
    }
    return false;
}

template <int KPL>
static int sgm_run_horizontal(pmx_ctx* ctx, const sgm_args& base, int mask);

// one pass of a sequential run: the accumulator the definition gives direction k (S, or T for the earlier paths of a later family)
template <int KPL>
static void sgm_launch_pass(pmx_ctx* ctx, const sgm_args& base, int mask, int k) {
    bool into_t = false;
    const int mode = sgm_pass_mode(mask, k, &into_t);
    sgm_args a = base;
    if (into_t) a.S = const_cast<float*>(base.T);
    sgm_launch_direction<KPL>(ctx, a, k, mode);
}

template <int KPL>
static int sgm_run(pmx_ctx* ctx, const sgm_args& base, int mask) {
    int rc = sgm_run_horizontal<KPL>(ctx, base, mask);  // the pair fused when both are wanted
    if (rc) return rc;
    for (int k = 2; k < 8; ++k)
        if (mask >> k & 1) sgm_launch_pass<KPL>(ctx, base, mask, k);
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

// ---- the horizontal pair in 12.5 B/cell instead of 20 ---------------------------------------------------------------------
// (0,+1) then (0,-1) as two line passes cost R C + W S, then R C + R S + W S.  The second read-modify-write of S exists only
// because a row's (0,+1) costs are long gone when the (0,-1) wave comes back through it - and a row's worth of path costs
// (W x D floats) fits no on-chip memory.  What fits is a SEGMENT: the forward pass keeps nothing but the state of its path at
// the end of every kSegCols-th column (checkpoint pass: R C, W 4/kSegCols B/cell); the backward pass walks the row right to
// left one segment at a time: it reloads the checkpoint before the segment, RE-COMPUTES the forward path costs of the segment
// into registers from the segment's costs (the same float32 operations on the same inputs: the same bits), then runs the
// backward recurrence through the segment and stores S = L(0,+1) + L(0,-1) once (R C, W S).  One extra direction-pass of
// arithmetic buys 7.5 B/cell of HBM traffic; both kernels stay HBM-bound.
static constexpr int kSegCols = 8;

// one SGM step: new costs from the previous pixel's (Lp, M).  Lanes without the disparity carry +inf.
template <int KPL>
__device__ __forceinline__ float sgm_line_step(const float (&Lp)[KPL], float M, const float (&cc)[KPL], int nv, float P1, float P2,
                                               float (&Ln)[KPL]) {
    const float below = from_lane_below(Lp[KPL - 1], f_inf());
    const float above = from_lane_above(Lp[0], f_inf());
    const float mp2 = M + P2;
    float lmin = f_inf();
#pragma unroll
    for (int k = 0; k < KPL; ++k) {
        const float lo = (k > 0) ? Lp[k - 1] : below;
        const float hi = (k < KPL - 1) ? Lp[k + 1] : above;
        const float nb = fmin2(lo, hi) + P1;
        float t = fmin2(Lp[k], nb);
        t = fmin2(t, mp2);
        const float l = cc[k] + (t - M);
        Ln[k] = k < nv ? l : f_inf();
        lmin = fmin2(lmin, Ln[k]);
    }
    return wave_min(lmin);
}

struct sgm_h_args {
    const float* C;
    float* S;
    float* ckpt;  // [H][nseg][64 * KPL + 64]: the (0,+1) path's costs after the last column of every segment, then its minimum
    int H, W, D, nseg;
    float P1, P2, invalid_cost;
    int is_max, overcounting, epilogue;
    uint32_t nan_bits;  // 0x7fc00000 (see nan_where)
};

// forward pass: wave = row, keeps only the checkpoints
template <int KPL>
__global__ __launch_bounds__(kWavesPerBlock * 64) void sgm_h_checkpoint_kernel(sgm_h_args a) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (row >= a.H) return;
    const int D = a.D, W = a.W;
    const int d_first = lane * KPL;
    const int nv = d_first >= D ? 0 : (D - d_first < KPL ? D - d_first : KPL);
    const unsigned pix_bytes = (unsigned)D * 4u, lane_load = (unsigned)(nv ? d_first : 0) * 4u;
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void*)(a.C + (size_t)row * W * D), 0, (unsigned)W * pix_bytes, kRsrcWord3);
    float* ck = a.ckpt + ((size_t)row * a.nseg) * (64 * KPL + 64) + lane * KPL;
    float cbuf[kPF][KPL];
    int pc = 0;
    auto prefetch = [&](float (&slot)[KPL]) {
        buf_load<KPL>(rsC, (unsigned)pc * pix_bytes + lane_load, slot);
        if (pc < W - 1) ++pc;
    };
#pragma unroll
    for (int i = 0; i < kPF; ++i) prefetch(cbuf[i]);
    float Lp[KPL];
#pragma unroll
    for (int k = 0; k < KPL; ++k) Lp[k] = k < nv ? 0.f : f_inf();
    float M = 0.f;
    int c = 0;
    auto step = [&](float (&slot)[KPL]) {
        float cc[KPL], Ln[KPL];
#pragma unroll
        for (int k = 0; k < KPL; ++k) {
            const float cr = slot[k];
            cc[k] = is_nan_bits(cr) ? a.invalid_cost : (a.is_max ? -cr : cr);
        }
        M = sgm_line_step<KPL>(Lp, M, cc, nv, a.P1, a.P2, Ln);
#pragma unroll
        for (int k = 0; k < KPL; ++k) Lp[k] = Ln[k];
        if ((c % kSegCols) == kSegCols - 1 && c < W - 1) {  // uniform: the state the next segment starts from
            float* dst = ck + (size_t)(c / kSegCols) * (64 * KPL + 64);
#pragma unroll
            for (int k = 0; k < KPL; ++k) dst[k] = Ln[k];
            if (lane == 0) dst[64 * KPL - lane * KPL] = M;  // (dst is lane-offset: the minimum sits behind the 64 * KPL costs)
        }
        prefetch(slot);
        ++c;
    };
    int i = 0;
    for (; i + kPF <= W; i += kPF) {
#pragma unroll
        for (int j = 0; j < kPF; ++j) step(cbuf[j]);
    }
#pragma unroll
    for (int j = 0; j < kPF - 1; ++j)
        if (i + j < W) step(cbuf[j]);
}

// backward pass: wave = row, right to left, one segment at a time
template <int KPL>
__global__ __launch_bounds__(kWavesPerBlock * 64) void sgm_h_backward_kernel(sgm_h_args a) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (row >= a.H) return;
    const int D = a.D, W = a.W;
    const int d_first = lane * KPL;
    const int nv = d_first >= D ? 0 : (D - d_first < KPL ? D - d_first : KPL);
    const bool is_tail = nv > 0 && nv < KPL;
    const int tailn = D % KPL;
    using P = pieces<KPL>;
    int cov = 0;
#pragma unroll
    for (int i = 0; i < P::N; ++i)
        if (P::start(i) + P::width(i) <= tailn) cov = P::start(i) + P::width(i);
    const int rem = tailn - cov;
    const unsigned pix_bytes = (unsigned)D * 4u, lane_load = (unsigned)(nv ? d_first : 0) * 4u;
    const unsigned lane_store = nv ? (unsigned)d_first * 4u : kOob;
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void*)(a.C + (size_t)row * W * D), 0, (unsigned)W * pix_bytes, kRsrcWord3);
    const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)(a.S + (size_t)row * W * D), 0, (unsigned)W * pix_bytes, kRsrcWord3);
    const float* ck = a.ckpt + ((size_t)row * a.nseg) * (64 * KPL + 64);
    // backward path state
    float Bp[KPL];
#pragma unroll
    for (int k = 0; k < KPL; ++k) Bp[k] = k < nv ? 0.f : f_inf();
    float MB = 0.f;
    // costs of the segment being processed and of the next one to the left (in flight)
    float cur[kSegCols][KPL], nxt[kSegCols][KPL];
    float fwd[kSegCols][KPL];
    auto load_segment = [&](int seg, float (&dst)[kSegCols][KPL]) {  // columns outside the row read as 0 and are never used
#pragma unroll
        for (int j = 0; j < kSegCols; ++j) {
            const int c = seg * kSegCols + j;
            buf_load<KPL>(rsC, (seg >= 0 && c < W) ? (unsigned)c * pix_bytes + lane_load : kOob, dst[j]);
        }
    };
    const int last = a.nseg - 1;
    load_segment(last, cur);
    for (int seg = last; seg >= 0; --seg) {
        load_segment(seg - 1, nxt);
        // forward state entering the segment
        float Fp[KPL], MF;
        if (seg == 0) {
#pragma unroll
            for (int k = 0; k < KPL; ++k) Fp[k] = k < nv ? 0.f : f_inf();
            MF = 0.f;
        } else {
            const float* src = ck + (size_t)(seg - 1) * (64 * KPL + 64);
#pragma unroll
            for (int k = 0; k < KPL; ++k) Fp[k] = src[lane * KPL + k];
            MF = src[64 * KPL];
        }
        const int ncols = (seg == last) ? W - seg * kSegCols : kSegCols;  // uniform
        float cc[kSegCols][KPL];
#pragma unroll
        for (int j = 0; j < kSegCols; ++j) {
#pragma unroll
            for (int k = 0; k < KPL; ++k) {
                const float cr = cur[j][k];
                cc[j][k] = is_nan_bits(cr) ? a.invalid_cost : (a.is_max ? -cr : cr);
            }
        }
        // the (0,+1) path through the segment, recomputed
#pragma unroll
        for (int j = 0; j < kSegCols; ++j) {  // (columns past the end of the row come last here: what they produce is never used)
            MF = sgm_line_step<KPL>(Fp, MF, cc[j], nv, a.P1, a.P2, fwd[j]);
#pragma unroll
            for (int k = 0; k < KPL; ++k) Fp[k] = fwd[j][k];
        }
        // the (0,-1) path, right to left, and the sum
#pragma unroll
        for (int j = kSegCols - 1; j >= 0; --j) {
            // columns past the end of the row (last segment only) come FIRST here: they must leave the path's state alone and
            // store nothing - a uniform select and an out-of-range offset, no branch around the memory instructions
            const bool real = j < ncols;
            float Ln[KPL], out[KPL];
            const float m = sgm_line_step<KPL>(Bp, MB, cc[j], nv, a.P1, a.P2, Ln);
            MB = real ? m : MB;
#pragma unroll
            for (int k = 0; k < KPL; ++k) {
                Bp[k] = real ? Ln[k] : Bp[k];
                float sv = fwd[j][k] + Ln[k];
                if (a.epilogue) {
                    if (a.overcounting) sv = sv - 7.0f * cc[j][k];
                    if (a.is_max) sv = -sv;
                    sv = nan_where(is_nan_bits(cur[j][k]), sv, a.nan_bits);
                }
                out[k] = sv;
            }
            buf_store<KPL>(rsS, real ? (unsigned)(seg * kSegCols + j) * pix_bytes + lane_store : kOob, nv, is_tail, cov, rem, out);
        }
#pragma unroll
        for (int j = 0; j < kSegCols; ++j)
#pragma unroll
            for (int k = 0; k < KPL; ++k) cur[j][k] = nxt[j][k];
    }
}

template <int KPL>
static int sgm_run_horizontal_fused(pmx_ctx* ctx, const sgm_args& base, int mask) {
    sgm_h_args h;
    h.C = base.C; h.S = base.S;
    h.H = base.H; h.W = base.W; h.D = base.D;
    h.nseg = (base.W + kSegCols - 1) / kSegCols;
    h.P1 = base.P1; h.P2 = base.P2; h.invalid_cost = base.invalid_cost;
    h.is_max = base.is_max; h.overcounting = base.overcounting;
    h.nan_bits = base.nan_bits;
    h.epilogue = (mask >> 2) == 0;
    const size_t bytes = (size_t)h.H * h.nseg * (64 * KPL + 64) * sizeof(float);
    float* ck = nullptr;
    PMX_HIP(pmx_pool_alloc(ctx, (void**)&ck, bytes));
    h.ckpt = ck;
    dim3 grid((h.H + kWavesPerBlock - 1) / kWavesPerBlock), block(kWavesPerBlock * 64);
    {
        pmx_stage_scope t(ctx, PMX_STAGE_SGM_PATH);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(sgm_h_checkpoint_kernel<KPL>), grid, block, 0, ctx->stream, h);
    }
    {
        pmx_stage_scope t(ctx, PMX_STAGE_SGM_PATH);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(sgm_h_backward_kernel<KPL>), grid, block, 0, ctx->stream, h);
    }
    pmx_pool_free(ctx, ck);  // stream-ordered reuse
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

// the horizontal pair of the family schedule (k_sgmfam.hip runs the six others)
template <int KPL>
static int sgm_run_horizontal(pmx_ctx* ctx, const sgm_args& base, int mask) {
    // both horizontal paths: the checkpoint / recompute pair above (PMX_SGM_HFUSED=0: test hook for the two line passes)
    const char* e = pmx_opt(ctx, "SGM_HFUSED");
    if ((mask & 3) == 3 && KPL <= 6 && !base.p2map && !(e && e[0] == '0')) return sgm_run_horizontal_fused<KPL>(ctx, base, mask);
    for (int k = 0; k < 2; ++k)
        if (mask >> k & 1) sgm_launch_pass<KPL>(ctx, base, mask, k);  // (the first family: always into S)
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

// S = sum of the enabled path volumes, accumulated in float32 family by family as the definition says (exactly what the sequential
// passes compute), then the epilogue of the last pass: overcounting, sign, NaN where the input was NaN.
__global__ __launch_bounds__(256) void sgm_sum_paths_kernel(const float* __restrict__ C, const float* __restrict__ L, size_t n,
                                                            float invalid_cost, int is_max, int overcounting, int mask,
                                                            uint32_t nan_bits, float* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t step = (size_t)gridDim.x * 256;
    for (; i < n; i += step) {
        float s = 0.f;
        bool any = false;
#pragma unroll
        for (int f = 0; f < 3; ++f) {  // family by family, as the definition adds them
            const int k0 = f == 0 ? 0 : (f == 1 ? 2 : 5), k1 = f == 0 ? 2 : (f == 1 ? 5 : 8);
            float fs = 0.f;
            bool fany = false;
#pragma unroll
            for (int k = k0; k < k1; ++k)
                if (mask >> k & 1) {
                    const float v = L[(size_t)k * n + i];
                    fs = fany ? fs + v : v;
                    fany = true;
                }
            if (fany) {
                s = any ? s + fs : fs;
                any = true;
            }
        }
        const float cr = C[i];
        const float cc = is_nan_bits(cr) ? invalid_cost : (is_max ? -cr : cr);
        if (overcounting) s = s - 7.0f * cc;
        if (is_max) s = -s;
        s = nan_where(is_nan_bits(cr), s, nan_bits);
        out[i] = s;
    }
}

// the eight directions side by side (one launch), then the ordered sum
template <int KPL>
static int sgm_run_parallel(pmx_ctx* ctx, sgm_args a, float* paths, size_t cells, int mask) {
    a.multi = 1;
    a.vol = cells;
    a.S = paths;
    a.mask = mask;
    const int nlines = a.H > a.W ? a.H : a.W;
    dim3 grid((nlines + kWavesPerBlock - 1) / kWavesPerBlock, 8);
    dim3 block(kWavesPerBlock * 64);
    {
        pmx_stage_scope t(ctx, PMX_STAGE_SGM_PATH);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(sgm_path_kernel<KPL, 0>), grid, block, 0, ctx->stream, a);
    }
    {
        pmx_stage_scope t(ctx, PMX_STAGE_SGM_PATH);
        const size_t want = (cells + 255) / 256;
        hipLaunchKernelGGL(sgm_sum_paths_kernel, dim3((unsigned)(want < 65536 ? want : 65536)), dim3(256), 0, ctx->stream, a.C, paths,
                           cells, a.invalid_cost, a.is_max, a.overcounting, mask, a.nan_bits, ctx->scratch);
    }
    PMX_HIP(hipGetLastError());
    return PMX_OK;
}

#define PMX_KPL_SWITCH(kpl, CALL)                                                        \
    switch (kpl) {                                                                       \
        case 1: rc = CALL(1); break;                                                     \
        case 2: rc = CALL(2); break;                                                     \
        case 3: rc = CALL(3); break;                                                     \
        case 4: rc = CALL(4); break;                                                     \
        case 5: rc = CALL(5); break;                                                     \
        case 6: rc = CALL(6); break;                                                     \
        case 7: rc = CALL(7); break;                                                     \
        case 8: rc = CALL(8); break;                                                     \
        default:                                                                         \
            pmx_set_error("pmx_sgm: D = %d not supported (max 512)", cv->D);             \
            return PMX_ERR_UNSUPPORTED;                                                  \
    }

int pmx_launch_sgm(pmx_ctx* ctx, pmx_cv* cv, float P1, float P2, int is_max, float invalid_cost, int overcounting) {
    const int mask = ctx->sgm_dir_mask & 0xff;
    PMX_CHECK(mask != 0, PMX_ERR_ARG, "pmx_sgm: empty direction mask");
    // accumulator volume: the context's scratch (or, for a deferred last pass, the handle's own partial-sum volume)
    size_t bytes = cv->cells() * sizeof(float) + 256;
    int rc = PMX_OK;
    if (cv->bytes < bytes) {
        // the input volume needs the same tail padding for the KPL-wide loads of its last pixel
        float* grown = nullptr;
        PMX_HIP(pmx_pool_alloc(ctx, (void**)&grown, bytes));
        PMX_HIP(hipMemcpyAsync(grown, cv->data, cv->cells() * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
        PMX_HIP(hipStreamSynchronize(ctx->stream));
        pmx_pool_free(ctx, cv->data);
        cv->data = grown;
        cv->bytes = bytes;
    }
    sgm_args a;
    a.C = cv->data;
    a.S = nullptr;  // set once the schedule is known
    a.T = nullptr;
    a.H = cv->H; a.W = cv->W; a.D = cv->D;
    a.dr = 0; a.dc = 0;
    a.P1 = P1; a.P2 = P2; a.invalid_cost = invalid_cost;
    a.is_max = is_max; a.overcounting = overcounting;
    a.multi = 0; a.vol = 0; a.mask = mask;
    a.p2map = ctx->sgm_p2maps;  // set by pmx_sgm_p2maps for the duration of its call
    a.nan_bits = 0x7fc00000u;
    const int kpl = (cv->D + 63) / 64;
    // Schedule.  Small volumes cannot fill the GPU one direction at a time (a wave per scanline: 375 - 450 waves on cones, each a
    // chain of dependent steps): "par" runs the eight directions side by side into eight path volumes and adds them in the
    // definition's order - the same float32 operations, 104 B/cell of traffic, 8 volumes of scratch.  Large volumes are HBM-bound:
    // "fam" reads the costs four times instead of eight and rewrites S three times instead of seven (~46 B/cell against the 92
    // of "seq", one launch per direction).  PMX_SGM_SCHED=seq|par|fam forces one (test hook; PMX_SGM_PAR=0/1 is the round-1 spelling).
    enum { SEQ, PAR, FAM } sched = SEQ;
    // (break-even measured ~2e8 cells in round 1, tools/bench_sgm_float.py; on round 6's kernels 7e7 - 1e8: 600 x 800 x 129 1.72 against
    //  1.98 ms and 900 x 1200 x 65 2.09 against 2.51 for "par", 800 x 1000 x 90 equal, 800 x 1000 x 129 2.94 against 2.66 and
    //  600 x 800 x 257 3.33 against 2.91 for "seq": profiles/r06_float_sched_rule.txt)
    if (cv->cells() <= ((size_t)96 << 20)) sched = PAR;
    // the marching passes advance one image row per ~2.5 - 4 us whatever the width: they pay from ~3500 columns on (a window
    // for every CU); measured 4096^2 x 257: 55 ms against 96, 10000^2 x 129: 131 against 265, 2048^2 x 129: 12.4 against 11.3
    // Round 6, re-measured on the round's marching kernel (profiles/r06_float_sched_rule.txt, tools/sweep_float_sched.py): the bound of
    // round 3 (3584 columns, 512 rows) had become far too careful - 2048 x 2600 x 129: 12.9 against 15.0 ms, 3000^2 x 129: 19.3
    // against 25.7, 1500 x 3000 x 257: 14.3 against 23.1, 2048 x 3072 x 257: 19.4 against 31.5, 400 rows x 4096 x 257: 7.9 against
    // 11.0; 2048^2 x 129 stays with "seq" (11.9 against 12.7), 1500 x 2000 x 65 too (4.8 against 6.8).
    // A marching pass costs time per image ROW (more of it the more disparities a lane holds), one launch per path time per CELL: a row
    // has to hold enough cells - measured break-evens W D = 150 000 at D = 33 (4096 columns: 7.8 ms "seq" against 8.6, 6000: 11.7
    // against 9.4), 205 000 at D = 49 ... 80 (4096 x 49: 9.4 against 10.5, 3500 x 65: 11.4 against 10.0), ~250 000 at D = 100, ~290 000 at
    // D = 129: 90 000 + 23 000 per disparity of the 16-lane map's lane.  (2048 x 2600 x 33: 5.5 ms against 8.3 the wrong way.)
    else if (cv->W >= 2400 && cv->H >= 384 && pmx_sgm_family_supported(ctx, cv) &&
             (size_t)cv->W * cv->D >= (size_t)90000 + 23000u * (cv->D <= 48 ? 3u : cv->D <= 80 ? 5u : cv->D <= 112 ? 7u : 9u))
        sched = FAM;
    if (const char* e = pmx_opt(ctx, "SGM_PAR")) sched = e[0] == '1' ? PAR : SEQ;
    if (const char* e = pmx_opt(ctx, "SGM_SCHED")) {
        if (e[0] == 's') sched = SEQ;
        else if (e[0] == 'p') sched = PAR;
        else if (e[0] == 'f') sched = FAM;
    }
    if (sched == PAR && kpl > 8) sched = SEQ;
    if (sched == FAM && !pmx_sgm_family_supported(ctx, cv)) sched = SEQ;
    if (a.p2map && sched == FAM) sched = SEQ;  // the marching kernels take the constant penalty only
    const char* ep_ = pmx_opt(ctx, "SGM_PENDING");
    const bool defer_ = sched == FAM && ctx->lazy && mask == 0xff && !(ep_ && ep_[0] == '0');
    if (!defer_) {
        rc = pmx_need_scratch(ctx, bytes);
        if (rc) return rc;
        a.S = ctx->scratch;
    }
    if (sched == PAR) {
        float* paths = nullptr;
        PMX_HIP(pmx_pool_alloc(ctx, (void**)&paths, 8 * cv->cells() * sizeof(float) + 256));
#define PMX_CALL(K) sgm_run_parallel<K>(ctx, a, paths, cv->cells(), mask)
        PMX_KPL_SWITCH(kpl, PMX_CALL)
#undef PMX_CALL
        pmx_pool_free(ctx, paths);  // stream-ordered reuse: the kernels above are queued on ctx->stream
    } else if (sched == FAM) {
        // Lazy mode, all eight paths: the horizontal pair and the downward family run now into the handle's own partial-sum
        // volumes; the upward family waits for whoever comes next - pmx_wta runs it in WTA mode (S is never written, the WTA's
        // read of it never happens: -8 B/cell), anything else runs it in store mode (pmx_sgm_finish_pending).
        const bool defer = defer_;  // (PMX_SGM_PENDING=0: test hook, always finish at once)
        if (defer) {
            if (cv->spart_bytes < bytes) {
                pmx_pool_free(ctx, cv->spart);
                cv->spart = nullptr;
                cv->spart_bytes = 0;
                PMX_HIP(pmx_pool_alloc(ctx, (void**)&cv->spart, bytes));
                cv->spart_bytes = bytes;
            }
            a.S = cv->spart;
        }
        const int hb = mask & 3, db = (mask >> 2) & 7, ub = (mask >> 5) & 7;
        // Round 6: with family-wise sums the downward family reads no earlier sum - it CAN write S_D into a volume of its own and run
        // BESIDE the horizontal pair (second stream), the upward family then adding (S_H + S_D) to its own sum: SGM_FAM_PAR=1.  Built,
        // exact, and measured slower than one after the other (alternated twice on one box, ms per pipeline step: 4096^2 x 257
        // 46.7 / 48.1 beside against 43.8 / 43.8 in line; 10000^2 x 129 133.9 / 128.9 against 125.4 / 125.5): the pair alone moves its
        // bytes at the rate the box gives a mixed stream, the marching pass nearly so - side by side they share it, and the pair's
        // wavefronts sit on the marching pass's SIMDs (profiles/r06_fam_par_ab.txt).  So the default stays in line; the hook stays a
        // tested route.
        const char* epar = pmx_opt(ctx, "SGM_FAM_PAR");
        bool par = hb && db && ub && epar && epar[0] == '1';
        float* SD = nullptr;
        bool sd_temp = false;
        if (par) {
            if (defer) {
                if (cv->spart2_bytes < bytes) {
                    pmx_pool_free(ctx, cv->spart2);
                    cv->spart2 = nullptr;
                    cv->spart2_bytes = 0;
                    if (pmx_pool_alloc(ctx, (void**)&cv->spart2, bytes) == hipSuccess) cv->spart2_bytes = bytes;
                    else { cv->spart2 = nullptr; (void)hipGetLastError(); }
                }
                SD = cv->spart2;
            } else if (pmx_pool_alloc(ctx, (void**)&SD, bytes) == hipSuccess) {
                sd_temp = true;
            } else {
                SD = nullptr;
                (void)hipGetLastError();
            }
            par = SD != nullptr;
        }
        if (db || ub) {
            rc = pmx_sgm_family_prepare(ctx, cv);
            if (rc) return rc;
        }
        {
            pmx_stage_scope span(ctx, PMX_STAGE_SGM_SPAN);  // the SGM step as the pipeline sees it (fork ... join)
            if (par) {
                if (!ctx->aux_stream) {
                    PMX_HIP(hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking));
                    PMX_HIP(hipEventCreateWithFlags(&ctx->aux_fork, hipEventDisableTiming));
                    PMX_HIP(hipEventCreateWithFlags(&ctx->aux_join, hipEventDisableTiming));
                }
                PMX_HIP(hipEventRecord(ctx->aux_fork, ctx->stream));
                PMX_HIP(hipStreamWaitEvent(ctx->aux_stream, ctx->aux_fork, 0));
                // the marching pass first: one workgroup per CU, the pair's wavefronts fill what is left of the SIMDs
                rc = pmx_launch_sgm_family(ctx, cv, 0, nullptr, nullptr, SD, P1, P2, is_max, invalid_cost, overcounting, db, false, nullptr,
                                           ctx->aux_stream);
                if (rc) return rc;
                PMX_HIP(hipEventRecord(ctx->aux_join, ctx->aux_stream));
            }
            // horizontal pair with the line kernel (checkpoint + recompute, this file)
#define PMX_CALL(K) sgm_run_horizontal<K>(ctx, a, mask)
            PMX_KPL_SWITCH(kpl, PMX_CALL)
#undef PMX_CALL
            if (rc) return rc;
            if (par) {
                PMX_HIP(hipStreamWaitEvent(ctx->stream, ctx->aux_join, 0));
            } else if (db) {
                rc = pmx_launch_sgm_family(ctx, cv, 0, hb ? a.S : nullptr, nullptr, a.S, P1, P2, is_max, invalid_cost, overcounting, db,
                                           ub == 0, nullptr, nullptr);
                if (rc) return rc;
            }
        }
        if (defer) {
            cv->pending = {P1, P2, invalid_cost, is_max, overcounting, par ? 1 : 0};
            cv->repr = PMX_REPR_SGM_UP_PENDING;
            return PMX_OK;  // data = the costs, spart (+ spart2) = the partial sums
        }
        if (ub) {
            rc = pmx_launch_sgm_family(ctx, cv, 1, (hb || db) ? a.S : nullptr, par ? SD : nullptr, a.S, P1, P2, is_max, invalid_cost,
                                       overcounting, ub, true, nullptr, nullptr);
        }
        if (sd_temp) pmx_pool_free(ctx, SD);  // stream-ordered reuse
        if (rc) return rc;
    } else {
        float* T = nullptr;  // the second accumulator of the family-wise sum (a later family with more than one path)
        if (sgm_seq_needs_t(mask)) PMX_HIP(pmx_pool_alloc(ctx, (void**)&T, bytes));
        a.T = T;
#define PMX_CALL(K) sgm_run<K>(ctx, a, mask)
        PMX_KPL_SWITCH(kpl, PMX_CALL)
#undef PMX_CALL
        pmx_pool_free(ctx, T);  // stream-ordered reuse
    }
    if (rc) return rc;
    // the accumulator becomes the volume; the old volume becomes the scratch
    float* old = cv->data;
    size_t old_bytes = cv->bytes;
    cv->data = ctx->scratch;
    cv->bytes = ctx->scratch_bytes;
    ctx->scratch = old;
    ctx->scratch_bytes = old_bytes;
    return PMX_OK;
}

// The upward family of a PMX_REPR_SGM_UP_PENDING handle: in WTA mode (wta != nullptr: the volume stays pending, the context's
// disparity map and winner cache are written) or in store mode (the handle becomes a plain float32 volume).
int pmx_sgm_finish_pending(pmx_ctx* ctx, pmx_cv* cv, const pmx_fam_wta* wta) {
    PMX_CHECK(cv->repr == PMX_REPR_SGM_UP_PENDING && cv->data && cv->spart && (!cv->pending.two || cv->spart2), PMX_ERR_STATE,
              "pmx_sgm_finish_pending: nothing pending");
    int rc = pmx_sgm_family_prepare(ctx, cv);
    if (rc) return rc;
    rc = pmx_launch_sgm_family(ctx, cv, 1, cv->spart, cv->pending.two ? cv->spart2 : nullptr, cv->spart, cv->pending.P1, cv->pending.P2,
                               cv->pending.is_max, cv->pending.invalid_cost, cv->pending.overcounting, 7, true, wta, nullptr);
    if (rc || wta) return rc;
    // the sums become the volume; the costs' buffer is kept with the handle for the next pair's partial sums
    float* costs = cv->data;
    const size_t costs_bytes = cv->bytes;
    cv->data = cv->spart;
    cv->bytes = cv->spart_bytes;
    cv->spart = costs;
    cv->spart_bytes = costs_bytes;
    cv->repr = PMX_REPR_FLOAT;
    return PMX_OK;
}
