// pmx_buf.h - a lane's run of consecutive float32 values through raw buffer instructions.  gfx950.
//
// The cost volume is [H][W][D] with the exact stride D (the reference's layout), so a lane's KPL consecutive disparities start
// at a 4-byte aligned address and a pixel's last lane may own fewer than KPL.  Buffer instructions need 4-byte alignment only
// for their 8- and 16-byte forms and check every access against the descriptor's range: reads past it return 0, writes past it
// are dropped.  The SGM kernels use that instead of branches - a memory instruction under a lane-varying branch makes the
// compiler's vmcnt bookkeeping wait for the youngest stores, and dword-by-dword tails cost one texture-addresser slot each.
#pragma once

#include <hip/hip_runtime.h>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr unsigned kRsrcWord3 = 0x00020000;  // raw buffer: out-of-range loads return 0, out-of-range stores are dropped
constexpr unsigned kOob = 0x80000000u;       // a byte offset beyond every row: lanes that must not touch memory use it

// The compiler's wait-count pass merges a loop's entry state with its back-edge state, and at the entry a prefetch ring's loads
// have just been issued: it then counts every load of the steady state as if only the prologue's few operations had followed it.
// An explicit "everything has arrived" in front of the loop (one memory latency, once per wavefront) makes the entry state empty,
// and the counts the pass derives are the steady state's.  The same pass counts NONE of the memory operations of a conditional
// block (uniform or not) on the paths that merge behind it: inside a ring's loop every load and store is issued unconditionally
// (masked lanes carry kOob), and an abort returns from inside the loop.  (s_waitcnt vmcnt(0), expcnt / lgkmcnt untouched)
#define PMX_LOOP_ENTRY_DRAIN() __builtin_amdgcn_s_waitcnt(0x0F70)

// ---- window tickets of the marching kernels (k_sgmfam.hip, k_sgmfam8.hip) --------------------------------------------------------------
// A marching launch is a one-directional pipeline of column windows: window w reads, row by row, what window w - 1 publishes.
// WHICH workgroup runs which window is decided when the workgroup starts, and two things are wanted of that decision:
//   (1) neighbours on one XCD: a hand-off that stays in an XCD's L2 costs a fraction of one over the fabric (round 6: chunks of G
//       consecutive windows belong to XCD (chunk mod 8); a workgroup prefers the chunks of the XCD it finds itself on);
//   (2) NO workgroup ever holds a window whose left neighbour has not been taken by a workgroup that is running (or done).  Then
//       every wait inside the launch is a wait for a resident workgroup, the leftmost unfinished window waits for nobody, and the
//       launch makes progress on whatever share of the device it gets - beside other kernels, other contexts, other PROCESSES.
// Round 6's first form (one ticket counter per XCD, taken blindly) had (1) without (2): a workgroup could sit on a CU with the
// first window of a chunk whose predecessor chunk - another XCD's - nobody had started, and eight processes sharing one GPU held
// each other's XCDs that way until the spin limits expired ("gave up waiting for a neighbouring window"; found by the 8-rank
// bench test).  Now: chunk c (of family f) has a counter cnt[f][c] of windows taken; a window of chunk c may only be taken once
// chunk c - 1 is FULL (cnt >= its size; readiness only ever turns true, so check-then-add needs no compare-and-swap; counters may
// overshoot, an overshooting ticket is no window).  Inside a chunk windows are taken in order.  So the taken windows of a family
// are always a prefix: (2).  A workgroup polls its own XCD's chunks for `patience` rounds - at launch the chunks fill one after
// the other, a ripple of a few microseconds per XCD - and then takes the leftmost window anybody may take, wherever it belongs
// (always ready, by the prefix property): (1) is a preference, (2) a guarantee.
// Called by ONE thread.  Returns family * nwin + window, or -1 when every window is taken.
struct pmx_win_tickets {
    unsigned* cnt;   // [nfam][nchunk] windows taken per chunk, zeroed per launch
    int G, nwin, nchunk, nfam;
};
__device__ __forceinline__ int pmx_take_window(const pmx_win_tickets& k, unsigned xcc, unsigned first_fam) {
    auto cap = [&](int c) { const int left = k.nwin - c * k.G; return (unsigned)(left < k.G ? left : k.G); };
    auto taken = [&](int f, int c) { return __hip_atomic_load(k.cnt + f * k.nchunk + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto try_chunk = [&](int f, int c, bool* not_ready) -> int {
        if (taken(f, c) >= cap(c)) return -1;                                        // full
        if (c > 0 && taken(f, c - 1) < cap(c - 1)) { *not_ready = true; return -1; } // its left neighbour chunk still has windows
        const unsigned t = atomicAdd(k.cnt + f * k.nchunk + c, 1u);
        return t < cap(c) ? f * k.nwin + c * k.G + (int)t : -1;                      // (filled meanwhile: no window)
    };
#ifndef PMX_TICKET_PATIENCE
#define PMX_TICKET_PATIENCE 128
#endif
    constexpr int kPatience = PMX_TICKET_PATIENCE;  // rounds of polling the own chunks (several microseconds each) before any window will do
    for (int round = 0; round < kPatience; ++round) {
        bool waiting = false;
        for (int c = (int)xcc; c < k.nchunk; c += 8)
            for (int ff = 0; ff < k.nfam; ++ff) {
                const int w = try_chunk((int)((first_fam + (unsigned)ff) % (unsigned)k.nfam), c, &waiting);
                if (w >= 0) return w;
            }
        if (!waiting) break;  // every chunk of this XCD is full
        __builtin_amdgcn_s_sleep(8);
    }
    for (;;) {  // the leftmost window that is free, of either family: its left neighbours are all taken
        bool any = false;
        for (int c = 0; c < k.nchunk; ++c)
            for (int ff = 0; ff < k.nfam; ++ff) {
                bool nr = false;
                const int w = try_chunk((int)((first_fam + (unsigned)ff) % (unsigned)k.nfam), c, &nr);
                if (w >= 0) return w;
                any |= nr;
            }
        if (!any) return -1;  // all full
        // (a chunk that is not ready behind a chunk that has just been filled by somebody else's add: look again)
    }
}

// A lane's KPL consecutive floats move as 16-byte pieces, then an 8-byte one, then a 4-byte one (4-byte alignment is enough
// for buffer instructions).  piece i = [piece_start(i), piece_start(i) + piece_width(i)).
template <int KPL>
struct pieces {
    static constexpr int N4 = KPL / 4, R = KPL % 4, N = N4 + (R ? 1 : 0);
    static constexpr int start(int i) { return 4 * i; }
    static constexpr int width(int i) { return i < N4 ? 4 : R; }  // the remainder moves as ONE 12-, 8- or 4-byte piece
};

typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));

template <int KPL>
__device__ __forceinline__ void buf_load(__amdgpu_buffer_rsrc_t rs, unsigned voff, float (&dst)[KPL]) {
    using P = pieces<KPL>;
#pragma unroll
    for (int i = 0; i < P::N; ++i) {
        const int k = P::start(i);
        if (P::width(i) == 4) {
            const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 4 * k, 0, 0);
            dst[k] = __uint_as_float(t.x); dst[k + 1] = __uint_as_float(t.y);
            dst[k + 2] = __uint_as_float(t.z); dst[k + 3] = __uint_as_float(t.w);
        } else if (P::width(i) == 3) {
            const u32x3 t = __builtin_amdgcn_raw_buffer_load_b96(rs, voff + 4 * k, 0, 0);
            dst[k] = __uint_as_float(t.x); dst[k + 1] = __uint_as_float(t.y); dst[k + 2] = __uint_as_float(t.z);
        } else if (P::width(i) == 2) {
            const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, voff + 4 * k, 0, 0);
            dst[k] = __uint_as_float(t.x); dst[k + 1] = __uint_as_float(t.y);
        } else {
            dst[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff + 4 * k, 0, 0));
        }
    }
}

// Stores v[0 .. nv) of every lane: nv = KPL on most lanes, tailn on the pixel's last lane when D is not a multiple of KPL, 0 on
// lanes without disparities.  No branch on lane-varying data: a lane takes part in a piece when the piece lies inside its
// nv, otherwise its offset is kOob and the buffer bounds check drops the write.  What the pieces leave of the tail lane (fewer
// than 4 values, starting at `cov`; cov and rem are the same for every pixel) goes out as single dwords.
template <int KPL>
__device__ __forceinline__ void buf_store(__amdgpu_buffer_rsrc_t rs, unsigned base_off, int nv, bool is_tail, int cov, int rem,
                                          const float (&v)[KPL]) {
    using P = pieces<KPL>;
#pragma unroll
    for (int i = 0; i < P::N; ++i) {
        const int k = P::start(i);
        const unsigned off = (nv >= k + P::width(i)) ? base_off + 4 * k : kOob;
        if (P::width(i) == 4) {
            u32x4 t;
            t.x = __float_as_uint(v[k]); t.y = __float_as_uint(v[k + 1]); t.z = __float_as_uint(v[k + 2]); t.w = __float_as_uint(v[k + 3]);
            __builtin_amdgcn_raw_buffer_store_b128(t, rs, off, 0, 0);
        } else if (P::width(i) == 3) {
            u32x3 t;
            t.x = __float_as_uint(v[k]); t.y = __float_as_uint(v[k + 1]); t.z = __float_as_uint(v[k + 2]);
            __builtin_amdgcn_raw_buffer_store_b96(t, rs, off, 0, 0);
        } else if (P::width(i) == 2) {
            u32x2 t;
            t.x = __float_as_uint(v[k]); t.y = __float_as_uint(v[k + 1]);
            __builtin_amdgcn_raw_buffer_store_b64(t, rs, off, 0, 0);
        } else {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[k]), rs, off, 0, 0);
        }
    }
    // the tail's leftover (rem < 4 values from `cov`, both uniform): the lanes that are not the tail are masked by their offset -
    // no LANE-VARYING branch around a memory instruction (that would make the compiler's vmcnt bookkeeping wait for the youngest
    // stores)
    float t0 = v[0], t1 = v[0], t2 = v[0];
#pragma unroll
    for (int i = 0; i < P::N; ++i) {
        const int k = P::start(i);
        if (cov == k) {
            t0 = v[k];
            t1 = v[k + 1 < KPL ? k + 1 : k];
            t2 = v[k + 2 < KPL ? k + 2 : k];
        }
    }
    // `rem` is the same for every lane and every pixel (D and the lane map): a uniform branch picks the ONE instruction the
    // leftover needs (12, 8 or 4 bytes) - two always-issued, mostly masked stores cost two texture-addresser slots per step
    const unsigned toff = is_tail ? base_off + 4u * (unsigned)cov : kOob;
    if (rem == 3) {
        u32x3 t;
        t.x = __float_as_uint(t0); t.y = __float_as_uint(t1); t.z = __float_as_uint(t2);
        __builtin_amdgcn_raw_buffer_store_b96(t, rs, toff, 0, 0);
    } else if (rem == 2) {
        u32x2 t01;
        t01.x = __float_as_uint(t0); t01.y = __float_as_uint(t1);
        __builtin_amdgcn_raw_buffer_store_b64(t01, rs, toff, 0, 0);
    } else if (rem == 1) {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(t0), rs, toff, 0, 0);
    }
}

// ---- a lane's run of N = 1 .. 5 dwords (the integer path's packed bytes) ----------------------------------------------------------
template <int N>
__device__ __forceinline__ void load_dwords(__amdgpu_buffer_rsrc_t rs, unsigned off, uint32_t (&x)[N], unsigned soff = 0) {
    static_assert(N >= 1 && N <= 5, "cost dwords per lane");
    if constexpr (N == 1) {
        x[0] = __builtin_amdgcn_raw_buffer_load_b32(rs, off, soff, 0);
    } else if constexpr (N == 2) {
        const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, off, soff, 0);
        x[0] = t.x; x[1] = t.y;
    } else if constexpr (N == 3) {
        const u32x3 t = __builtin_amdgcn_raw_buffer_load_b96(rs, off, soff, 0);
        x[0] = t.x; x[1] = t.y; x[2] = t.z;
    } else {
        const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, off, soff, 0);
        x[0] = t.x; x[1] = t.y; x[2] = t.z; x[3] = t.w;
        if constexpr (N == 5) x[4] = __builtin_amdgcn_raw_buffer_load_b32(rs, off + 16, soff, 0);
    }
}
template <int N>
__device__ __forceinline__ void store_dwords(__amdgpu_buffer_rsrc_t rs, unsigned off, const uint32_t (&x)[N], unsigned soff = 0) {
    static_assert(N >= 1 && N <= 5, "bytes per lane / 4");
    if constexpr (N == 1) {
        __builtin_amdgcn_raw_buffer_store_b32(x[0], rs, off, soff, 0);
    } else if constexpr (N == 2) {
        u32x2 t; t.x = x[0]; t.y = x[1];
        __builtin_amdgcn_raw_buffer_store_b64(t, rs, off, soff, 0);
    } else if constexpr (N == 3) {
        u32x3 t; t.x = x[0]; t.y = x[1]; t.z = x[2];
        __builtin_amdgcn_raw_buffer_store_b96(t, rs, off, soff, 0);
    } else {
        u32x4 t; t.x = x[0]; t.y = x[1]; t.z = x[2]; t.w = x[3];
        __builtin_amdgcn_raw_buffer_store_b128(t, rs, off, soff, 0);
        if constexpr (N == 5) __builtin_amdgcn_raw_buffer_store_b32(x[4], rs, off == kOob ? kOob : off + 16, soff, 0);
    }
}

// ---- a prefetch ring the compiler cannot see: registers above its own ----------------------------------------------------------------
// VALIDATED ON: ROCm 7.2.0 (hipcc / HIP 7.2.26015-fc0010cf6a, AMD clang 22.0.0git roc-7.2.0), gfx950.  A toolchain change is a re-validation: `make` runs
// tools/check_hring.py on the object (nothing but the ring's statements touches its registers; every take's vmcnt equals the memory
// instructions issued since its slot's load) and FAILS the build otherwise - it does not repair it; then tools/flaky_fam8.py and
// tests/test_gpu_fam8.py on the GPU.
// A register ring of loads issued PF steps ahead wants `s_waitcnt vmcnt(N)` with N = the memory instructions issued since the load
// that is due.  The compiler derives N itself, and where a step also STORES it gets it wrong in ways the source cannot steer: the
// marching kernels' three-step body was waited for with vmcnt(3), vmcnt(4), vmcnt(0) where 6 is right - one row of look-ahead
// instead of three, and in every third row a wait for the stores issued a few hundred cycles before (cycle stamps: that row's
// compute phase 1900-3200 cycles instead of 1085); with the loop reshaped so that its counts came out right it copied
// the ring's registers at the loop header, i.e. read them - and waited - one row after the load.  Loads hidden in inline asm with
// ordinary outputs are no way out either: the compiler believes such a value is there when the asm statement ends and may copy it
// before the hand-written wait (round 4's "rare wrong block").  So the ring lives in registers the compiler does not allocate in
// these kernels: above their own (the allocator fills from v0).  Accumulation registers would be the natural home, but
// a kernel that names any gets its register budget split in half between the two files, and then spills into the AGPRs the ring
// does not name.  hring_load<BASE, N> issues the load of N dwords per lane (offset `off` of the descriptor `rs`) into
// v[BASE .. BASE + N - 1]; hring_take<BASE, N, CNT> waits until at most CNT younger memory instructions are in flight (vmcnt counts
// loads and stores alike, in order) and copies them into compiler-visible registers.  The register numbers are immediate operands of
// the asm statements ("n"), so BASE may come from template arithmetic; what an asm statement cannot take from a template is its
// clobber list, so the kernel names the ring's LAST register once (PMX_HRING_RESERVE("v111")): that is what makes the kernel's
// descriptor cover the ring.  Rules for a loop that uses it: every memory instruction between a
// slot's load and its take is unconditional, and CNT is their number.  `make` checks each object that nothing but these two statements touches
// the ring's registers (tools/check_hring.py <object> <first register>) - if the compiler ever needs that many, the build fails
// instead of the results.
typedef unsigned int pmx_rsrc_words __attribute__((ext_vector_type(4)));
__device__ __forceinline__ pmx_rsrc_words rsrc_words(const void* base, unsigned bytes) {
    pmx_rsrc_words w;
    w.x = (unsigned)(uintptr_t)base;
    w.y = (unsigned)((uintptr_t)base >> 32) & 0xffffu;
    w.z = bytes;
    w.w = kRsrcWord3;
    return w;
}
#define PMX_HRING_RESERVE(LAST_REG) asm volatile("; hring: registers up to " LAST_REG ::: LAST_REG)
// v[BASE .. BASE + N - 1] <- N dwords per lane, as 16-byte pieces and one remainder piece (BASE even: tuples are 2-aligned).
// (An immediate operand above 64 prints in hexadecimal, which is no register name: a number goes in as "tens" and "ones".)
#define PMX_RN(R) "n"((R) / 10), "n"((R) % 10)
template <int BASE, int N>
__device__ __forceinline__ void hring_load(pmx_rsrc_words rs, unsigned off) {
    static_assert(BASE % 2 == 0 && BASE >= 32 && BASE + N <= 256 && N >= 1 && N <= 12 && 16 * (N / 4) <= 64, "hidden ring registers");
#pragma unroll
    for (int i = 0; i < N / 4; ++i)
        asm volatile("buffer_load_dwordx4 v[%2%3:%4%5], %0, %1, 0 offen offset:%6" ::"v"(off), "s"(rs), PMX_RN(BASE + 4 * i),
                     PMX_RN(BASE + 4 * i + 3), "n"(16 * i)
                     : "memory");
    constexpr int R = N % 4, B = BASE + 4 * (N / 4), O = 16 * (N / 4);
    if constexpr (R == 3)
        asm volatile("buffer_load_dwordx3 v[%2%3:%4%5], %0, %1, 0 offen offset:%6" ::"v"(off), "s"(rs), PMX_RN(B), PMX_RN(B + 2), "n"(O) : "memory");
    else if constexpr (R == 2)
        asm volatile("buffer_load_dwordx2 v[%2%3:%4%5], %0, %1, 0 offen offset:%6" ::"v"(off), "s"(rs), PMX_RN(B), PMX_RN(B + 1), "n"(O) : "memory");
    else if constexpr (R == 1)
        asm volatile("buffer_load_dword v%2%3, %0, %1, 0 offen offset:%4" ::"v"(off), "s"(rs), PMX_RN(B), "n"(O) : "memory");
}
// memory instructions hring_load<., N> issues
constexpr int hring_loads(int n) { return n / 4 + (n % 4 ? 1 : 0); }
template <int BASE, int N, int CNT, typename T>
__device__ __forceinline__ void hring_take(T (&x)[N]) {
    static_assert(sizeof(T) == 4 && CNT >= 0 && CNT < 64, "dwords, vmcnt");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT) : "memory");
#pragma unroll
    for (int k = 0; k < N; ++k) asm volatile("v_mov_b32 %0, v%1%2" : "=v"(x[k]) : PMX_RN(BASE + k) : "memory");
}
