// vmarch.hip - what does a kernel that marches DOWN the rows of a [H][W][D] float volume get out of the memory system at
// BASELINE's largest size (10000 x 10000 x 129: rows of 5.16 MB, 51.6 GB per volume)?  CBCA's pass V reads E_h and writes the
// aggregated volume that way: thread = (column, disparity), one serial fp32 prefix per thread.  Here only the pattern: out = running
// sum of in, R rows of read-ahead, every workgroup covering BS * CPT * 4 contiguous bytes of each row.
//   variants: BS threads, CPT cells per thread (cells of one thread are BS floats apart: every instruction of a wavefront is
//   256 contiguous bytes), V = floats per lane and instruction (1: dword, 2: dwordx2, 4: dwordx4), workgroup order by atomic
//   ticket or by blockIdx, an optional throttle (a workgroup never runs more than K rows ahead of its left neighbour).
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/vmarch tools/ubench/vmarch.hip ; run: /tmp/vmarch [H W D]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));       \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

struct args {
    const float* in;
    float* out;
    int H;
    size_t row;      // floats per row
    unsigned* ctl;   // [0] ticket, [16 + w] rows done by workgroup w
    int throttle;    // 0: none; K > 0: at most K rows ahead of the left neighbour
    int ticket;      // workgroup index from an atomic ticket
    int mode;        // 0 read + write, 1 read only, 2 write only
};

template <int V>
struct vec;
template <>
struct vec<1> { typedef float t; };
template <>
struct vec<2> { typedef float2 t; };
template <>
struct vec<4> { typedef float4 t; };

template <int BS, int CPT, int V, int R>
__global__ __launch_bounds__(BS) void march(args a) {
    typedef typename vec<V>::t vt;
    __shared__ unsigned wg_s;
    unsigned wg = blockIdx.x;
    if (a.ticket) {
        if (threadIdx.x == 0) wg_s = atomicAdd(a.ctl, 1u);
        __syncthreads();
        wg = wg_s;
    }
    const size_t span = (size_t)BS * CPT * V;  // floats of a row this workgroup owns
    const size_t base = (size_t)wg * span + (size_t)threadIdx.x * V;
    bool live[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) live[c] = base + (size_t)c * BS * V + V <= a.row;
    vt cur[R][CPT];
    float acc[CPT][V];
#pragma unroll
    for (int c = 0; c < CPT; ++c)
#pragma unroll
        for (int v = 0; v < V; ++v) acc[c][v] = 0.f;
    auto ld = [&](int r, int c) {
        vt x{};
        if (a.mode != 2 && live[c] && r < a.H) x = *reinterpret_cast<const vt*>(a.in + (size_t)r * a.row + base + (size_t)c * BS * V);
        return x;
    };
#pragma unroll
    for (int j = 0; j < R; ++j)
#pragma unroll
        for (int c = 0; c < CPT; ++c) cur[j][c] = ld(j, c);
    volatile unsigned* prog = a.ctl + 16;
    for (int r = 0; r < a.H; r += R) {
        if (a.throttle && wg > 0) {
            if (threadIdx.x == 0) {
                int spins = 0;
                while ((int)__hip_atomic_load(&a.ctl[16 + wg - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + a.throttle < r + R && ++spins < (1 << 22))
                    __builtin_amdgcn_s_sleep(2);
            }
            __syncthreads();
        }
        vt nxt[R][CPT];
#pragma unroll
        for (int j = 0; j < R; ++j)
#pragma unroll
            for (int c = 0; c < CPT; ++c) nxt[j][c] = ld(r + R + j, c);
#pragma unroll
        for (int j = 0; j < R; ++j) {
            if (r + j < a.H) {
#pragma unroll
                for (int c = 0; c < CPT; ++c) {
                    vt o;
                    float* of = reinterpret_cast<float*>(&o);
                    const float* cf = reinterpret_cast<const float*>(&cur[j][c]);
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        acc[c][v] += cf[v];
                        of[v] = acc[c][v];
                    }
                    if (a.mode != 1 && live[c]) *reinterpret_cast<vt*>(a.out + (size_t)(r + j) * a.row + base + (size_t)c * BS * V) = o;
                }
            }
        }
        if (a.mode == 1) {  // keep the loads alive
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < CPT; ++c) s += acc[c][0];
            if (s == 12345.678f) a.out[base] = s;
        }
#pragma unroll
        for (int j = 0; j < R; ++j)
#pragma unroll
            for (int c = 0; c < CPT; ++c) cur[j][c] = nxt[j][c];
        if (a.throttle && threadIdx.x == 0) __hip_atomic_store(&a.ctl[16 + wg], (unsigned)(r + R), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (a.throttle && threadIdx.x == 0) __hip_atomic_store(&a.ctl[16 + wg], 0x3fffffffu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    (void)prog;
}

// The horizontal passes' pattern: one wavefront per image ROW streams its row (5.16 MB) in pieces of `CH` KB, `AH` pieces ahead -
// thousands of sequential streams, every one in its own pages.  mode as above.
template <int CH, int AH>
__global__ __launch_bounds__(256) void rowstream(args a) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.H) return;
    const float4* in = reinterpret_cast<const float4*>(a.in + (size_t)row * a.row);
    float4* out = reinterpret_cast<float4*>(a.out + (size_t)row * a.row);
    constexpr int V = CH * 1024 / 16 / 64;  // float4 per lane and piece
    const size_t npiece = a.row * 4 / (CH * 1024);
    float4 buf[AH][V];
    float acc = 0.f;
    auto ld = [&](size_t p, float4 (&dst)[V]) {
#pragma unroll
        for (int v = 0; v < V; ++v) dst[v] = (a.mode != 2 && p < npiece) ? in[p * (CH * 64) + v * 64 + lane] : float4{0, 0, 0, 0};
    };
#pragma unroll
    for (int i = 0; i < AH; ++i) ld(i, buf[i]);
    for (size_t p = 0; p < npiece; p += AH) {
#pragma unroll
        for (int i = 0; i < AH; ++i) {
            float4 cur[V];
#pragma unroll
            for (int v = 0; v < V; ++v) cur[v] = buf[i][v];
            ld(p + AH + i, buf[i]);
            if (p + i < npiece) {
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    acc += cur[v].x;
                    cur[v].x = acc;
                    if (a.mode != 1) out[(p + i) * (CH * 64) + v * 64 + lane] = cur[v];
                }
            }
        }
    }
    if (a.mode == 1 && acc == 12345.678f) a.out[row] = acc;
}

int main(int argc, char** argv) {
    const int H = argc > 1 ? atoi(argv[1]) : 10000, W = argc > 2 ? atoi(argv[2]) : 10000, D = argc > 3 ? atoi(argv[3]) : 129;
    const size_t row = (size_t)W * D, n = (size_t)H * row;
    float *in, *out;
    unsigned* ctl;
    CK(hipMalloc(&in, n * 4 + 4096));
    CK(hipMalloc(&out, n * 4 + 4096));
    CK(hipMalloc(&ctl, 4 << 20));
    CK(hipMemset(in, 0, n * 4));
    CK(hipMemset(out, 0, n * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("volume %d x %d x %d: rows of %.2f MB, %.1f GB\n", H, W, D, row * 4 / 1e6, n * 4 / 1e9);
    auto timeit = [&](const char* name, int mode, int ticket, int throttle, auto launch) {
        args a{in, out, H, row, ctl, throttle, ticket, mode};
        float best = 1e30f;
        for (int i = 0; i < 2; ++i) {
            CK(hipMemsetAsync(ctl, 0, 4 << 20, 0));
            CK(hipEventRecord(e0));
            launch(a);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipGetLastError());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double bytes = (mode == 0 ? 2.0 : 1.0) * n * 4;
        printf("%-64s %s%s%-3d %8.2f ms  %.2f TB/s\n", name, mode == 0 ? "rw " : mode == 1 ? "r  " : "w  ", ticket ? "ticket " : "       ", throttle, best,
               bytes / best / 1e9);
        fflush(stdout);
    };
#define RUN(BS, CPT, V, R, mode, ticket, throttle)                                                                                      \
    timeit("BS " #BS " x " #CPT " cells x " #V " floats, " #R " rows ahead", mode, ticket, throttle, [&](const args& a) {             \
        const size_t span = (size_t)BS * CPT * V;                                                                                       \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(march<BS, CPT, V, R>), dim3((unsigned)((row + span - 1) / span)), dim3(BS), 0, 0, a);        \
    })
    RUN(512, 1, 1, 4, 0, 0, 0);   // pass V as it is: 2 KB per workgroup and row
    RUN(512, 1, 1, 4, 1, 0, 0);
    RUN(512, 1, 1, 4, 2, 0, 0);
    RUN(256, 1, 1, 4, 0, 0, 0);
    RUN(1024, 1, 1, 4, 0, 0, 0);
    RUN(512, 1, 1, 8, 0, 0, 0);
    RUN(512, 2, 1, 4, 0, 0, 0);   // 4 KB
    RUN(512, 4, 1, 4, 0, 0, 0);   // 8 KB
    RUN(256, 4, 1, 4, 0, 0, 0);   // 4 KB
    RUN(256, 8, 1, 4, 0, 0, 0);   // 8 KB
    RUN(256, 1, 2, 4, 0, 0, 0);   // 8 bytes per lane (row length must divide: the last partial vector is dropped)
    RUN(256, 1, 4, 4, 0, 0, 0);   // 16 bytes per lane, 4 KB
    RUN(256, 2, 4, 4, 0, 0, 0);   // 16 bytes per lane, 8 KB
    RUN(256, 4, 4, 4, 0, 0, 0);   // 16 bytes per lane, 16 KB
    RUN(512, 1, 1, 4, 0, 1, 0);   // ticket order
    RUN(512, 1, 1, 4, 0, 1, 4);   // ticket + throttle
    RUN(512, 1, 1, 4, 0, 1, 16);
    RUN(512, 1, 1, 4, 0, 1, 64);
    RUN(256, 4, 1, 4, 0, 1, 8);
    RUN(256, 1, 4, 4, 0, 1, 8);
#define RUNROW(CH, AH, mode)                                                                                                         \
    timeit("one wavefront per row, pieces of " #CH " KB, " #AH " ahead", mode, 0, 0, [&](const args& a) {                            \
        hipLaunchKernelGGL(HIP_KERNEL_NAME(rowstream<CH, AH>), dim3((unsigned)((H + 3) / 4)), dim3(256), 0, 0, a);                   \
    })
    RUNROW(1, 8, 1);
    RUNROW(4, 2, 1);
    RUNROW(4, 4, 1);
    RUNROW(1, 8, 0);
    RUNROW(4, 2, 0);
    RUNROW(4, 4, 0);
    return 0;
}
