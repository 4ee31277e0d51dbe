// cbca_pattern.hip - what bounds the CBCA scans?  The memory access pattern of pass H (thread = (row, disparity), marching along
// the columns of a [H][W][D] float32 volume with D = 129: every wavefront moves 256 unaligned bytes per instruction, 516 bytes
// further every step), stripped of the scan arithmetic, against variants that move the same bytes differently.
//   A  the pattern as the kernels have it: 4 B per lane, G columns in flight
//   B  two disparities per lane (8 B)
//   C  a block owns R whole rows: G columns of a row are G*D*4 contiguous bytes, moved with 16 B per lane through LDS
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/cbca_pattern tools/ubench/cbca_pattern.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e = (x);                                                    \
        if (e != hipSuccess) {                                                 \
            printf("%s: %s\n", #x, hipGetErrorString(e));                      \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

template <int G>
__global__ __launch_bounds__(256) void pat_a(const float* __restrict__ in, float* __restrict__ out, int H, int W, int D) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= H * D) return;
    const int r = t / D, k = t - r * D;
    const float* p = in + (size_t)r * W * D + k;
    float* q = out + (size_t)r * W * D + k;
    float cur[G], nxt[G];
#pragma unroll
    for (int j = 0; j < G; ++j) cur[j] = p[(size_t)j * D];
    float acc = 0.f;
    for (int c = 0; c < W; c += G) {
        const int cn = c + G < W ? c + G : c;
#pragma unroll
        for (int j = 0; j < G; ++j) nxt[j] = p[(size_t)(cn + j) * D];
#pragma unroll
        for (int j = 0; j < G; ++j) {
            acc += cur[j];
            q[(size_t)(c + j) * D] = acc;
        }
#pragma unroll
        for (int j = 0; j < G; ++j) cur[j] = nxt[j];
    }
}

template <int G>
__global__ __launch_bounds__(256) void pat_b(const float* __restrict__ in, float* __restrict__ out, int H, int W, int D) {
    const int D2 = (D + 1) / 2;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= H * D2) return;
    const int r = t / D2, k = (t - r * D2) * 2;
    const bool two = k + 1 < D;
    const float* p = in + (size_t)r * W * D + k;
    float* q = out + (size_t)r * W * D + k;
    float cur[G][2], nxt[G][2];
    auto ld = [&](float (&d)[2], size_t off) {
        d[0] = p[off];
        d[1] = two ? p[off + 1] : 0.f;
    };
#pragma unroll
    for (int j = 0; j < G; ++j) ld(cur[j], (size_t)j * D);
    float a0 = 0.f, a1 = 0.f;
    for (int c = 0; c < W; c += G) {
        const int cn = c + G < W ? c + G : c;
#pragma unroll
        for (int j = 0; j < G; ++j) ld(nxt[j], (size_t)(cn + j) * D);
#pragma unroll
        for (int j = 0; j < G; ++j) {
            a0 += cur[j][0];
            a1 += cur[j][1];
            q[(size_t)(c + j) * D] = a0;
            if (two) q[(size_t)(c + j) * D + 1] = a1;
        }
#pragma unroll
        for (int j = 0; j < G; ++j) { cur[j][0] = nxt[j][0]; cur[j][1] = nxt[j][1]; }
    }
}

// C: block = R rows x D threads (rounded up to whole waves); chunk = G columns.  LDS: in[2][R][G*D], out[R][G*D].
template <int R, int G>
__global__ void pat_c(const float* __restrict__ in, float* __restrict__ out, int H, int W, int D) {
    extern __shared__ float lds[];
    const int GD = G * D;  // floats per row chunk (multiple of 4 when G is)
    float* sin = lds;                  // [2][R][GD]
    float* sout = lds + 2 * R * GD;    // [R][GD]
    const int nthreads = blockDim.x;
    const int r0 = blockIdx.x * R;
    const int tid = threadIdx.x;
    const int lr = tid / D, k = tid - lr * D;
    const bool owner = lr < R && r0 + lr < H;
    const int vec_per_row = GD / 4;
    auto stage_in = [&](int buf, int c0) {
        for (int i = tid; i < R * vec_per_row; i += nthreads) {
            const int rr = i / vec_per_row, v = i - rr * vec_per_row;
            if (r0 + rr < H) {
                const float4 x = *reinterpret_cast<const float4*>(in + ((size_t)(r0 + rr) * W + c0) * D + 4 * v);
                *reinterpret_cast<float4*>(sin + ((size_t)buf * R + rr) * GD + 4 * v) = x;
            }
        }
    };
    float acc = 0.f;
    stage_in(0, 0);
    __syncthreads();
    int buf = 0;
    for (int c0 = 0; c0 < W; c0 += G) {
        if (c0 + G < W) stage_in(buf ^ 1, c0 + G);
        if (owner) {
#pragma unroll
            for (int j = 0; j < G; ++j) {
                acc += sin[((size_t)buf * R + lr) * GD + j * D + k];
                sout[(size_t)lr * GD + j * D + k] = acc;
            }
        }
        __syncthreads();
        for (int i = tid; i < R * vec_per_row; i += nthreads) {
            const int rr = i / vec_per_row, v = i - rr * vec_per_row;
            if (r0 + rr < H)
                *reinterpret_cast<float4*>(out + ((size_t)(r0 + rr) * W + c0) * D + 4 * v) = *reinterpret_cast<const float4*>(sout + (size_t)rr * GD + 4 * v);
        }
        __syncthreads();
        buf ^= 1;
    }
}

// pass V's pattern: thread = (column, disparity) marching down the rows
template <int G>
__global__ __launch_bounds__(256) void pat_va(const float* __restrict__ in, float* __restrict__ out, int H, int W, int D) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= W * D) return;
    const size_t rs = (size_t)W * D;
    const float* p = in + t;
    float* q = out + t;
    float cur[G], nxt[G];
#pragma unroll
    for (int j = 0; j < G; ++j) cur[j] = p[(size_t)j * rs];
    float acc = 0.f;
    for (int r = 0; r < H; r += G) {
        const int rn = r + G < H ? r + G : r;
#pragma unroll
        for (int j = 0; j < G; ++j) nxt[j] = p[(size_t)(rn + j) * rs];
#pragma unroll
        for (int j = 0; j < G; ++j) {
            acc += cur[j];
            q[(size_t)(r + j) * rs] = acc;
        }
#pragma unroll
        for (int j = 0; j < G; ++j) cur[j] = nxt[j];
    }
}

// V staged: block = R columns (R*D contiguous floats per row), G rows per LDS chunk
template <int R, int G>
__global__ void pat_vc(const float* __restrict__ in, float* __restrict__ out, int H, int W, int D) {
    extern __shared__ float lds[];
    const int RD = R * D;             // floats per row piece
    const int RDp = (RD + 3) & ~3;    // padded to whole 16-byte vectors in LDS
    float* sin = lds;                 // [2][G][RDp]
    float* sout = lds + 2 * G * RDp;  // [G][RDp]
    const int nthreads = blockDim.x;
    const int c0 = blockIdx.x * R;
    const int tid = threadIdx.x;
    const bool owner = tid < RD && c0 * D + tid < W * D;
    const size_t rs = (size_t)W * D;
    const int vec = RDp / 4;
    const size_t base = (size_t)c0 * D;
    const int valid = min(RD, W * D - c0 * D);
    auto stage_in = [&](int buf, int r0) {
        for (int i = tid; i < G * vec; i += nthreads) {
            const int g = i / vec, v = i - g * vec;
            const float* src = in + (size_t)(r0 + g) * rs + base + 4 * v;
            float4 x;
            if (4 * v + 3 < valid) __builtin_memcpy(&x, src, 16);
            else { x.x = 4 * v < valid ? src[0] : 0.f; x.y = 4 * v + 1 < valid ? src[1] : 0.f; x.z = 4 * v + 2 < valid ? src[2] : 0.f; x.w = 0.f; }
            *reinterpret_cast<float4*>(sin + ((size_t)buf * G + g) * RDp + 4 * v) = x;
        }
    };
    float acc = 0.f;
    stage_in(0, 0);
    __syncthreads();
    int buf = 0;
    for (int r0 = 0; r0 < H; r0 += G) {
        if (r0 + G < H) stage_in(buf ^ 1, r0 + G);
        if (owner) {
#pragma unroll
            for (int j = 0; j < G; ++j) {
                acc += sin[((size_t)buf * G + j) * RDp + tid];
                sout[(size_t)j * RDp + tid] = acc;
            }
        }
        __syncthreads();
        for (int i = tid; i < G * vec; i += nthreads) {
            const int g = i / vec, v = i - g * vec;
            float* dst = out + (size_t)(r0 + g) * rs + base + 4 * v;
            const float4 x = *reinterpret_cast<const float4*>(sout + (size_t)g * RDp + 4 * v);
            if (4 * v + 3 < valid) __builtin_memcpy(dst, &x, 16);
            else { if (4 * v < valid) dst[0] = x.x; if (4 * v + 1 < valid) dst[1] = x.y; if (4 * v + 2 < valid) dst[2] = x.z; }
        }
        __syncthreads();
        buf ^= 1;
    }
}

int main(int argc, char** argv) {
    const int H = 2048, W = 2048, D = argc > 1 ? atoi(argv[1]) : 129;
    const size_t n = (size_t)H * W * D;
    float *in, *out;
    CK(hipMalloc(&in, n * 4 + 64));
    CK(hipMalloc(&out, n * 4 + 64));
    CK(hipMemset(in, 0, n * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch) {
        launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < 3; ++i) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 3;
        printf("%-44s %.3f ms  %.2f TB/s (read + write)\n", name, ms, 2.0 * n * 4 / ms / 1e9);
    };
    timeit("A  4 B/lane, 4 columns in flight", [&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(pat_a<4>), dim3((H * D + 255) / 256), dim3(256), 0, 0, in, out, H, W, D); });
    timeit("A  4 B/lane, 8 columns in flight", [&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(pat_a<8>), dim3((H * D + 255) / 256), dim3(256), 0, 0, in, out, H, W, D); });
    timeit("A  4 B/lane, 16 columns in flight", [&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(pat_a<16>), dim3((H * D + 255) / 256), dim3(256), 0, 0, in, out, H, W, D); });
    const int D2 = (D + 1) / 2;
    timeit("B  8 B/lane, 8 columns in flight", [&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(pat_b<8>), dim3((H * D2 + 255) / 256), dim3(256), 0, 0, in, out, H, W, D); });
    {
        constexpr int R = 4, G = 8;
        const int threads = ((R * D + 63) / 64) * 64;
        const size_t lds = (size_t)3 * R * G * D * 4;
        timeit("C  4 rows per block, 8 columns per LDS chunk", [&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(pat_c<R, G>), dim3((H + R - 1) / R), dim3(threads), lds, 0, in, out, H, W, D); });
    }
    {
        constexpr int R = 2, G = 8;
        const int threads = ((R * D + 63) / 64) * 64;
        const size_t lds = (size_t)3 * R * G * D * 4;
        timeit("C  2 rows per block, 8 columns per LDS chunk", [&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(pat_c<R, G>), dim3((H + R - 1) / R), dim3(threads), lds, 0, in, out, H, W, D); });
    }
    {
        constexpr int R = 2, G = 16;
        const int threads = ((R * D + 63) / 64) * 64;
        const size_t lds = (size_t)3 * R * G * D * 4;
        timeit("C  2 rows per block, 16 columns per LDS chunk", [&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(pat_c<R, G>), dim3((H + R - 1) / R), dim3(threads), lds, 0, in, out, H, W, D); });
    }
    timeit("VA 4 B/lane, 8 rows in flight", [&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(pat_va<8>), dim3((W * D + 255) / 256), dim3(256), 0, 0, in, out, H, W, D); });
#define VC(R, G)                                                                                                                       \
    {                                                                                                                                  \
        const int threads = ((R * D + 63) / 64) * 64;                                                                                  \
        const size_t lds = (size_t)3 * G * ((R * D + 3) & ~3) * 4;                                                                     \
        timeit("VC " #R " columns per block, " #G " rows per LDS chunk", [&] { hipLaunchKernelGGL(HIP_KERNEL_NAME(pat_vc<R, G>), dim3((W + R - 1) / R), dim3(threads), lds, 0, in, out, H, W, D); }); \
    }
    VC(2, 4) VC(2, 8) VC(4, 4) VC(4, 8) VC(8, 4)
    return 0;
}
