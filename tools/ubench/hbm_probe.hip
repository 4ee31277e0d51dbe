// hbm_probe.hip - what plain streaming kernels reach on this device: read / fill / copy in several shapes (the
// guide measures 6.29 TB/s for a float4 copy, pmx_measure_hbm's grid-stride probe got 4.9 - 5.1 on this pool's boxes).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/hbm_probe.hip -o tools/ubench/_bin/hbm_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_k(const v4u* __restrict__ src, v4u* __restrict__ dst, size_t n) {
    const size_t step = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * step < n; i += U * step) {
        v4u v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * step) : src[i + u * step];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], dst + i + u * step); else dst[i + u * step] = v[u]; }
    }
    for (; i < n; i += step) dst[i] = src[i];
}
// a block owns a contiguous chunk (the guide's shape: consecutive blocks on consecutive chunks)
template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_chunk_k(const v4u* __restrict__ src, v4u* __restrict__ dst, size_t n) {
    const size_t per = (n + gridDim.x - 1) / gridDim.x;
    const size_t b0 = (size_t)blockIdx.x * per, b1 = b0 + per < n ? b0 + per : n;
    size_t i = b0 + threadIdx.x;
    for (; i + (U - 1) * 256 < b1; i += U * 256) {
        v4u v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * 256) : src[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], dst + i + u * 256); else dst[i + u * 256] = v[u]; }
    }
    for (; i < b1; i += 256) dst[i] = src[i];
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void read_k(const v4u* __restrict__ src, size_t n, unsigned* sink) {
    const size_t step = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    unsigned acc = 0;
    for (; i + (U - 1) * step < n; i += U * step) {
        v4u v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * step) : src[i + u * step];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x9e3779b9u) sink[0] = acc;
}
template <bool NT>
__global__ __launch_bounds__(256) void fill_k(v4u* __restrict__ dst, size_t n, unsigned v) {
    const size_t step = (size_t)gridDim.x * 256;
    const v4u w = {v, v + 1, v + 2, v + 3};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += step) { if (NT) __builtin_nontemporal_store(w, dst + i); else dst[i] = w; }
}

int main(int argc, char** argv) {
    const size_t bytes = (argc > 1 ? (size_t)atoll(argv[1]) : (size_t)4096) << 20;
    const size_t n = bytes / 16;
    v4u *a, *b; unsigned* sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, double moved, auto launch) {
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r && ms < best) best = ms;
        }
        printf("%-44s %8.3f ms  %7.1f GB/s\n", name, best, moved / (best * 1e-3) / 1e9);
    };
    const int grids[] = {1024, 2048, 4096, 16384, 65536};
    for (int g : grids) {
        char nm[96];
        snprintf(nm, 96, "copy  stride U1 plain  grid %d", g); timeit(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL((copy_k<1, false>), dim3(g), dim3(256), 0, 0, a, b, n); });
        snprintf(nm, 96, "copy  stride U4 plain  grid %d", g); timeit(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL((copy_k<4, false>), dim3(g), dim3(256), 0, 0, a, b, n); });
        snprintf(nm, 96, "copy  stride U4 nt     grid %d", g); timeit(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL((copy_k<4, true>), dim3(g), dim3(256), 0, 0, a, b, n); });
        snprintf(nm, 96, "copy  chunk  U4 plain  grid %d", g); timeit(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL((copy_chunk_k<4, false>), dim3(g), dim3(256), 0, 0, a, b, n); });
        snprintf(nm, 96, "copy  chunk  U4 nt     grid %d", g); timeit(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL((copy_chunk_k<4, true>), dim3(g), dim3(256), 0, 0, a, b, n); });
        snprintf(nm, 96, "read  stride U4 plain  grid %d", g); timeit(nm, 1.0 * bytes, [&] { hipLaunchKernelGGL((read_k<4, false>), dim3(g), dim3(256), 0, 0, a, n, sink); });
        snprintf(nm, 96, "read  stride U8 plain  grid %d", g); timeit(nm, 1.0 * bytes, [&] { hipLaunchKernelGGL((read_k<8, false>), dim3(g), dim3(256), 0, 0, a, n, sink); });
        snprintf(nm, 96, "fill  plain            grid %d", g); timeit(nm, 1.0 * bytes, [&] { hipLaunchKernelGGL((fill_k<false>), dim3(g), dim3(256), 0, 0, b, n, 7u); });
        snprintf(nm, 96, "fill  nt               grid %d", g); timeit(nm, 1.0 * bytes, [&] { hipLaunchKernelGGL((fill_k<true>), dim3(g), dim3(256), 0, 0, b, n, 7u); });
    }
    timeit("hipMemcpyAsync d2d", 2.0 * bytes, [&] { CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); });
    return 0;
}
