// fetch_calib.hip - calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against KNOWN byte counts for the access widths the
// CBCA scans use (the guide's "FETCH_SIZE reports half of a wide coalesced read" is stated for 16 B per lane
// only).  Every kernel reads a 4 GiB buffer exactly once (larger than the 256 MB Infinity Cache) and writes 4 bytes per thread.
//   stream<B>   a wavefront reads 64 * B contiguous bytes per instruction, B = 4 / 8 / 16 per lane, grid-stride
//   rows4       pass V's pattern: thread = (column, disparity) of a [H][W][D] float32 volume marching down the rows - a
//               workgroup's 256 cells are 1 KB contiguous, the next access is W * D * 4 bytes further
//   rows4x2     the same plus a second 4-byte stream per cell from a small [H][W] array at column c + d (the right image's arms)
// Run:  rocprofv3 --pmc FETCH_SIZE -d out -o f -- tools/ubench/fetch_calib ; rocprofv3 --pmc WRITE_SIZE ... ; tools/rocpd_pmc.py
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/fetch_calib tools/ubench/fetch_calib.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));

template <int B>
__global__ __launch_bounds__(256) void stream(const char* __restrict__ in, size_t bytes, uint32_t* __restrict__ out) {
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nthreads = (size_t)gridDim.x * 256;
    uint32_t acc = 0;
    for (size_t off = tid * B; off + B <= bytes; off += nthreads * B) {
        if (B == 16) { const u4 v = *(const u4*)(in + off); acc += v.x ^ v.y ^ v.z ^ v.w; }
        if (B == 8) { const u2 v = *(const u2*)(in + off); acc += v.x ^ v.y; }
        if (B == 4) acc += *(const uint32_t*)(in + off);
    }
    out[tid] = acc;
}

template <bool TWO>
__global__ __launch_bounds__(256) void rows4(const float* __restrict__ vol, const uint32_t* __restrict__ arms, int H, int W, int D,
                                             uint32_t* __restrict__ out) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;  // (column, disparity) cell of a row
    if (t >= (size_t)W * D) return;
    const int c = (int)(t / D), d = (int)(t - (size_t)c * D);
    const int q = min(c + d, W - 1);
    float acc = 0.f;
    uint32_t a = 0;
    for (int r = 0; r < H; ++r) {
        acc += vol[(size_t)r * W * D + t];
        if (TWO) a += arms[(size_t)r * W + q];
    }
    out[t] = __float_as_uint(acc) + a;
}

int main() {
    const size_t bytes = 4ull << 30;
    char* in;
    uint32_t *out, *arms;
    hipMalloc(&in, bytes);
    hipMemset(in, 1, bytes);
    const int W = 4096, D = 128, H = (int)(bytes / ((size_t)W * D * 4));  // 2048 rows of 2 MB
    hipMalloc(&out, (size_t)W * D * 4 + 256 * 8192 * 4);
    hipMalloc(&arms, (size_t)H * W * 4);
    hipMemset(arms, 0, (size_t)H * W * 4);
    hipLaunchKernelGGL(stream<16>, dim3(8192), dim3(256), 0, 0, in, bytes, out);
    hipLaunchKernelGGL(stream<8>, dim3(8192), dim3(256), 0, 0, in, bytes, out);
    hipLaunchKernelGGL(stream<4>, dim3(8192), dim3(256), 0, 0, in, bytes, out);
    hipLaunchKernelGGL(rows4<false>, dim3((W * D + 255) / 256), dim3(256), 0, 0, (const float*)in, arms, H, W, D, out);
    hipLaunchKernelGGL(rows4<true>, dim3((W * D + 255) / 256), dim3(256), 0, 0, (const float*)in, arms, H, W, D, out);
    hipDeviceSynchronize();
    printf("known bytes read: stream<16|8|4> and rows4 %zu (4 GiB = 4194304 KiB); rows4<two> + %zu arm bytes requested (%zu KiB unique)\n", bytes,
           (size_t)H * W * D * 4, (size_t)H * W * 4 / 1024);
    printf("known bytes written: stream %d KiB, rows4 %d KiB\n", 8192 * 256 * 4 / 1024, W * D * 4 / 1024);
    return 0;
}
