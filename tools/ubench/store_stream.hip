// store_stream.hip - what a kernel of ZNCC's shape gets out of its store stream (round 6, docs/experiments.md 7.32): workgroups of 4
// wavefronts, `lds_kb` of LDS each (occupancy: 40 -> 4 workgroups per CU, 16 wavefronts), marching over `rows` rows; per row every
// thread does `work` dependent FMAs (the row's arithmetic) and `nst` 16-byte stores (a wavefront's store = 1 KB contiguous, the
// workgroup's row = 4 nst KB contiguous, rows `pitch` bytes apart).  Reports ms and TB/s of the stores.
// hipcc --offload-arch=gfx950 -O3 -o tools/ubench/_bin/store_stream tools/ubench/store_stream.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

// LOADS: every row also requests 4 bytes per thread from a small (L2-resident) image one row ahead and uses last row's request - as
// ZNCC's rows do.  A wavefront's memory counter is one and in order: the wait for a load is a wait for every store issued before it.
template <bool LOADS>
__global__ __launch_bounds__(256) void march(u4* out, size_t pitch16, int rows, int work, int nst, float seed, const float* img) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    u4* p = out + (size_t)blockIdx.x * rows * pitch16 + tid;
    float a = seed + tid, b = 1.0001f;
    lds[tid] = a;
    const float* q = img + ((size_t)blockIdx.x * 977 + tid) % (1 << 20);
    float cur = LOADS ? q[0] : 0.f;
    for (int r = 0; r < rows; ++r) {
        float nxt = 0.f;
        if (LOADS) nxt = q[(size_t)(r + 1) * 4096 % (1 << 20)];  // requested now, used next row
        a += cur;
        cur = nxt;
        for (int i = 0; i < work; ++i) a = a * b + 0.5f;  // dependent chain: the row's arithmetic
        u4 v;
        v.x = __float_as_uint(a); v.y = v.x + 1; v.z = v.x + 2; v.w = v.x + 3;
        for (int s = 0; s < nst; ++s) p[(size_t)s * 256] = v;
        p += pitch16;
        __syncthreads();
    }
    if (a == 123.456f) lds[0] = a;
}

// ZNCC's own addresses: a workgroup = (tile of 54 columns, strip of 64 rows, block of 32 disparities) of a [H][W][D] float volume;
// thread = (pixel, 16-byte piece of the block's 128 bytes), two passes of 32 pixels; D = 257: a pixel's run starts at 4-byte alignment
__global__ __launch_bounds__(256) void march_zncc(float* out, int W, int D, int ntile, int ndblock, int rows, int work, int stores, float seed) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    const unsigned xcd = blockIdx.x & 7u, seq = blockIdx.x >> 3;
    const unsigned group = (seq / (unsigned)ndblock) * 8u + xcd;
    const int dblock = seq % (unsigned)ndblock, tile = group % ntile, strip = group / ntile;
    const int part = tid & 7, pix0 = tid >> 3, k = dblock * 32 + 4 * part;
    float a = seed + tid, b = 1.0001f;
    lds[tid] = a;
    for (int r = 0; r < rows; ++r) {
        for (int i = 0; i < work; ++i) a = a * b + 0.5f;
        if (stores) {
            for (int pass = 0; pass < 2; ++pass) {
                const int pix = pix0 + 32 * pass, pc = tile * 54 + pix;
                if (pix < 54 && pc < W && k + 4 <= D) {
                    float* dst = out + ((size_t)(strip * rows + r) * W + pc) * D + k;
                    const float4 v = make_float4(a, a + 1, a + 2, a + 3);
                    __builtin_memcpy(dst, &v, 16);
                }
            }
        }
        __syncthreads();
    }
    if (a == 123.456f) lds[0] = a;
}

int main(int argc, char** argv) {
    const int lds_kb = argc > 1 ? atoi(argv[1]) : 40;
    const int rows = 64;
    size_t cap = (size_t)18 << 30;
    u4* buf = nullptr;
    if (hipMalloc(&buf, cap) != hipSuccess) return 1;
    hipMemset(buf, 0, cap);
    hipFuncSetAttribute((const void*)march<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024);
    hipFuncSetAttribute((const void*)march<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("# %d KB of LDS per workgroup of 256 threads, %d rows per workgroup\n", lds_kb, rows);
    float* img = nullptr;
    hipMalloc(&img, (size_t)(2 << 20) * 4);
    hipMemset(img, 0, (size_t)(2 << 20) * 4);
    const bool loads = argc > 2 && atoi(argv[2]);
    printf("# loads per row: %s\n", loads ? "yes (one dword per thread, one row ahead)" : "no");
    printf("# work nst   ms     TB/s   (ms with no stores)\n");
    for (int nst : {2, 4, 8}) {
        for (int work : {0, 100, 200, 400, 800}) {
            const size_t row_bytes = (size_t)nst * 4096;
            const size_t wg_bytes = row_bytes * rows;
            const int nwg = (int)(cap / wg_bytes);
            const size_t total = (size_t)nwg * wg_bytes;
            float ms[2] = {0, 0};
            for (int k = 0; k < 2; ++k) {
                const int ns = k == 0 ? nst : 0;
                for (int rep = 0; rep < 3; ++rep) {
                    hipEventRecord(e0);
                    if (loads) hipLaunchKernelGGL(march<true>, dim3(nwg), dim3(256), lds_kb * 1024, 0, buf, row_bytes / 16, rows, work, ns, 1.0f + rep, img);
                    else hipLaunchKernelGGL(march<false>, dim3(nwg), dim3(256), lds_kb * 1024, 0, buf, row_bytes / 16, rows, work, ns, 1.0f + rep, img);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float t;
                    hipEventElapsedTime(&t, e0, e1);
                    if (rep == 0 || t < ms[k]) ms[k] = t;
                }
            }
            printf("%5d %3d %7.3f %6.2f   %7.3f\n", work, nst, ms[0], total / (ms[0] * 1e-3) / 1e12, ms[1]);
        }
    }
    if (argc > 3) {
        for (int D : {256, 257}) {
            const int W = 4096, H = 4096, ntile = (W + 53) / 54, ndblock = (D + 31) / 32, nstrip = H / rows;
            const int nwg = ((ntile * nstrip + 7) / 8) * 8 * ndblock;
            hipFuncSetAttribute((const void*)march_zncc, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024);
            printf("# ZNCC's store pattern, %d x %d x %d (%.2f GB), %d workgroups\n# work   ms     TB/s   (ms with no stores)\n", H, W, D, (double)H * W * D * 4 / 1e9, nwg);
            for (int work : {0, 100, 200, 300, 400}) {
                float ms[2] = {0, 0};
                for (int k = 0; k < 2; ++k)
                    for (int rep = 0; rep < 3; ++rep) {
                        hipEventRecord(e0);
                        hipLaunchKernelGGL(march_zncc, dim3(nwg), dim3(256), lds_kb * 1024, 0, (float*)buf, W, D, ntile, ndblock, rows, work, k == 0, 1.0f + rep);
                        hipEventRecord(e1);
                        hipEventSynchronize(e1);
                        float t;
                        hipEventElapsedTime(&t, e0, e1);
                        if (rep == 0 || t < ms[k]) ms[k] = t;
                    }
                printf("%5d %7.3f %6.2f   %7.3f\n", work, ms[0], (double)H * W * D * 4 / (ms[0] * 1e-3) / 1e12, ms[1]);
            }
        }
    }
    return 0;
}
