// Micro-benchmark: issue rate of the float64 / conversion ops the ZNCC kernel leans on (gfx950, wave64, 4 waves/SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP 4096
template <int OP>
__global__ __launch_bounds__(256) void k(double* out, double seed) {
    double a[8];
    float f[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * (i + 1); f[i] = (float)a[i]; }
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) a[i] = a[i] + a[(i + 1) & 7];                      // v_add_f64
            if (OP == 1) a[i] = a[i] * a[(i + 1) & 7];                      // v_mul_f64
            if (OP == 2) a[i] = __builtin_fma(a[i], a[(i + 1) & 7], a[(i + 2) & 7]);  // v_fma_f64
            if (OP == 3) { a[i] = a[i] + (double)f[i]; }                    // v_cvt_f64_f32 + v_add_f64
            if (OP == 4) { f[i] = (float)(a[i]) + f[(i + 1) & 7]; }         // v_cvt_f32_f64 + v_add_f32
            if (OP == 5) { f[i] = f[i] * f[(i + 1) & 7]; }                  // v_mul_f32
        }
    }
    double acc = 0;
    for (int i = 0; i < 8; ++i) acc += a[i] + (double)f[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int OP>
void run(const char* name, int ops_per_iter) {
    double* d;
    hipMalloc(&d, 256 * 1024 * 4 * sizeof(double));
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    int blocks = 256 * 4;
    k<OP><<<blocks, 256>>>(d, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<OP><<<blocks, 256>>>(d, 2.0);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    double per_simd = (double)blocks * 4 * REP * 8 * ops_per_iter / 1024.0;
    printf("%-28s %.3f ms  -> %.2f cycles per wave-instruction per SIMD @2.4GHz\n", name, ms, ms * 1e6 / per_simd * 2.4);
    hipFree(d);
}

int main() {
    run<0>("add_f64", 1);
    run<1>("mul_f64", 1);
    run<2>("fma_f64", 1);
    run<3>("cvt_f64_f32+add_f64", 2);
    run<4>("cvt_f32_f64+add_f32", 2);
    run<5>("mul_f32", 1);
    return 0;
}
