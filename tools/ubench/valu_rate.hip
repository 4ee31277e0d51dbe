// Micro-benchmark: issue rate of a few VALU ops on gfx950 (wave64), 4 waves/SIMD, independent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP 4096
template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint32_t seed) {
    uint32_t a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * (i + 1);
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = (float)a[i];
    uint32_t s = seed | 3;
    float fs = (float)s;
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) a[i] = min(a[i] + s, a[(i + 1) & 7]);                    // v_add_u32 + v_min_u32
            if (OP == 1) f[i] = fminf(f[i] + fs, f[(i + 1) & 7]);                 // v_add_f32 + v_min_f32
            if (OP == 2) a[i] = __popc(a[i] ^ a[(i + 1) & 7]) + s;                // v_xor + v_bcnt
            if (OP == 3) a[i] = min(min(a[i], a[(i + 1) & 7]), a[(i + 2) & 7]) + 1;  // v_min3_u32 + add
            if (OP == 4) {  // packed u16 min + add
                typedef unsigned short us2 __attribute__((ext_vector_type(2)));
                us2 x = __builtin_bit_cast(us2, a[i]), y = __builtin_bit_cast(us2, a[(i + 1) & 7]);
                us2 z = __builtin_elementwise_min(x, y) + __builtin_bit_cast(us2, s);
                a[i] = __builtin_bit_cast(uint32_t, z);
            }
            if (OP == 5) f[i] = fminf(fminf(f[i], f[(i + 1) & 7]), f[(i + 2) & 7]) + 1.0f;  // v_min3_f32 + add
        }
    }
    uint32_t acc = 0;
    for (int i = 0; i < 8; ++i) acc += a[i] + (uint32_t)f[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int OP>
void run(const char* name, int ops_per_iter) {
    uint32_t* d;
    hipMalloc(&d, 256 * 1024 * 4 * sizeof(uint32_t));
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    int blocks = 256 * 4;  // 4 blocks/CU = 4 waves/SIMD
    k<OP><<<blocks, 256>>>(d, 1);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<OP><<<blocks, 256>>>(d, 2);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    double wave_instr = (double)blocks * 4 * REP * 8 * ops_per_iter;  // per whole chip
    double per_simd = wave_instr / 1024.0;
    printf("%-28s %.3f ms  -> %.2f ns per wave-instruction per SIMD (%.2f cycles @2.4GHz)\n", name, ms, ms * 1e6 / per_simd,
           ms * 1e6 / per_simd * 2.4);
    hipFree(d);
}

int main() {
    run<0>("add_u32+min_u32", 2);
    run<1>("add_f32+min_f32", 2);
    run<2>("xor+bcnt(+add)", 2);
    run<3>("min3_u32+add_u32", 2);
    run<4>("pk_min_u16+pk_add_u16", 2);
    run<5>("min3_f32+add_f32", 2);
    return 0;
}
