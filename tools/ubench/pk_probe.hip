// Micro-benchmark for the packed integer SGM family kernel (k_sgmfam8.hip), gfx950:
//   (1) are small integers carried as f16 DENORMAL bit patterns exact under v_pk_add_f16 / v_pk_min_f16 / v_pk_minimum3_f16?
//       (pattern n = n * 2^-24 for n < 2048: sums and minima of such values are the integer results while they stay < 2048)
//   (2) issue rates of the instructions the kernel is made of, at 1 / 2 / 4 waves per SIMD, in shader cycles (s_memtime) and ns.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/pk_probe.hip -o tools/ubench/pk_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

__device__ __forceinline__ uint32_t f16_add(uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_pk_add_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t f16_sub(uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t f16_min(uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_pk_min_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ uint32_t f16_min3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t d;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// ---- exactness -------------------------------------------------------------------------------------------------------------
__global__ void exact_kernel(unsigned long long* bad) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;  // 0 .. 2047
    unsigned long long nbad = 0;
    for (uint32_t y = 0; y < 2048; ++y) {
        const uint32_t px = x | (y << 16), py = y | (x << 16);
        if (x + y < 2048) {
            const uint32_t s = f16_add(px, py);
            nbad += (s != ((x + y) | ((x + y) << 16)));
        }
        const uint32_t m = f16_min(px, py), mn = x < y ? x : y;
        nbad += (m != (mn | (mn << 16)));
        const uint32_t z = (x * 7 + y * 13) & 2047;
        const uint32_t m3 = f16_min3(px, py, z | (z << 16));
        const uint32_t lo = min(min(x, y), z), hi = min(min(y, x), z);
        nbad += (m3 != (lo | (hi << 16)));
        if (x >= y) {
            const uint32_t d = f16_sub(x | (x << 16), y | (y << 16));
            nbad += (d != ((x - y) | ((x - y) << 16)));
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}

// ---- rates -----------------------------------------------------------------------------------------------------------------
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t u16_min(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
}
__device__ __forceinline__ uint32_t u16_add(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, (us2)(__builtin_bit_cast(us2, a) + __builtin_bit_cast(us2, b)));
}

#define REP 2048
template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(uint32_t* out, unsigned long long* cyc, uint32_t seed) {
    uint32_t a[8];
    for (int i = 0; i < 8; ++i) a[i] = ((seed + threadIdx.x * (i + 1)) & 0x003f003fu);
    const uint32_t s = (seed & 3) | ((seed & 3) << 16);
    const int idx = ((threadIdx.x + 16) & 63) * 4;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t n = a[(i + 1) & 7];
            if (OP == 0) a[i] = u16_min(u16_add(a[i], s), n);                       // 2: v_pk_add_u16 + v_pk_min_u16
            if (OP == 1) a[i] = f16_min(f16_add(a[i], s), n);                       // 2: v_pk_add_f16 + v_pk_min_f16 (denormals)
            if (OP == 2) a[i] = f16_min3(f16_add(a[i], s), n, a[(i + 2) & 7]);      // 2: v_pk_add_f16 + v_pk_minimum3_f16
            if (OP == 3) a[i] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)n, 0x93, 0xf, 0xf, false) + a[i];  // 2: v_mov_dpp quad_perm:[3,0,1,2] (may fold) + add
            if (OP == 4) a[i] = __builtin_amdgcn_alignbit(a[i], n, 16) + s;         // 2: v_alignbit + v_add
            if (OP == 5) a[i] = (uint32_t)__builtin_amdgcn_ds_bpermute(idx, (int)n) + a[i];  // 2: ds_bpermute + add
            if (OP == 6) a[i] = ((n >> 5) & 0x001f001fu) | a[i];                    // 2: v_lshrrev + v_and_or
            if (OP == 7) a[i] = __builtin_amdgcn_permlane32_swap(a[i], n, false, false)[0] + s;  // 2: permlane32_swap + add
            if (OP == 8) a[i] = a[i] + n;                                           // 1: v_add_u32
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t acc = 0;
    for (int i = 0; i < 8; ++i) acc += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) atomicAdd(cyc, t1 - t0);
}

template <int OP>
void run(const char* name, int ops) {
    uint32_t* d;
    unsigned long long* c;
    hipMalloc(&d, 256 * 8 * 256 * sizeof(uint32_t));
    hipMalloc(&c, 8);
    for (int per_cu : {1, 2, 4, 8}) {
        const int blocks = 256 * per_cu;  // 256-thread blocks: per_cu blocks per CU = per_cu waves per SIMD
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        rate_kernel<OP><<<blocks, 256>>>(d, c, 1);
        hipDeviceSynchronize();
        hipMemset(c, 0, 8);
        hipEventRecord(e0);
        rate_kernel<OP><<<blocks, 256>>>(d, c, 2);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long cy;
        hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
        const double per_wave_cycles = (double)cy / (blocks * 4);
        const double instr_per_wave = (double)REP * 8 * ops;
        const double per_simd = instr_per_wave * per_cu;  // instructions a SIMD issued
        printf("%-34s %d waves/SIMD: %.3f ms, %.2f ns/instr/SIMD, wave-time %.0f cyc -> %.2f cyc/instr/SIMD (memtime), clock %.2f GHz\n", name,
               per_cu, ms, ms * 1e6 / per_simd, per_wave_cycles, per_wave_cycles / per_simd, per_wave_cycles / (ms * 1e6));
    }
    hipFree(d);
    hipFree(c);
}

int main() {
    unsigned long long* bad;
    hipMalloc(&bad, 8);
    hipMemset(bad, 0, 8);
    exact_kernel<<<8, 256>>>(bad);
    unsigned long long nb = 0;
    hipMemcpy(&nb, bad, 8, hipMemcpyDeviceToHost);
    printf("f16-denormal integer arithmetic (add / sub / min / minimum3 on 0..2047): %llu mismatches\n", nb);
    run<0>("pk_add_u16+pk_min_u16", 2);
    run<1>("pk_add_f16+pk_min_f16 (denorm)", 2);
    run<2>("pk_add_f16+pk_minimum3_f16", 2);
    run<3>("mov_dpp quad_perm+add", 2);
    run<4>("alignbit+add", 2);
    run<5>("ds_bpermute+add", 2);
    run<6>("lshr+and_or", 2);
    run<7>("permlane32_swap+add", 2);
    run<8>("add_u32", 1);
    return 0;
}
