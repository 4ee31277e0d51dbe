// What one wavefront (and 2, 4 per SIMD) gets out of the vector unit per instruction kind: dependent chains and four interleaved
// chains of the instructions the integer SGM kernels are made of.  Build: hipcc --offload-arch=gfx950 -O3 -o issue_rate issue_rate.hip
// Usage: ./issue_rate   (prints cycles per instruction and wavefront at the clock measured by s_memtime)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

#define KERNEL(name, dep_asm, ind_asm)                                                                                  \
    __global__ void name##_dep(uint32_t* out, int iters, uint32_t seed) {                                               \
        uint32_t a = threadIdx.x + seed, b = seed * 3 + 1, c = seed + 7, d = 5;                                         \
        for (int i = 0; i < iters; ++i) { asm volatile(REP64(dep_asm) : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }          \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;                                                     \
    }                                                                                                                   \
    __global__ void name##_ind(uint32_t* out, int iters, uint32_t seed) {                                               \
        uint32_t a = threadIdx.x + seed, b = seed * 3 + 1, c = seed + 7, d = 5;                                         \
        for (int i = 0; i < iters; ++i) { asm volatile(REP64(ind_asm) : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); }          \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;                                                     \
    }

// dep: every instruction reads the one before; ind: four chains a, b, c, d in turn (REP64 of a 4-instruction group = 256 instr)
KERNEL(add, "v_add_u32 %0, %0, %1\n", "v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n")
KERNEL(add3, "v_add3_u32 %0, %0, %1, %2\n", "v_add3_u32 %0, %0, 1, 2\n v_add3_u32 %1, %1, 1, 2\n v_add3_u32 %2, %2, 1, 2\n v_add3_u32 %3, %3, 1, 2\n")
KERNEL(pkmin, "v_pk_min_f16 %0, %0, %1\n", "v_pk_min_f16 %0, %0, %0\n v_pk_min_f16 %1, %1, %1\n v_pk_min_f16 %2, %2, %2\n v_pk_min_f16 %3, %3, %3\n")
KERNEL(pkmin3, "v_pk_minimum3_f16 %0, %0, %1, %2\n", "v_pk_minimum3_f16 %0, %0, %0, %0\n v_pk_minimum3_f16 %1, %1, %1, %1\n v_pk_minimum3_f16 %2, %2, %2, %2\n v_pk_minimum3_f16 %3, %3, %3, %3\n")
KERNEL(pkminu, "v_pk_min_u16 %0, %0, %1\n", "v_pk_min_u16 %0, %0, %0\n v_pk_min_u16 %1, %1, %1\n v_pk_min_u16 %2, %2, %2\n v_pk_min_u16 %3, %3, %3\n")
KERNEL(min3u, "v_min3_u32 %0, %0, %1, %2\n", "v_min3_u32 %0, %0, %0, %0\n v_min3_u32 %1, %1, %1, %1\n v_min3_u32 %2, %2, %2, %2\n v_min3_u32 %3, %3, %3, %3\n")
KERNEL(perm, "v_perm_b32 %0, %0, %1, %2\n", "v_perm_b32 %0, %0, %0, %0\n v_perm_b32 %1, %1, %1, %1\n v_perm_b32 %2, %2, %2, %2\n v_perm_b32 %3, %3, %3, %3\n")
KERNEL(alignb, "v_alignbit_b32 %0, %0, %1, 16\n", "v_alignbit_b32 %0, %0, %0, 16\n v_alignbit_b32 %1, %1, %1, 16\n v_alignbit_b32 %2, %2, %2, 16\n v_alignbit_b32 %3, %3, %3, 16\n")
KERNEL(lshlor, "v_lshl_or_b32 %0, %0, 1, %1\n", "v_lshl_or_b32 %0, %0, 1, 1\n v_lshl_or_b32 %1, %1, 1, 1\n v_lshl_or_b32 %2, %2, 1, 1\n v_lshl_or_b32 %3, %3, 1, 1\n")
KERNEL(bcnt, "v_bcnt_u32_b32 %0, %0, %1\n", "v_bcnt_u32_b32 %0, %0, 1\n v_bcnt_u32_b32 %1, %1, 1\n v_bcnt_u32_b32 %2, %2, 1\n v_bcnt_u32_b32 %3, %3, 1\n")
KERNEL(dppmin, "s_nop 1\n v_min_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n", "v_min_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n")
KERNEL(dppbc, "s_nop 1\n v_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n", "v_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_min_u32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_min_u32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_min_u32_dpp %3, %3, %3 row_bcast:15 row_mask:0xa bank_mask:0xf\n")
KERNEL(dppws, "s_nop 1\n v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n", "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n")
KERNEL(sdwa, "v_min_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n", "v_min_u32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n v_min_u32_sdwa %1, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n v_min_u32_sdwa %2, %2, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n v_min_u32_sdwa %3, %3, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n")
KERNEL(snop, "s_nop 0\n", "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n")

// the same dependent v_add3 chain in loop bodies of 64 .. 8192 instructions (8 bytes each): what instruction fetch costs a wavefront
#define REP512(x) REP8(REP64(x))
#define REP4096(x) REP8(REP512(x))
#define BODY(name, REPN)                                                                                               \
    __global__ void name(uint32_t* out, int iters, uint32_t seed) {                                                    \
        uint32_t a = threadIdx.x + seed, b = seed * 3 + 1, c = seed + 7;                                               \
        for (int i = 0; i < iters; ++i) { asm volatile(REPN("v_add3_u32 %0, %0, %1, %2\n") : "+v"(a), "+v"(b), "+v"(c)); } \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c;                                                        \
    }
BODY(body64, REP64)
BODY(body512, REP512)
#define REP2048(x) REP512(x) REP512(x) REP512(x) REP512(x)
BODY(body2048, REP2048)
BODY(body4096, REP4096)
#define REP8192(x) REP4096(x) REP4096(x)
BODY(body8192, REP8192)

__global__ void clock_probe(unsigned long long* out) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long w0 = wall_clock64();
    for (int i = 0; i < 200000; ++i) asm volatile("s_nop 7");
    out[0] = __builtin_readcyclecounter() - t0;
    out[1] = wall_clock64() - w0;
}

template <typename K>
static double time_ms(K k, int blocks, int threads, uint32_t* out, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, 8, 1u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, iters, 1u);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    uint32_t* out;
    (void)hipMalloc(&out, 1 << 24);
    const double ghz = 2.4;  // (GRBM_GUI_ACTIVE / duration of the headline's kernels, profiles/r04_h_northstar_pmc_clk.csv)
    const int iters = 4000;
    printf("cycles per instruction and wavefront at %.1f GHz; waves per SIMD = 1, 2, 4 (256 CUs x 4 SIMDs)\n", ghz);
    printf("%-10s %8s %8s %8s   %8s %8s %8s\n", "instr", "dep x1", "dep x2", "dep x4", "ind x1", "ind x2", "ind x4");
#define ROW(name, ndep, nind)                                                                                  \
    {                                                                                                          \
        double r[6];                                                                                           \
        int wps[3] = {1, 2, 4};                                                                                \
        for (int i = 0; i < 3; ++i) {                                                                          \
            r[i] = time_ms(name##_dep, 256, 256 * wps[i], out, iters) * 1e-3 * ghz * 1e9 / ((double)iters * ndep); \
            r[3 + i] = time_ms(name##_ind, 256, 256 * wps[i], out, iters) * 1e-3 * ghz * 1e9 / ((double)iters * nind); \
        }                                                                                                      \
        printf("%-10s %8.2f %8.2f %8.2f   %8.2f %8.2f %8.2f\n", #name, r[0], r[1], r[2], r[3], r[4], r[5]);       \
    }
    ROW(add, 64, 256) ROW(add3, 64, 256) ROW(pkmin, 64, 256) ROW(pkmin3, 64, 256) ROW(pkminu, 64, 256) ROW(min3u, 64, 256)
    ROW(perm, 64, 256) ROW(alignb, 64, 256) ROW(lshlor, 64, 256) ROW(bcnt, 64, 256) ROW(dppmin, 128, 256) ROW(dppbc, 128, 256)
    ROW(dppws, 128, 256) ROW(sdwa, 64, 256) ROW(snop, 64, 256)
    printf("\nloop body length (dependent v_add3_u32, 8 bytes each): cycles per instruction and wavefront at 1 / 2 / 4 waves per SIMD\n");
#define BROW(name, n)                                                                                          \
    {                                                                                                          \
        double r[3];                                                                                           \
        int wps[3] = {1, 2, 4};                                                                                \
        for (int i = 0; i < 3; ++i) r[i] = time_ms(name, 256, 256 * wps[i], out, 262144 / n) * 1e-3 * ghz * 1e9 / (262144.0); \
        printf("%-10s %8.2f %8.2f %8.2f\n", #name, r[0], r[1], r[2]);                                          \
    }
    BROW(body64, 64) BROW(body512, 512) BROW(body2048, 2048) BROW(body4096, 4096) BROW(body8192, 8192)
    return 0;
}
