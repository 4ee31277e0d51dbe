// Micro-benchmark: per-wave instruction latency / throughput on gfx950 with 1, 2, 4 waves per SIMD,
// dependent vs independent integer VALU chains, and SALU chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 8192
template <int MODE>
__global__ void k(uint32_t* out, uint32_t seed) {
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    uint32_t s = __builtin_amdgcn_readfirstlane(seed | 3), t = s * 7;
    for (int r = 0; r < REP; ++r) {
        if (MODE == 0) {  // 8 dependent VALU pairs (one chain, 16 ops)
            a0 = min(a0 + s, a1); a0 = min(a0 + s, a2); a0 = min(a0 + s, a3); a0 = min(a0 + s, a4);
            a0 = min(a0 ^ s, a5); a0 = min(a0 + s, a6); a0 = min(a0 ^ s, a7); a0 = min(a0 + 1, a1);
        } else if (MODE == 1) {  // 8 independent chains, 2 ops each
            a0 = min(a0 + s, a1); a1 = min(a1 + s, a2); a2 = min(a2 + s, a3); a3 = min(a3 + s, a4);
            a4 = min(a4 + s, a5); a5 = min(a5 + s, a6); a6 = min(a6 + s, a7); a7 = min(a7 + s, a0);
        } else {  // SALU dependent chain
            s = s * 3 + t; t = t ^ (s >> 3); s = s + t; t = t * 5 + 1; s = s ^ t; t = t + (s << 1); s = s * 7; t = t ^ s;
            asm volatile("" : "+s"(s), "+s"(t));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + s + t;
}
template <int MODE>
void run(const char* name, int wps, int ops) {
    uint32_t* d;
    hipMalloc(&d, 1 << 22);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    int blocks = 256, threads = 64 * 4 * wps;  // wps waves per SIMD
    k<MODE><<<blocks, threads>>>(d, 1);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE><<<blocks, threads>>>(d, 2);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    double per_wave_ops = (double)REP * ops;
    printf("%-36s waves/SIMD=%d  %.3f ms -> %.1f ns/instr/wave (%.1f cyc @2.4GHz); SIMD rate %.2f cyc/instr\n", name,
           wps, ms, ms * 1e6 / per_wave_ops, ms * 1e6 / per_wave_ops * 2.4, ms * 1e6 / per_wave_ops * 2.4 / wps);
    hipFree(d);
}
int main() {
    for (int w : {1, 2, 4}) {
        run<0>("dependent VALU chain (16 ops)", w, 16);
        run<1>("8 independent VALU chains (16 ops)", w, 16);
        run<2>("dependent SALU chain (~12 ops)", w, 12);
    }
    return 0;
}
