// Micro-benchmark: reading 8 byte streams at the same pixel offset (what sum8_wta_kernel does) from (A) eight volumes
// H*W*Dp bytes apart, (B) ONE buffer in which blocks of 32 pixels hold their 8 directions back to back.  Also the mirrored
// write pattern (8 concurrent writers, what the SGM path kernel does).  Run several processes: is A bimodal, is B steady?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <ctime>

constexpr int DP = 132, BLK = 32;

// 4 pixels per wave, 16 lanes per pixel, lane reads 3 dwords per direction (lanes 0..10 active)
template <int LAYOUT>
__global__ __launch_bounds__(256) void read8(const uint8_t* __restrict__ base, size_t npix, size_t vol, uint32_t* __restrict__ out) {
    const int lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (size_t)gridDim.x * 4;
    uint32_t acc = 0;
    for (size_t quad = wave; quad * 4 < npix; quad += nw) {
        const size_t pix = quad * 4 + grp;
        const int off = (sub < 11 ? sub : 0) * 12;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint8_t* p = LAYOUT == 0 ? base + (size_t)k * vol + pix * DP + off
                                           : base + ((pix / BLK) * 8 + k) * (size_t)(BLK * DP) + (pix % BLK) * DP + off;
            uint32_t x[3];
            __builtin_memcpy(x, p, 12);
            acc += x[0] + x[1] + x[2];
        }
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
    const size_t H = 2048, W = 2048, npix = H * W, vol = npix * DP;
    uint8_t* buf; uint32_t* out;
    hipMalloc(&buf, 8 * vol + 4096);
    hipMalloc(&out, 65536 * 256 * 4);
    hipMemset(buf, 1, 8 * vol + 4096);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    // VMM allocation: physical chunks of a chosen size mapped into one contiguous virtual range
    if (argc > 2) {
        const size_t chunk = (size_t)atoll(argv[2]) << 20;  // MiB
        const int nb = atoi(argv[1]);
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        size_t gran = 0;
        (void)hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
        printf("recommended granularity %zu, chunk %zu\n", gran, chunk);
        for (int i = 0; i < nb; ++i) {
            struct timespec t0, t1;
            clock_gettime(CLOCK_MONOTONIC, &t0);
            const size_t total = ((8 * vol + 4096 + chunk - 1) / chunk) * chunk;
            void* va = nullptr;
            if (hipMemAddressReserve(&va, total, chunk, nullptr, 0) != hipSuccess) { printf("reserve failed\n"); return 1; }
            for (size_t off = 0; off < total; off += chunk) {
                hipMemGenericAllocationHandle_t h;
                if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) { printf("create failed at %zu\n", off); return 1; }
                if (hipMemMap((char*)va + off, chunk, 0, h, 0) != hipSuccess) { printf("map failed\n"); return 1; }
                (void)hipMemRelease(h);
            }
            hipMemAccessDesc acc = {};
            acc.location = prop.location;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            if (hipMemSetAccess(va, total, &acc, 1) != hipSuccess) { printf("access failed\n"); return 1; }
            clock_gettime(CLOCK_MONOTONIC, &t1);
            printf("  alloc+map %.1f ms; ", (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) / 1e6);
            (void)hipMemset(va, 1, total);
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                (void)hipEventRecord(a);
                read8<0><<<65536, 256>>>((const uint8_t*)va, npix, vol, out);
                (void)hipEventRecord(b); (void)hipEventSynchronize(b);
                float ms; (void)hipEventElapsedTime(&ms, a, b);
                if (rep) best = ms < best ? ms : best;
            }
            printf("vmm buffer %d at %p: %.3f ms\n", i, va, best);
        }
        return 0;
    }
    // placement probe: the same kernel on several separately allocated buffers of this process
    if (argc > 1) {
        const int nb = atoi(argv[1]);
        uint8_t* bufs[16];
        for (int i = 0; i < nb && i < 16; ++i) { (void)hipMalloc(&bufs[i], 8 * vol + 4096); (void)hipMemset(bufs[i], 1, 8 * vol + 4096); }
        for (int round = 0; round < 2; ++round)
        for (int i = 0; i < nb && i < 16; ++i) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                (void)hipEventRecord(a);
                read8<0><<<65536, 256>>>(bufs[i], npix, vol, out);
                (void)hipEventRecord(b); (void)hipEventSynchronize(b);
                float ms; (void)hipEventElapsedTime(&ms, a, b);
                if (rep) best = ms < best ? ms : best;
            }
            printf("round %d buffer %d at %p: %.3f ms\n", round, i, (void*)bufs[i], best);
        }
        return 0;
    }
    for (int layout = 0; layout < 2; ++layout) {
        float best = 1e9f, worst = 0.f;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(a);
            if (layout == 0) read8<0><<<65536, 256>>>(buf, npix, vol, out); else read8<1><<<65536, 256>>>(buf, npix, vol, out);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep) { best = ms < best ? ms : best; worst = ms > worst ? ms : worst; }
        }
        printf("layout %s: %.3f - %.3f ms  (%.2f TB/s best)\n", layout ? "B block-interleaved" : "A eight volumes", best, worst, 8.0 * vol / best / 1e9);
    }
    return 0;
}
