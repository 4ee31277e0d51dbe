// Micro-benchmark: is the streaming rate of device memory a property of WHERE in the physical memory a block lies?
// Takes most of the device as physical chunks (hipMemCreate, in allocation order), maps each alone and times a read and a fill of it;
// then maps pairs / interleavings of chunks from different regions into one range and times those.
//   phys_map [chunk_mb=1024] [n_chunks=240] [interleave_mb=2]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(256) void read_kernel(const uint4* __restrict__ p, size_t n, uint32_t* __restrict__ sink) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t step = (size_t)gridDim.x * 256;
    uint32_t acc = 0;
    for (; i < n; i += step) {
        const uint4 v = p[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x9e3779b9u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void fill_kernel(uint4* __restrict__ p, size_t n, uint32_t v) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t step = (size_t)gridDim.x * 256;
    const uint4 w = make_uint4(v, v + 1u, v + 2u, v + 3u);
    for (; i < n; i += step) p[i] = w;
}

static hipEvent_t ea, eb;
static uint32_t* sink;
static void rates(void* buf, size_t bytes, float* rd, float* wr) {
    const size_t n = bytes / 16;
    float best[2] = {1e9f, 1e9f};
    for (int rep = 0; rep < 4; ++rep)
        for (int kind = 0; kind < 2; ++kind) {
            (void)hipEventRecord(ea);
            if (kind == 0) fill_kernel<<<16384, 256>>>((uint4*)buf, n, (uint32_t)rep);
            else read_kernel<<<16384, 256>>>((const uint4*)buf, n, sink);
            (void)hipEventRecord(eb);
            (void)hipEventSynchronize(eb);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, ea, eb);
            if (rep && ms < best[kind]) best[kind] = ms;
        }
    *wr = bytes / best[0] / 1e6f;
    *rd = bytes / best[1] / 1e6f;
}

int main(int argc, char** argv) {
    const size_t chunk = (size_t)(argc > 1 ? atoll(argv[1]) : 1024) << 20;
    int nchunks = argc > 2 ? atoi(argv[2]) : 240;
    (void)hipEventCreate(&ea);
    (void)hipEventCreate(&eb);
    (void)hipMalloc(&sink, 64);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    std::vector<hipMemGenericAllocationHandle_t> hs;
    for (int i = 0; i < nchunks; ++i) {
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
        hs.push_back(h);
    }
    nchunks = (int)hs.size();
    printf("%d chunks of %zu MB\n", nchunks, chunk >> 20);
    void* va = nullptr;
    if (hipMemAddressReserve(&va, chunk, (size_t)2 << 20, nullptr, 0) != hipSuccess) return 1;
    std::vector<float> rd(nchunks), wr(nchunks);
    for (int i = 0; i < nchunks; ++i) {
        if (hipMemMap(va, chunk, 0, hs[i], 0) != hipSuccess || hipMemSetAccess(va, chunk, &acc, 1) != hipSuccess) { printf("map %d failed\n", i); return 1; }
        rates(va, chunk, &rd[i], &wr[i]);
        (void)hipDeviceSynchronize();
        (void)hipMemUnmap(va, chunk);
    }
    // a second pass over the same chunks: is a chunk's rate its own (the two passes agree) or noise?
    std::vector<float> rd2(nchunks), wr2(nchunks);
    for (int i = 0; i < nchunks; ++i) {
        if (hipMemMap(va, chunk, 0, hs[i], 0) != hipSuccess || hipMemSetAccess(va, chunk, &acc, 1) != hipSuccess) { printf("map %d failed\n", i); return 1; }
        rates(va, chunk, &rd2[i], &wr2[i]);
        (void)hipDeviceSynchronize();
        (void)hipMemUnmap(va, chunk);
    }
    for (int i = 0; i < nchunks; ++i) printf("chunk %3d  read %5.0f %5.0f  fill %5.0f %5.0f GB/s\n", i, rd[i], rd2[i], wr[i], wr2[i]);
    {
        double mx = 0, my = 0, sxx = 0, syy = 0, sxy = 0;
        for (int i = 0; i < nchunks; ++i) { mx += rd[i]; my += rd2[i]; }
        mx /= nchunks; my /= nchunks;
        for (int i = 0; i < nchunks; ++i) { sxx += (rd[i] - mx) * (rd[i] - mx); syy += (rd2[i] - my) * (rd2[i] - my); sxy += (rd[i] - mx) * (rd2[i] - my); }
        printf("correlation of the two read passes over the chunks: %.3f\n", sxy / sqrt(sxx * syy + 1e-30));
    }
    // a range of 4 chunks: neighbours in allocation order against chunks taken a quarter of the device apart
    if (nchunks >= 16) {
        void* va4 = nullptr;
        if (hipMemAddressReserve(&va4, 4 * chunk, (size_t)2 << 20, nullptr, 0) != hipSuccess) return 1;
        auto four = [&](const char* what, int a, int b, int c, int d) {
            const int ids[4] = {a, b, c, d};
            for (int k = 0; k < 4; ++k)
                if (hipMemMap((char*)va4 + k * chunk, chunk, 0, hs[ids[k]], 0) != hipSuccess) { printf("map4 failed\n"); exit(1); }
            if (hipMemSetAccess(va4, 4 * chunk, &acc, 1) != hipSuccess) { printf("access4 failed\n"); exit(1); }
            float r, w;
            rates(va4, 4 * chunk, &r, &w);
            (void)hipDeviceSynchronize();
            (void)hipMemUnmap(va4, 4 * chunk);
            printf("%-28s chunks %3d %3d %3d %3d  read %5.0f  fill %5.0f GB/s\n", what, a, b, c, d, r, w);
        };
        const int q = nchunks / 4;
        for (int s = 0; s + 3 < nchunks; s += nchunks / 8) four("neighbours", s, s + 1, s + 2, s + 3);
        for (int s = 0; s < q; s += q / 4 > 0 ? q / 4 : 1) four("a quarter apart", s, s + q, s + 2 * q, s + 3 * q);
    }
    return 0;
}
