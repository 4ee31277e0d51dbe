// Micro-benchmark: what a streaming read / fill / copy gets from a volume-sized buffer (4096 x 4096 x 260 bytes) depending on HOW
// the buffer was allocated: hipMalloc (several held at once, then again after all were freed), physical chunks of a chosen size
// mapped into one virtual range through the VMM API (hipMemCreate / hipMemMap), hipExtMallocWithFlags.  DESIGN 4: the rate is a
// property of the buffer; this asks which allocation makes the fast kind on purpose.
//   alloc_probe [n_bufs=6]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <ctime>
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
// the marching kernels' pattern: every workgroup walks down the rows of its own 32-pixel column window (260 bytes per pixel,
// 8320 contiguous bytes per row and workgroup, one image row = 1 MB further)
__global__ __launch_bounds__(256) void march_kernel(const uint8_t* __restrict__ p, int H, int W, int Dp, uint32_t* __restrict__ sink) {
    const size_t row_bytes = (size_t)W * Dp;
    const size_t col0 = (size_t)blockIdx.x * 32 * Dp;
    uint32_t acc = 0;
    for (int r = 0; r < H; ++r) {
        const uint8_t* row = p + (size_t)r * row_bytes + col0;
        for (int o = threadIdx.x * 16; o + 16 <= 32 * Dp; o += 256 * 16) {
            uint4 v;
            __builtin_memcpy(&v, row + o, 16);
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x9e3779b9u) sink[0] = acc;
}

static hipEvent_t ea, eb;
static uint32_t* sink;

static void measure(const char* what, int idx, void* buf, size_t bytes) {
    const size_t n = bytes / 16;
    float best[3] = {1e9f, 1e9f, 1e9f};
    for (int rep = 0; rep < 4; ++rep) {
        for (int kind = 0; kind < 3; ++kind) {
            (void)hipEventRecord(ea);
            if (kind == 0) fill_kernel<<<16384, 256>>>((uint4*)buf, n, (uint32_t)rep);
            else if (kind == 1) read_kernel<<<16384, 256>>>((const uint4*)buf, n, sink);
            else march_kernel<<<128, 256>>>((const uint8_t*)buf, 4096, 4096, 260, sink);
            (void)hipEventRecord(eb);
            (void)hipEventSynchronize(eb);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, ea, eb);
            if (rep && ms < best[kind]) best[kind] = ms;
        }
    }
    printf("%-22s %2d  va %p (align 2^%d)  fill %.3f ms %.0f GB/s | read %.3f ms %.0f GB/s | march(128 wg) %.3f ms\n", what, idx, buf,
           __builtin_ctzll((unsigned long long)(uintptr_t)buf), best[0], bytes / best[0] / 1e6, best[1], bytes / best[1] / 1e6, best[2]);
    fflush(stdout);
}

struct vmm_buf { void* va; size_t total; };
static vmm_buf vmm_alloc(size_t bytes, size_t chunk, size_t va_align, double* ms_out) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    const size_t total = ((bytes + chunk - 1) / chunk) * chunk;
    void* va = nullptr;
    if (hipMemAddressReserve(&va, total, va_align, nullptr, 0) != hipSuccess) { printf("reserve failed\n"); exit(1); }
    for (size_t off = 0; off < total; off += chunk) {
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) { printf("create failed at %zu\n", off); exit(1); }
        if (hipMemMap((char*)va + off, chunk, 0, h, 0) != hipSuccess) { printf("map failed\n"); exit(1); }
        (void)hipMemRelease(h);
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(va, total, &acc, 1) != hipSuccess) { printf("access failed\n"); exit(1); }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    *ms_out = (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) / 1e6;
    return {va, total};
}
static void vmm_free(vmm_buf b) {
    (void)hipMemUnmap(b.va, b.total);
    (void)hipMemAddressFree(b.va, b.total);
}

int main(int argc, char** argv) {
    const int nb = argc > 1 ? atoi(argv[1]) : 6;
    const size_t bytes = (size_t)4096 * 4096 * 260;
    (void)hipEventCreate(&ea);
    (void)hipEventCreate(&eb);
    (void)hipMalloc(&sink, 64);
    {
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        size_t gmin = 0, grec = 0;
        (void)hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum);
        (void)hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended);
        printf("granularity min %zu recommended %zu\n", gmin, grec);
    }
    std::vector<void*> bufs;
    // 1. fresh process: hipMalloc, all held
    for (int i = 0; i < nb; ++i) { void* p = nullptr; if (hipMalloc(&p, bytes) != hipSuccess) break; bufs.push_back(p); }
    for (size_t i = 0; i < bufs.size(); ++i) measure("hipMalloc fresh", (int)i, bufs[i], bytes);
    // 2. VMM, chunks of 2 MB / 32 MB / 1 GB / one piece, while the hipMalloc'd ones are still held
    const size_t chunks[] = {(size_t)2 << 20, (size_t)32 << 20, (size_t)1 << 30, 0};
    for (size_t chunk : chunks) {
        for (int i = 0; i < 2; ++i) {
            const size_t c = chunk ? chunk : ((bytes + ((size_t)2 << 20) - 1) / ((size_t)2 << 20)) * ((size_t)2 << 20);
            double ms = 0;
            vmm_buf b = vmm_alloc(bytes, c, c < ((size_t)1 << 30) ? c : ((size_t)1 << 30), &ms);
            char what[64];
            snprintf(what, sizeof what, "vmm %zu MB (%.0f ms)", c >> 20, ms);
            measure(what, i, b.va, bytes);
            vmm_free(b);
        }
    }
    // 3. everything freed, then hipMalloc again
    for (void* p : bufs) (void)hipFree(p);
    bufs.clear();
    for (int i = 0; i < nb; ++i) { void* p = nullptr; if (hipMalloc(&p, bytes) != hipSuccess) break; bufs.push_back(p); }
    for (size_t i = 0; i < bufs.size(); ++i) measure("hipMalloc after free", (int)i, bufs[i], bytes);
    for (void* p : bufs) (void)hipFree(p);
    bufs.clear();
    // 4. one large arena held and returned, then hipMalloc
    { void* arena = nullptr; if (hipMalloc(&arena, (size_t)96 << 30) == hipSuccess) { (void)hipMemset(arena, 0, (size_t)1 << 30); (void)hipDeviceSynchronize(); (void)hipFree(arena); } else printf("arena failed\n"); }
    for (int i = 0; i < nb; ++i) { void* p = nullptr; if (hipMalloc(&p, bytes) != hipSuccess) break; bufs.push_back(p); }
    for (size_t i = 0; i < bufs.size(); ++i) measure("hipMalloc after arena", (int)i, bufs[i], bytes);
    // 5. sub-ranges of ONE large hipMalloc (an arena the library would carve its volumes from)
    for (void* p : bufs) (void)hipFree(p);
    bufs.clear();
    { char* arena = nullptr;
      if (hipMalloc((void**)&arena, bytes * nb + ((size_t)2 << 20)) == hipSuccess) {
          for (int i = 0; i < nb; ++i) measure("slice of one arena", i, arena + (size_t)i * bytes, bytes);
          (void)hipFree(arena);
      } }
    return 0;
}
