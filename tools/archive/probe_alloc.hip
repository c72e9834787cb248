// probe_alloc.hip — does the way a buffer was ALLOCATED decide the read+write rate?
// Round-2 finding (profiles/r02_ubench_rw_runtime_ab.txt): the same binary writes 6.5-6.8 TB/s on PyTorch's bundled
// HIP 7.0 runtime and 5.5-5.9 TB/s on /opt/rocm's 7.2 runtime, same box, same virtual addresses. This probe runs a
// write-only, a copy and a read-only kernel over buffers obtained in different ways, to find which allocation path
// carries the difference (and whether the library can pick the fast one for ma_dev_alloc).
//   hipcc -O3 --offload-arch=gfx950 tools/probe_alloc.hip -o /tmp/probe_alloc && /tmp/probe_alloc [rows]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(x)                                                             \
    do {                                                                  \
        hipError_t e = (x);                                               \
        if (e != hipSuccess) {                                            \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e)); \
            exit(1);                                                      \
        }                                                                 \
    } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int U = 8, BLOCK = 256, WAVES = 4;
constexpr size_t WAVE_VECS = 64 * U, TILE_VECS = WAVE_VECS * WAVES;

template <int MODE>  // 0 read, 1 write, 2 copy
__global__ __launch_bounds__(BLOCK) void k(const d2* __restrict__ a, d2* __restrict__ out, size_t n_tiles, double* sink) {
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    d2 acc = {0.0, 0.0};
    for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const size_t v0 = t * TILE_VECS + wave * WAVE_VECS + lane;
        d2 x[U];
        if (MODE != 1) {
#pragma unroll
            for (int u = 0; u < U; ++u) x[u] = __builtin_nontemporal_load(a + v0 + (size_t)u * 64);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (MODE == 0) acc += x[u];
            else __builtin_nontemporal_store(MODE == 1 ? d2{1.5, 2.5} : x[u], out + v0 + (size_t)u * 64);
        }
    }
    if (MODE == 0 && acc[0] + acc[1] == 123.456) *sink = acc[0];
}

struct Buf {
    std::string how;
    void* src = nullptr;
    void* dst = nullptr;
};

static bool vmm_alloc_aligned(size_t bytes, size_t va_align, size_t round_to, void** out) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t sz = ((bytes + round_to - 1) / round_to) * round_to;
    hipMemGenericAllocationHandle_t h;
    if (hipMemCreate(&h, sz, &prop, 0) != hipSuccess) return false;
    void* p = nullptr;
    if (hipMemAddressReserve(&p, sz, va_align, nullptr, 0) != hipSuccess) return false;
    if (hipMemMap(p, sz, 0, h, 0) != hipSuccess) return false;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(p, sz, &acc, 1) != hipSuccess) return false;
    *out = p;
    return true;
}

static bool vmm_alloc(size_t bytes, void** out) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess) return false;
    size_t sz = ((bytes + gran - 1) / gran) * gran;
    hipMemGenericAllocationHandle_t h;
    if (hipMemCreate(&h, sz, &prop, 0) != hipSuccess) return false;
    void* p = nullptr;
    if (hipMemAddressReserve(&p, sz, gran, nullptr, 0) != hipSuccess) return false;
    if (hipMemMap(p, sz, 0, h, 0) != hipSuccess) return false;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(p, sz, &acc, 1) != hipSuccess) return false;
    printf("  (VMM granularity %zu)\n", gran);
    *out = p;
    return true;
}

int main(int argc, char** argv) {
    const size_t rows = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1000000000ull;
    const size_t bytes = rows * 8;
    int rt = 0, drv = 0;
    CK(hipRuntimeGetVersion(&rt));
    CK(hipDriverGetVersion(&drv));
    printf("hipRuntimeGetVersion %d, hipDriverGetVersion %d, rows %zu\n", rt, drv, rows);
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    double* sink;
    CK(hipMalloc(&sink, 64));
    const size_t n_tiles = bytes / 16 / TILE_VECS;
    const int grid = (int)std::min<size_t>(n_tiles, (size_t)cus * 6);

    std::vector<Buf> bufs;
    auto add = [&](const char* how, auto alloc) {
        Buf b;
        b.how = how;
        if (alloc(&b.src) && alloc(&b.dst)) bufs.push_back(b);
        else {
            (void)hipGetLastError();
            printf("%-34s unavailable\n", how);
        }
    };
    if (argc > 2 && std::string(argv[2]) == "vmm") {
        // VA alignment study: hipMemAddressReserve alignment x physical size rounding, hipMalloc pairs in between
        const size_t MiB = (size_t)1 << 20, GiB = (size_t)1 << 30;
        for (int rep = 0; rep < 2; ++rep) {
            add("hipMalloc", [&](void** p) { return hipMalloc(p, bytes) == hipSuccess; });
            add("VMM va 2 MiB, size 2 MiB", [&](void** p) { return vmm_alloc_aligned(bytes, 2 * MiB, 2 * MiB, p); });
            add("VMM va 1 GiB, size 2 MiB", [&](void** p) { return vmm_alloc_aligned(bytes, GiB, 2 * MiB, p); });
            add("VMM va 1 GiB, size 1 GiB", [&](void** p) { return vmm_alloc_aligned(bytes, GiB, GiB, p); });
            add("VMM va 8 GiB, size 1 GiB", [&](void** p) { return vmm_alloc_aligned(bytes, 8 * GiB, GiB, p); });
        }
    } else if (argc > 2) {
        // size study: hipMalloc of the listed sizes (bytes), the kernels always touch the first `bytes` of each
        for (int i = 2; i < argc; ++i) {
            const size_t sz = strtoull(argv[i], nullptr, 10);
            char nm[64];
            snprintf(nm, sizeof(nm), "hipMalloc(%zu)", sz);
            if (sz < bytes) continue;
            add(nm, [&](void** p) { return hipMalloc(p, sz) == hipSuccess; });
        }
    } else {
    add("hipMalloc", [&](void** p) { return hipMalloc(p, bytes) == hipSuccess; });
    add("hipExtMallocWithFlags(Default)", [&](void** p) { return hipExtMallocWithFlags(p, bytes, hipDeviceMallocDefault) == hipSuccess; });
    add("hipExtMallocWithFlags(Uncached)", [&](void** p) { return hipExtMallocWithFlags(p, bytes, hipDeviceMallocUncached) == hipSuccess; });
    add("hipExtMallocWithFlags(FineGrained)", [&](void** p) { return hipExtMallocWithFlags(p, bytes, hipDeviceMallocFinegrained) == hipSuccess; });
    add("hipExtMallocWithFlags(Contiguous)", [&](void** p) { return hipExtMallocWithFlags(p, bytes, hipDeviceMallocContiguous) == hipSuccess; });
    add("hipMallocAsync (default pool)", [&](void** p) { return hipMallocAsync(p, bytes, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess; });
    add("VMM hipMemCreate + hipMemMap", [&](void** p) { return vmm_alloc(bytes, p); });
    add("hipMalloc of 2x, second half", [&](void** p) {
        void* q = nullptr;
        if (hipMalloc(&q, 2 * bytes + (1 << 21)) != hipSuccess) return false;
        *p = (char*)q + bytes + (1 << 21);
        return true;
    });
    }

    for (auto& b : bufs) {
        CK(hipMemsetAsync(b.src, 0x11, bytes, s));
        CK(hipStreamSynchronize(s));
        double best[3] = {1e30, 1e30, 1e30};
        for (int round = 0; round < 3; ++round)
            for (int mode = 0; mode < 3; ++mode) {
                auto launch = [&]() {
                    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(cus), dim3(BLOCK), 0, s, (const d2*)b.src, (d2*)b.dst, n_tiles, sink);
                    else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(BLOCK), 0, s, (const d2*)b.src, (d2*)b.dst, n_tiles, sink);
                    else hipLaunchKernelGGL(k<2>, dim3(grid), dim3(BLOCK), 0, s, (const d2*)b.src, (d2*)b.dst, n_tiles, sink);
                };
                launch();
                CK(hipEventRecord(e0, s));
                for (int i = 0; i < 5; ++i) launch();
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                best[mode] = std::min<double>(best[mode], ms / 5);
            }
        printf("%-34s read %7.1f  write %7.1f  copy %7.1f GB/s   (src %p dst %p)\n", b.how.c_str(), bytes / best[0] / 1e6,
               bytes / best[1] / 1e6, 2.0 * bytes / best[2] / 1e6, b.src, b.dst);
    }
    return 0;
}
