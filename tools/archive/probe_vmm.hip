// probe_vmm.hip — at what granularity is the write rate of HBM a property of the physical region, and can a block be
// BUILT from fast regions? (1) hipMalloc'd 16-GiB blocks probed per 256-MiB / 1-GiB / 4-GiB window; (2) physical handles
// of `handle_gib` GiB from hipMemCreate, each mapped and probed (whole and per 256 MiB); (3) the fastest handles mapped
// back to back into one virtual range: write rate and copy rate of that block against a block built from the slowest.
//   hipcc -O3 --offload-arch=gfx950 tools/probe_vmm.hip -o /tmp/probe_vmm && /tmp/probe_vmm [handle_gib=2] [n_handles=48]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e = (x);                                                            \
        if (e != hipSuccess) {                                                         \
            fprintf(stderr, "%s failed: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int U = 8, BLOCK = 256, WAVES = 4;
constexpr size_t WAVE_VECS = 64 * U, TILE_VECS = WAVE_VECS * WAVES, TILE_BYTES = TILE_VECS * 16;

template <int MODE>  // 1 write, 2 copy
__global__ __launch_bounds__(BLOCK) void k(const d2* __restrict__ a, d2* __restrict__ out, size_t n_tiles) {
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const size_t v0 = t * TILE_VECS + wave * WAVE_VECS + lane;
        d2 x[U];
        if (MODE == 2) {
#pragma unroll
            for (int u = 0; u < U; ++u) x[u] = __builtin_nontemporal_load(a + v0 + (size_t)u * 64);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_nontemporal_store(MODE == 1 ? d2{1.5, 2.5} : x[u], out + v0 + (size_t)u * 64);
    }
}

static hipStream_t s;
static hipEvent_t e0, e1;
static int cus;

static double rate(int mode, const char* src, char* dst, size_t bytes, int reps) {
    const size_t n_tiles = bytes / TILE_BYTES;
    const size_t cap = (size_t)cus * 6;
    const int grid = (int)std::min(n_tiles, cap);
    float best = 1e30f;
    for (int round = 0; round < 2; ++round) {
        auto launch = [&]() {
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(BLOCK), 0, s, (const d2*)src, (d2*)dst, n_tiles);
            else hipLaunchKernelGGL(k<2>, dim3(grid), dim3(BLOCK), 0, s, (const d2*)src, (d2*)dst, n_tiles);
        };
        launch();
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms / reps);
    }
    return (mode == 2 ? 2.0 : 1.0) * bytes / best / 1e6;
}

int main(int argc, char** argv) {
    const size_t handle_gib = argc > 1 ? atoi(argv[1]) : 2;
    const int n_handles = argc > 2 ? atoi(argv[2]) : 48;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    cus = prop.multiProcessorCount;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const size_t MiB = (size_t)1 << 20, GiB = (size_t)1 << 30;

    // (1) granularity inside plain hipMalloc blocks
    printf("== (1) hipMalloc 16-GiB blocks: write rate per window (GB/s)\n");
    std::vector<char*> plain;
    for (int b = 0; b < 3; ++b) {
        char* p = nullptr;
        CK(hipMalloc(&p, 16 * GiB));
        plain.push_back(p);
        printf("block %d at %p\n  4-GiB :", b, (void*)p);
        for (size_t o = 0; o < 16 * GiB; o += 4 * GiB) printf(" %5.0f", rate(1, nullptr, p + o, 4 * GiB, 2));
        printf("\n  1-GiB :");
        for (size_t o = 0; o < 16 * GiB; o += GiB) printf(" %5.0f", rate(1, nullptr, p + o, GiB, 3));
        printf("\n  256-MiB of the first 4 GiB:");
        for (size_t o = 0; o < 4 * GiB; o += 256 * MiB) printf(" %5.0f", rate(1, nullptr, p + o, 256 * MiB, 6));
        printf("\n");
        fflush(stdout);
    }

    // (2) physical handles
    hipMemAllocationProp ap = {};
    ap.type = hipMemAllocationTypePinned;
    ap.location.type = hipMemLocationTypeDevice;
    ap.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &ap, hipMemAllocationGranularityRecommended));
    const size_t hbytes = handle_gib * GiB;
    printf("== (2) %d physical handles of %zu GiB (granularity %zu KiB)\n", n_handles, handle_gib, gran >> 10);
    hipMemAccessDesc acc = {};
    acc.location = ap.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    std::vector<double> hrate;
    void* scratch_va = nullptr;
    CK(hipMemAddressReserve(&scratch_va, hbytes, 2 * GiB, nullptr, 0));
    for (int i = 0; i < n_handles; ++i) {
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, hbytes, &ap, 0) != hipSuccess) {
            (void)hipGetLastError();
            printf("hipMemCreate stopped at %d handles\n", i);
            break;
        }
        CK(hipMemMap(scratch_va, hbytes, 0, h, 0));
        CK(hipMemSetAccess(scratch_va, hbytes, &acc, 1));
        const double whole = rate(1, nullptr, (char*)scratch_va, hbytes, 3);
        printf("handle %2d: whole %5.0f | per 256 MiB:", i, whole);
        double lo = 1e9, hi = 0;
        for (size_t o = 0; o < hbytes; o += 256 * MiB) {
            const double r = rate(1, nullptr, (char*)scratch_va + o, 256 * MiB, 6);
            lo = std::min(lo, r);
            hi = std::max(hi, r);
            if (hbytes <= 2 * GiB) printf(" %5.0f", r);
        }
        printf("  [min %5.0f max %5.0f]\n", lo, hi);
        fflush(stdout);
        CK(hipStreamSynchronize(s));
        CK(hipMemUnmap(scratch_va, hbytes));
        handles.push_back(h);
        hrate.push_back(whole);
    }

    // (3) blocks built from chosen handles
    const size_t per_block = (8 * GiB + hbytes - 1) / hbytes;
    if (handles.size() >= 3 * per_block) {
        std::vector<int> order(handles.size());
        std::iota(order.begin(), order.end(), 0);
        std::sort(order.begin(), order.end(), [&](int x, int y) { return hrate[x] > hrate[y]; });
        auto build = [&](const std::vector<int>& which) {
            void* va = nullptr;
            CK(hipMemAddressReserve(&va, which.size() * hbytes, 2 * GiB, nullptr, 0));
            for (size_t j = 0; j < which.size(); ++j) CK(hipMemMap((char*)va + j * hbytes, hbytes, 0, handles[which[j]], 0));
            CK(hipMemSetAccess(va, which.size() * hbytes, &acc, 1));
            return (char*)va;
        };
        std::vector<int> fast(order.begin(), order.begin() + per_block);
        std::vector<int> fast2(order.begin() + per_block, order.begin() + 2 * per_block);
        std::vector<int> slow(order.end() - per_block, order.end());
        char* bf = build(fast);
        char* bf2 = build(fast2);
        char* bs = build(slow);
        const size_t bytes = 8000000000ull / TILE_BYTES * TILE_BYTES;
        printf("== (3) 8-GB blocks built from handles: fastest %zu (", per_block);
        for (int x : fast) printf("%d ", x);
        printf("), next %zu, slowest %zu\n", per_block, per_block);
        printf("write-only: built-fast %5.0f  built-next %5.0f  built-slow %5.0f  hipMalloc %5.0f\n", rate(1, nullptr, bf, bytes, 3),
               rate(1, nullptr, bf2, bytes, 3), rate(1, nullptr, bs, bytes, 3), rate(1, nullptr, plain[0], bytes, 3));
        printf("copy  hipMalloc -> built-fast %5.0f   hipMalloc -> built-slow %5.0f   hipMalloc -> hipMalloc %5.0f\n",
               rate(2, plain[1], bf, bytes, 3), rate(2, plain[1], bs, bytes, 3), rate(2, plain[1], plain[0], bytes, 3));
        printf("copy  built-slow -> built-fast %5.0f   built-next -> built-fast %5.0f   built-fast -> built-slow %5.0f\n",
               rate(2, bs, bf, bytes, 3), rate(2, bf2, bf, bytes, 3), rate(2, bf, bs, bytes, 3));
    }
    return 0;
}
