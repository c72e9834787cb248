// probe_vmm3.hip — the follow-up of probe_vmm2: the same physical handles wrote fast in one virtual range and slowly in
// another. Is it the virtual range, the handles, or what else is mapped at the time? Two sets of eight 1-GiB handles, each
// mapped alone at base + k x 8 GiB (k = 0..7), then both at once in either order.
//   hipcc -O3 --offload-arch=gfx950 tools/probe_vmm3.hip -o /tmp/probe_vmm3 && /tmp/probe_vmm3
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                 \
    do {                                                                                      \
        hipError_t e = (x);                                                                   \
        if (e != hipSuccess) {                                                                \
            fprintf(stderr, "%s failed: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); \
            exit(1);                                                                          \
        }                                                                                     \
    } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int U = 8, BLOCK = 256, WAVES = 4;
constexpr size_t WAVE_VECS = 64 * U, TILE_VECS = WAVE_VECS * WAVES, TILE_BYTES = TILE_VECS * 16;

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

static hipStream_t s;
static hipEvent_t e0, e1;
static int cus;
static double* sink;

static double rate(int mode, const char* src, char* dst, size_t bytes, int reps) {
    const size_t n_tiles = bytes / TILE_BYTES;
    const size_t cap = (size_t)cus * (mode == 0 ? 1 : 6);
    const int grid = (int)std::min(n_tiles, cap);
    float best = 1e30f;
    for (int round = 0; round < 2; ++round) {
        auto launch = [&]() {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(BLOCK), 0, s, (const d2*)src, (d2*)dst, n_tiles, sink);
            else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(BLOCK), 0, s, (const d2*)src, (d2*)dst, n_tiles, sink);
            else hipLaunchKernelGGL(k<2>, dim3(grid), dim3(BLOCK), 0, s, (const d2*)src, (d2*)dst, n_tiles, sink);
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
    const size_t MiB = (size_t)1 << 20, GiB = (size_t)1 << 30;
    const size_t hbytes = GiB;
    const int per_block = 8, n_sets = argc > 1 ? atoi(argv[1]) : 3;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    cus = prop.multiProcessorCount;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipMalloc(&sink, 64));
    hipMemAllocationProp ap = {};
    ap.type = hipMemAllocationTypePinned;
    ap.location.type = hipMemLocationTypeDevice;
    ap.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = ap.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    void* reserved = nullptr;
    CK(hipMemAddressReserve(&reserved, 72 * GiB, 4 * GiB, nullptr, 0));
    char* base = (char*)(((uintptr_t)reserved + 4 * GiB - 1) & ~(uintptr_t)(4 * GiB - 1));
    printf("aligned base %p\n", (void*)base);
    std::vector<std::vector<hipMemGenericAllocationHandle_t>> sets(n_sets);
    for (auto& set : sets)
        for (int i = 0; i < per_block; ++i) {
            hipMemGenericAllocationHandle_t h;
            CK(hipMemCreate(&h, hbytes, &ap, 0));
            set.push_back(h);
        }
    const size_t bytes = 8000000000ull / TILE_BYTES * TILE_BYTES;
    auto map_set = [&](int which, int k) {
        char* p = base + (size_t)k * 8 * GiB;
        for (int j = 0; j < per_block; ++j) CK(hipMemMap(p + (size_t)j * hbytes, hbytes, 0, sets[which][j], 0));
        CK(hipMemSetAccess(p, per_block * hbytes, &acc, 1));
        return p;
    };
    auto unmap = [&](char* p) {
        CK(hipStreamSynchronize(s));
        CK(hipMemUnmap(p, per_block * hbytes));
    };
    printf("== each set of 8 handles mapped ALONE at base + k x 8 GiB: write rate (GB/s)\n set");
    for (int k = 0; k < 8; ++k) printf("   k=%d ", k);
    printf("  k=0 again\n");
    for (int which = 0; which < n_sets; ++which) {
        printf("%4d", which);
        for (int kk = 0; kk < 9; ++kk) {
            char* p = map_set(which, kk % 8);
            printf("  %5.0f", rate(1, nullptr, p, bytes, 3));
            unmap(p);
        }
        printf("\n");
        fflush(stdout);
    }
    printf("== per handle of each set, mapped alone at base: write rate (GB/s)\n");
    for (int which = 0; which < n_sets; ++which) {
        printf("%4d", which);
        for (int j = 0; j < per_block; ++j) {
            CK(hipMemMap(base, hbytes, 0, sets[which][j], 0));
            CK(hipMemSetAccess(base, hbytes, &acc, 1));
            printf("  %5.0f", rate(1, nullptr, base, hbytes, 4));
            CK(hipStreamSynchronize(s));
            CK(hipMemUnmap(base, hbytes));
        }
        printf("\n");
    }
    printf("== two sets mapped at once: (set, k) pairs, write rate of each, copy first -> second\n");
    const int combos[][4] = {{0, 0, 1, 2}, {1, 0, 0, 2}, {0, 2, 1, 0}, {0, 0, 1, 1}, {1, 4, 2, 6}, {2, 0, 1, 1}};
    for (auto& c : combos) {
        if (c[0] >= n_sets || c[2] >= n_sets) continue;
        char* p = map_set(c[0], c[1]);
        char* q = map_set(c[2], c[3]);
        printf("  set %d at k=%d: %5.0f   set %d at k=%d: %5.0f   copy: %5.0f   copy back: %5.0f\n", c[0], c[1], rate(1, nullptr, p, bytes, 3), c[2],
               c[3], rate(1, nullptr, q, bytes, 3), rate(2, p, q, bytes, 3), rate(2, q, p, bytes, 3));
        fflush(stdout);
        unmap(p);
        unmap(q);
    }
    return 0;
}
