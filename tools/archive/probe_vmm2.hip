// probe_vmm2.hip — is the write rate of a block a property of its MAPPING (virtual address alignment against the
// physical chunk: page-table fragment size, TLB reach) rather than of the physical region? One physical handle mapped at
// virtual offsets 0 / 2 MiB / 32 MiB / 256 MiB / 512 MiB / 1 GiB from a 4-GiB-aligned base, measured each time; then 8-GB
// blocks built from handles at an aligned and at a misaligned base.
//   hipcc -O3 --offload-arch=gfx950 tools/probe_vmm2.hip -o /tmp/probe_vmm2 && /tmp/probe_vmm2 [handle_mib=1024] [n_handles=16]
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
    const size_t hbytes = (size_t)(argc > 1 ? atoi(argv[1]) : 1024) * MiB;
    const int n_handles = argc > 2 ? atoi(argv[2]) : 16;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    cus = prop.multiProcessorCount;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipMalloc(&sink, 64));
    char* src = nullptr;
    CK(hipMalloc(&src, 8 * GiB));
    CK(hipMemset(src, 0x11, 8 * GiB));
    printf("hipMalloc'd source at %p: read %5.0f write %5.0f GB/s\n", (void*)src, rate(0, src, src, 8 * GiB, 3), rate(1, nullptr, src, 8 * GiB, 3));

    hipMemAllocationProp ap = {};
    ap.type = hipMemAllocationTypePinned;
    ap.location.type = hipMemLocationTypeDevice;
    ap.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location = ap.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    void* reserved = nullptr;
    const size_t span = 40 * GiB;
    CK(hipMemAddressReserve(&reserved, span, 4 * GiB, nullptr, 0));
    char* base = (char*)(((uintptr_t)reserved + 4 * GiB - 1) & ~(uintptr_t)(4 * GiB - 1));
    printf("reserved %p (asked for 4-GiB alignment), aligned base %p, handles of %zu MiB\n", reserved, (void*)base, hbytes / MiB);

    std::vector<hipMemGenericAllocationHandle_t> handles;
    for (int i = 0; i < n_handles; ++i) {
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, hbytes, &ap, 0) != hipSuccess) {
            (void)hipGetLastError();
            break;
        }
        handles.push_back(h);
    }
    const size_t offs[] = {0, 2 * MiB, 32 * MiB, 256 * MiB, 512 * MiB, GiB, 2 * GiB};
    printf("== one handle mapped at base + offset: write rate (GB/s)\nhandle");
    for (size_t o : offs) printf("  +%4zuM", o / MiB);
    printf("   +0 again\n");
    const int n_single = std::min<int>((int)handles.size(), 10);
    for (int i = 0; i < n_single; ++i) {
        printf("%6d", i);
        for (int pass = 0; pass < 8; ++pass) {
            const size_t o = pass < 7 ? offs[pass] : 0;
            CK(hipMemMap(base + o, hbytes, 0, handles[i], 0));
            CK(hipMemSetAccess(base + o, hbytes, &acc, 1));
            printf("  %6.0f", rate(1, nullptr, base + o, hbytes, 4));
            CK(hipStreamSynchronize(s));
            CK(hipMemUnmap(base + o, hbytes));
        }
        printf("\n");
        fflush(stdout);
    }

    const size_t per_block = (8 * GiB) / hbytes;
    if (handles.size() >= per_block) {
        const size_t bytes = 8000000000ull / TILE_BYTES * TILE_BYTES;
        printf("== an 8-GiB block of %zu handles mapped back to back at base + offset: write / copy-into / read (GB/s)\n", per_block);
        for (size_t o : {(size_t)0, 2 * MiB, 64 * MiB, GiB, (size_t)0}) {
            for (size_t j = 0; j < per_block; ++j) CK(hipMemMap(base + o + j * hbytes, hbytes, 0, handles[j], 0));
            CK(hipMemSetAccess(base + o, per_block * hbytes, &acc, 1));
            const double w = rate(1, nullptr, base + o, bytes, 3);
            const double c = rate(2, src, base + o, bytes, 3);
            const double r = rate(0, base + o, base + o, bytes, 3);
            const double c2 = rate(2, base + o, src, bytes, 3);
            printf("  +%4zuM: write %5.0f  copy hipMalloc->block %5.0f  read %5.0f  copy block->hipMalloc %5.0f\n", o / MiB, w, c, r, c2);
            fflush(stdout);
            CK(hipStreamSynchronize(s));
            CK(hipMemUnmap(base + o, per_block * hbytes));
        }
        if (handles.size() >= 2 * per_block) {
            printf("== two such blocks (aligned): copy block A -> block B\n");
            char* A = base;
            char* B = base + 16 * GiB;
            for (size_t j = 0; j < per_block; ++j) {
                CK(hipMemMap(A + j * hbytes, hbytes, 0, handles[j], 0));
                CK(hipMemMap(B + j * hbytes, hbytes, 0, handles[per_block + j], 0));
            }
            CK(hipMemSetAccess(A, per_block * hbytes, &acc, 1));
            CK(hipMemSetAccess(B, per_block * hbytes, &acc, 1));
            printf("  write A %5.0f  write B %5.0f  copy A->B %5.0f  copy B->A %5.0f  copy hipMalloc->B %5.0f\n", rate(1, nullptr, A, bytes, 3),
                   rate(1, nullptr, B, bytes, 3), rate(2, A, B, bytes, 3), rate(2, B, A, bytes, 3), rate(2, src, B, bytes, 3));
        }
    }
    return 0;
}
