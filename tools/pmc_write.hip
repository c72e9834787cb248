// pmc_write.hip — what distinguishes a slow-writing block from a fast one in the hardware counters? N hipMalloc'd 8-GB
// blocks, the elementwise kernels' store pattern on each (3 dispatches per block, in block order), then a copy into each
// from block 0. Run under `rocprofv3 --kernel-trace --pmc <counters>`: dispatch i of write_k belongs to block i / 3.
//   hipcc -O3 --offload-arch=gfx950 tools/pmc_write.hip -o /tmp/pmc_write && /tmp/pmc_write [n=8]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
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

__global__ __launch_bounds__(BLOCK) void write_k(d2* __restrict__ out, size_t n_tiles) {
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const size_t v0 = t * TILE_VECS + wave * WAVE_VECS + lane;
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_nontemporal_store(d2{1.5, 2.5}, out + v0 + (size_t)u * 64);
    }
}
__global__ __launch_bounds__(BLOCK) void copy_k(const d2* __restrict__ a, d2* __restrict__ out, size_t n_tiles) {
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const size_t v0 = t * TILE_VECS + wave * WAVE_VECS + lane;
        d2 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = __builtin_nontemporal_load(a + v0 + (size_t)u * 64);
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_nontemporal_store(x[u], out + v0 + (size_t)u * 64);
    }
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 8;
    const size_t bytes = 8000000000ull;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<char*> b(n);
    for (auto& p : b) CK(hipMalloc(&p, bytes));
    const size_t n_tiles = bytes / 16 / TILE_VECS;
    for (int i = 0; i < n; ++i) {
        float ms[3];
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(write_k, dim3(cus * 6), dim3(BLOCK), 0, s, (d2*)b[i], n_tiles);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms[r], e0, e1));
        }
        printf("write block %d at %p: %.0f %.0f %.0f GB/s\n", i, (void*)b[i], bytes / ms[0] / 1e6, bytes / ms[1] / 1e6, bytes / ms[2] / 1e6);
    }
    for (int i = 1; i < n; ++i) {
        float ms[2];
        for (int r = 0; r < 2; ++r) {
            CK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(copy_k, dim3(cus * 6), dim3(BLOCK), 0, s, (const d2*)b[0], (d2*)b[i], n_tiles);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms[r], e0, e1));
        }
        printf("copy block 0 -> %d: %.0f %.0f GB/s\n", i, 2.0 * bytes / ms[0] / 1e6, 2.0 * bytes / ms[1] / 1e6);
    }
    return 0;
}
