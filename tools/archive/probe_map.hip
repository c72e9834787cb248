// probe_map.hip — write / read rate of every 4-GiB window of (nearly) all of HBM: is the per-buffer write rate of
// profiles/r02_probe_alloc_*.txt a property of physical REGIONS? Allocates `n_big` buffers of `big_gib` GiB and measures
// a write-only and a read-only kernel on each 4-GiB window (the device's own address-to-channel interleave is the same
// for all of them; the windows differ only in where the driver put them).
//   hipcc -O3 --offload-arch=gfx950 tools/probe_map.hip -o /tmp/probe_map && /tmp/probe_map [n_big=8] [big_gib=32]
#include <hip/hip_runtime.h>

#include <algorithm>
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

template <int MODE>  // 0 read, 1 write
__global__ __launch_bounds__(BLOCK) void k(d2* __restrict__ p, size_t n_tiles, double* sink) {
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    d2 acc = {0.0, 0.0};
    for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const size_t v0 = t * TILE_VECS + wave * WAVE_VECS + lane;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (MODE == 0) acc += __builtin_nontemporal_load(p + v0 + (size_t)u * 64);
            else __builtin_nontemporal_store(d2{1.5, 2.5}, p + v0 + (size_t)u * 64);
        }
    }
    if (MODE == 0 && acc[0] + acc[1] == 123.456) *sink = acc[0];
}

int main(int argc, char** argv) {
    const int n_big = argc > 1 ? atoi(argv[1]) : 8;
    const size_t big = (size_t)(argc > 2 ? atoi(argv[2]) : 32) << 30;
    const size_t win = (size_t)4 << 30;
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
    std::vector<char*> bufs;
    for (int i = 0; i < n_big; ++i) {
        char* p = nullptr;
        if (hipMalloc(&p, big) != hipSuccess) {
            (void)hipGetLastError();
            break;
        }
        bufs.push_back(p);
    }
    printf("%zu buffers of %zu GiB\n", bufs.size(), big >> 30);
    const size_t n_tiles = win / 16 / TILE_VECS;
    const int grid = cus * 6;
    for (size_t b = 0; b < bufs.size(); ++b) {
        printf("buffer %zu at %p:", b, (void*)bufs[b]);
        for (size_t off = 0; off + win <= big; off += win) {
            d2* p = (d2*)(bufs[b] + off);
            float best_w = 1e30f, best_r = 1e30f;
            for (int round = 0; round < 2; ++round) {
                hipLaunchKernelGGL(k<1>, dim3(grid), dim3(BLOCK), 0, s, p, n_tiles, sink);
                CK(hipEventRecord(e0, s));
                for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(BLOCK), 0, s, p, n_tiles, sink);
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                best_w = std::min(best_w, ms / 3);
                CK(hipEventRecord(e0, s));
                for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<0>, dim3(cus), dim3(BLOCK), 0, s, p, n_tiles, sink);
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                best_r = std::min(best_r, ms / 3);
            }
            printf("  w%4.0f/r%4.0f", win / best_w / 1e6, win / best_r / 1e6);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
