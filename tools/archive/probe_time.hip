// probe_time.hip — is the write rate of a fresh process TIME dependent? One buffer pair, write-only / copy / read-only
// kernels measured every ~0.25 s for N seconds after process start (background VRAM clearing of memory a previous
// process released would show as a rate that climbs with time, independent of which buffer is measured).
//   hipcc -O3 --offload-arch=gfx950 tools/probe_time.hip -o /tmp/probe_time && /tmp/probe_time [seconds] [prealloc_gb]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

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

template <int MODE>
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

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 15.0;
    const size_t prealloc_gb = argc > 2 ? strtoull(argv[2], nullptr, 10) : 0;
    const auto t_start = std::chrono::steady_clock::now();
    auto now = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); };
    const size_t bytes = 8000000000ull;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    void* hold = nullptr;
    if (prealloc_gb) CK(hipMalloc(&hold, prealloc_gb << 30));  // push the measured pair further into VRAM
    char *a, *out;
    double* sink;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&out, bytes));
    CK(hipMalloc(&sink, 64));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CK(hipMemsetAsync(a, 0x11, bytes, s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const size_t n_tiles = bytes / 16 / TILE_VECS;
    const int grid = (int)std::min<size_t>(n_tiles, (size_t)cus * 6);
    size_t free_b = 0, total_b = 0;
    CK(hipMemGetInfo(&free_b, &total_b));
    printf("a %p out %p, prealloc %zu GiB, free %.1f of %.1f GB at t=%.2f s\n", (void*)a, (void*)out, prealloc_gb, free_b / 1e9,
           total_b / 1e9, now());
    while (now() < seconds) {
        double rate[3];
        for (int mode = 0; mode < 3; ++mode) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < 3; ++i) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(cus), dim3(BLOCK), 0, s, (const d2*)a, (d2*)out, n_tiles, sink);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(BLOCK), 0, s, (const d2*)a, (d2*)out, n_tiles, sink);
                else hipLaunchKernelGGL(k<2>, dim3(grid), dim3(BLOCK), 0, s, (const d2*)a, (d2*)out, n_tiles, sink);
            }
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            rate[mode] = (mode == 2 ? 2.0 : 1.0) * bytes / (ms / 3) / 1e6;
        }
        printf("t=%6.2f s  read %7.1f  write %7.1f  copy %7.1f GB/s\n", now(), rate[0], rate[1], rate[2]);
        fflush(stdout);
        std::this_thread::sleep_for(std::chrono::milliseconds(200));
    }
    return 0;
}
