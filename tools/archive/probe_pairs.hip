// probe_pairs.hip — does a block's write-only rate (the probe of ma_dev_alloc_output) predict the rate of a COPY into it,
// and does the source block matter? N 8-GB blocks; for each destination: write-only rate (tile pattern, 6 workgroups per
// CU), read-only rate, copy rate from one fixed source, copy rate from the neighbouring block, copy in the other direction.
//   hipcc -O3 --offload-arch=gfx950 tools/probe_pairs.hip -o /tmp/probe_pairs && /tmp/probe_pairs [n=12]
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

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 12;
    const size_t bytes = 8000000000ull;
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
    std::vector<char*> b(n + 1);
    for (auto& p : b) CK(hipMalloc(&p, bytes));
    CK(hipMemset(b[0], 0x11, bytes));
    const size_t n_tiles = bytes / 16 / TILE_VECS;
    auto timed = [&](int mode, char* src, char* dst) {
        float best = 1e30f;
        for (int round = 0; round < 2; ++round) {
            auto launch = [&]() {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(cus), dim3(BLOCK), 0, s, (const d2*)src, (d2*)dst, n_tiles, sink);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(cus * 6), dim3(BLOCK), 0, s, (const d2*)src, (d2*)dst, n_tiles, sink);
                else hipLaunchKernelGGL(k<2>, dim3(cus * 6), dim3(BLOCK), 0, s, (const d2*)src, (d2*)dst, n_tiles, sink);
            };
            launch();
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < 3; ++i) launch();
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms / 3);
        }
        return (mode == 2 ? 2.0 : 1.0) * bytes / best / 1e6;
    };
    printf("block  address         write-only  read-only  copy<-b0  copy<-prev  copy b_i->b0   (GB/s)\n");
    for (int i = 1; i <= n; ++i) {
        const double w = timed(1, b[0], b[i]);
        const double r = timed(0, b[i], b[i]);
        const double c0 = timed(2, b[0], b[i]);
        const double cp = timed(2, b[i - 1], b[i]);
        const double cb = timed(2, b[i], b[0]);
        printf("%5d  %14p  %10.0f  %9.0f  %8.0f  %10.0f  %11.0f\n", i, (void*)b[i], w, r, c0, cp, cb);
        fflush(stdout);
    }
    return 0;
}
