// ubench_stream.hip — tuning microbenchmark for 2-read/1-write and 1-read/1-write f64 streams on MI355X.
// Explores the design space of the elementwise kernels (load/store cache policy, unroll, workgroups per CU,
// workgroup size, tile->workgroup mapping) in one process with interleaved rounds, and prints the practical
// ceiling (a 16-byte copy) next to them. Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/ubench_stream.hip -o /tmp/ubench_stream && /tmp/ubench_stream
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e = (x);                                                            \
        if (e != hipSuccess) {                                                         \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e));              \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

template <bool NT>
__device__ __forceinline__ d2 ld(const d2* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT>
__device__ __forceinline__ void st(d2* p, d2 v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// MODE 0: out = a + b   MODE 1: out = a * s   MODE 2: out = a (copy)   MODE 3: out = s (write only)   MODE 4: fma a*b+c
// MAP 0: tile t -> workgroup t % grid (round robin)   MAP 1: each workgroup owns a contiguous span of tiles
template <int MODE, int UNROLL, int BLOCK, bool NTL, bool NTS, int MAP>
__global__ __launch_bounds__(BLOCK) void stream_kernel(const d2* __restrict__ a, const d2* __restrict__ b,
                                                       d2* __restrict__ out, size_t n_tiles, double s) {
    constexpr int WAVES = BLOCK / 64;
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr size_t WAVE_VECS = (size_t)64 * UNROLL;
    constexpr size_t TILE_VECS = WAVE_VECS * WAVES;
    size_t t0, t1, step;
    if (MAP == 0) {
        t0 = blockIdx.x;
        t1 = n_tiles;
        step = gridDim.x;
    } else {
        size_t per = (n_tiles + gridDim.x - 1) / gridDim.x;
        t0 = (size_t)blockIdx.x * per;
        t1 = t0 + per < n_tiles ? t0 + per : n_tiles;
        step = 1;
    }
    for (size_t t = t0; t < t1; t += step) {
        const size_t v0 = t * TILE_VECS + wave * WAVE_VECS + lane;
        d2 x[UNROLL], y[UNROLL];
        if (MODE != 3) {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) x[u] = ld<NTL>(a + v0 + (size_t)u * 64);
        }
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) y[u] = ld<NTL>(b + v0 + (size_t)u * 64);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            d2 r;
            if (MODE == 0) r = x[u] + y[u];
            else if (MODE == 1) r = x[u] * s;
            else if (MODE == 3) r = d2{s, s};
            else r = x[u];
            st<NTS>(out + v0 + (size_t)u * 64, r);
        }
    }
}

struct Variant {
    std::string name;
    int mode;
    int bytes_per_row;
    void (*launch)(const d2*, const d2*, d2*, size_t rows, int bpc, int cus, hipStream_t);
    int bpc;
    double best_ms = 1e30;
};

template <int MODE, int UNROLL, int BLOCK, bool NTL, bool NTS, int MAP>
static void launch(const d2* a, const d2* b, d2* out, size_t rows, int bpc, int cus, hipStream_t s) {
    size_t tile_rows = (size_t)2 * 64 * UNROLL * (BLOCK / 64);
    size_t n_tiles = rows / tile_rows;
    int grid = (int)std::min<size_t>(n_tiles, (size_t)cus * bpc);
    hipLaunchKernelGGL((stream_kernel<MODE, UNROLL, BLOCK, NTL, NTS, MAP>), dim3(grid), dim3(BLOCK), 0, s, a, b, out,
                       n_tiles, 2.5);
}

#define ADD(MODE, U, B, NTL, NTS, MAP)                                                                      \
    for (int bpc : bpcs)                                                                                    \
        vars.push_back({std::string(MODE == 0 ? "add_aa" : MODE == 1 ? "mul_as" : MODE == 3 ? "write " : "copy  ") + " U" #U " B" #B \
                            " ntl" #NTL " nts" #NTS " map" #MAP,                                            \
                        MODE, MODE == 0 ? 24 : MODE == 3 ? 8 : 16, launch<MODE, U, B, NTL, NTS, MAP>, bpc});

int main(int argc, char** argv) {
    size_t rows = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1000000000ull;
    int rounds = argc > 2 ? atoi(argv[2]) : 3;
    int reps = 5;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    d2 *a, *b, *out;
    CK(hipMalloc(&a, rows * 8));
    CK(hipMalloc(&b, rows * 8));
    CK(hipMalloc(&out, rows * 8));
    CK(hipMemset(a, 0x11, rows * 8));
    CK(hipMemset(b, 0x22, rows * 8));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));

    std::vector<Variant> vars;
    std::vector<int> bpcs = {2, 6, 10, 16, 32};
    ADD(0, 8, 256, true, true, 0) ADD(0, 4, 256, true, true, 0) ADD(0, 16, 256, true, true, 0)
    ADD(0, 8, 256, true, false, 0) ADD(0, 8, 512, true, true, 0)
    ADD(1, 8, 256, true, true, 0) ADD(1, 4, 256, true, true, 0) ADD(1, 16, 256, true, true, 0)
    ADD(2, 8, 256, true, true, 0)
    ADD(3, 8, 256, true, true, 0) ADD(3, 4, 256, true, true, 0) ADD(3, 8, 256, true, false, 0) ADD(3, 16, 256, true, true, 0)
    for (int r = 0; r < rounds; ++r) {
        for (auto& v : vars) {
            v.launch(a, b, out, rows, v.bpc, cus, s);  // warm
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < reps; ++i) v.launch(a, b, out, rows, v.bpc, cus, s);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            v.best_ms = std::min<double>(v.best_ms, ms / reps);
        }
    }
    std::sort(vars.begin(), vars.end(), [](const Variant& x, const Variant& y) {
        return x.mode != y.mode ? x.mode < y.mode : x.best_ms < y.best_ms;
    });
    for (auto& v : vars)
        printf("%-40s bpc=%d  %8.4f ms  %8.1f GB/s  %7.1f Grows/s\n", v.name.c_str(), v.bpc, v.best_ms,
               rows * (double)v.bytes_per_row / v.best_ms / 1e6, rows / v.best_ms / 1e6);
    return 0;
}
