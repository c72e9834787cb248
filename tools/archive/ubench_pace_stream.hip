// ubench_pace_stream.hip — load / store pacing for the read+write streams (a + b: 2 reads 1 write; a * s: 1 read
// 1 write), the shapes of the elementwise kernels (8 accesses per operand per wave, 16 bytes per lane, non-temporal).
// PL = idle cycles between consecutive loads, PS = between consecutive stores. Companion of tools/ubench_pace.hip.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/ubench_pace_stream.hip -o /tmp/ups && /tmp/ups
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(x)                                                      \
    do {                                                           \
        hipError_t e = (x);                                        \
        if (e != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); \
            exit(1);                                               \
        }                                                          \
    } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

template <int CYCLES>
__device__ __forceinline__ void pace() {
    if constexpr (CYCLES > 16) asm volatile("s_nop 15" ::: "memory");
    if constexpr (CYCLES % 16 != 0) asm volatile("s_nop %0" ::"n"(CYCLES % 16 - 1) : "memory");
    if constexpr (CYCLES == 16 || CYCLES == 32) asm volatile("s_nop 15" ::: "memory");
}

template <int MODE, int U, int PL, int PS>
__global__ __launch_bounds__(256) void stream_kernel(const d2* __restrict__ a, const d2* __restrict__ b, d2* __restrict__ out,
                                                     size_t n_tiles, double s) {
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr size_t WAVE_VECS = (size_t)64 * U, TILE_VECS = WAVE_VECS * 4;
    for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const size_t v0 = t * TILE_VECS + wave * WAVE_VECS + lane;
        d2 x[U], y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            x[u] = __builtin_nontemporal_load(a + v0 + (size_t)u * 64);
            if (PL) pace<PL>();
        }
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                y[u] = __builtin_nontemporal_load(b + v0 + (size_t)u * 64);
                if (PL && u + 1 < U) pace<PL>();
            }
        }
        if (PL) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            d2 r = MODE == 0 ? x[u] + y[u] : x[u] * s;
            __builtin_nontemporal_store(r, out + v0 + (size_t)u * 64);
            if (PS && u + 1 < U) pace<PS>();
        }
    }
}

struct Variant {
    std::string name;
    int bytes_per_row;
    void (*launch)(const d2*, const d2*, d2*, size_t, int, hipStream_t);
    int grid;
    double best = 1e30;
};

template <int MODE, int U, int PL, int PS>
static void launch(const d2* a, const d2* b, d2* out, size_t rows, int grid, hipStream_t s) {
    size_t n_tiles = rows / ((size_t)2 * 64 * U * 4);
    hipLaunchKernelGGL((stream_kernel<MODE, U, PL, PS>), dim3(grid), dim3(256), 0, s, a, b, out, n_tiles, 2.5);
}

#define ADD(MODE, U, PL, PS)                                                                                              \
    for (int bpc : {2, 6})                                                                                                \
        vars.push_back({std::string(MODE == 0 ? "a+b " : "a*s ") + "U" #U " PL=" #PL " PS=" #PS " bpc=" + std::to_string(bpc), \
                        MODE == 0 ? 24 : 16, launch<MODE, U, PL, PS>, cus * bpc});


// ORDER 1: loads of a and b alternate (a0 b0 a1 b1 ...). ORDER 2: software pipelined — the next tile's loads are issued
// before the current tile's stores.
template <int U, int ORDER>
__global__ __launch_bounds__(256) void add_kernel(const d2* __restrict__ a, const d2* __restrict__ b, d2* __restrict__ out,
                                                  size_t n_tiles) {
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr size_t WAVE_VECS = (size_t)64 * U, TILE_VECS = WAVE_VECS * 4;
    if (ORDER == 1) {
        for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
            const size_t v0 = t * TILE_VECS + wave * WAVE_VECS + lane;
            d2 x[U], y[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                x[u] = __builtin_nontemporal_load(a + v0 + (size_t)u * 64);
                y[u] = __builtin_nontemporal_load(b + v0 + (size_t)u * 64);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) __builtin_nontemporal_store(x[u] + y[u], out + v0 + (size_t)u * 64);
        }
    } else {
        size_t t = blockIdx.x;
        d2 x[U], y[U], r[U];
        if (t < n_tiles) {
            const size_t v0 = t * TILE_VECS + wave * WAVE_VECS + lane;
#pragma unroll
            for (int u = 0; u < U; ++u) x[u] = __builtin_nontemporal_load(a + v0 + (size_t)u * 64);
#pragma unroll
            for (int u = 0; u < U; ++u) y[u] = __builtin_nontemporal_load(b + v0 + (size_t)u * 64);
        }
        while (t < n_tiles) {
            const size_t v0 = t * TILE_VECS + wave * WAVE_VECS + lane;
#pragma unroll
            for (int u = 0; u < U; ++u) r[u] = x[u] + y[u];
            const size_t tn = t + gridDim.x;
            if (tn < n_tiles) {
                const size_t v1 = tn * TILE_VECS + wave * WAVE_VECS + lane;
#pragma unroll
                for (int u = 0; u < U; ++u) x[u] = __builtin_nontemporal_load(a + v1 + (size_t)u * 64);
#pragma unroll
                for (int u = 0; u < U; ++u) y[u] = __builtin_nontemporal_load(b + v1 + (size_t)u * 64);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) __builtin_nontemporal_store(r[u], out + v0 + (size_t)u * 64);
            t = tn;
        }
    }
}

template <int U, int ORDER>
static void launch_add(const d2* a, const d2* b, d2* out, size_t rows, int grid, hipStream_t s) {
    size_t n_tiles = rows / ((size_t)2 * 64 * U * 4);
    hipLaunchKernelGGL((add_kernel<U, ORDER>), dim3(grid), dim3(256), 0, s, a, b, out, n_tiles);
}

#define ADDO(U, ORDER)                                                                                          \
    for (int bpc : {2, 4, 6})                                                                                   \
        vars.push_back({std::string("a+b U" #U " order=" #ORDER " bpc=") + std::to_string(bpc), 24, launch_add<U, ORDER>, cus * bpc});

int main(int argc, char** argv) {
    size_t rows = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1000000000ull;
    int rounds = argc > 2 ? atoi(argv[2]) : 3, reps = 5;
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
    ADD(0, 8, 0, 0) ADD(0, 8, 8, 0) ADD(0, 8, 16, 0) ADD(0, 8, 24, 0) ADD(0, 8, 0, 8) ADD(0, 8, 0, 16) ADD(0, 8, 8, 8) ADD(0, 8, 16, 16) ADD(0, 8, 16, 8)
    ADD(1, 8, 0, 0) ADD(1, 8, 8, 0) ADD(1, 8, 16, 0) ADD(1, 8, 24, 0) ADD(1, 8, 0, 8) ADD(1, 8, 0, 16) ADD(1, 8, 8, 8) ADD(1, 8, 16, 16) ADD(1, 8, 16, 8)
    ADDO(8, 1) ADDO(8, 2) ADDO(4, 2)
    ADD(0, 4, 0, 0) ADD(0, 4, 16, 0) ADD(0, 4, 16, 16) ADD(1, 4, 0, 0) ADD(1, 4, 16, 0) ADD(1, 4, 16, 16)
    for (int r = 0; r < rounds; ++r) {
        for (auto& v : vars) {
            v.launch(a, b, out, rows, v.grid, s);
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < reps; ++i) v.launch(a, b, out, rows, v.grid, s);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            v.best = std::min<double>(v.best, ms / reps);
        }
    }
    std::sort(vars.begin(), vars.end(), [](const Variant& x, const Variant& y) {
        return x.bytes_per_row != y.bytes_per_row ? x.bytes_per_row > y.bytes_per_row : x.best < y.best;
    });
    for (auto& v : vars)
        printf("%-36s %8.4f ms  %8.1f GB/s  %5.1f%% of 8 TB/s\n", v.name.c_str(), v.best, rows * (double)v.bytes_per_row / v.best / 1e6,
               rows * (double)v.bytes_per_row / v.best / 1e6 / 80.0);
    return 0;
}
