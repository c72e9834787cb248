// ubench_pace.hip — does spacing out a wave's load instructions change the read rate? (MI355X, read-only stream)
// tools/ubench_sum.hip found its bounds-checked piece-interleaved loop (8 loads a dozen instructions apart, one
// s_waitcnt vmcnt(0), then all adds) FASTER than tight loops that issue their 8 loads back to back. This isolates the
// effect: the library's tile loop (tile t -> workgroup t mod grid, a wave owns U KiB) and the piece-interleaved loop,
// each with PACE = nothing / s_nop / s_sleep between consecutive loads, and with all-at-once vs progressive consumption.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_pace.hip -o /tmp/ubench_pace && /tmp/ubench_pace [rows] [rounds]
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

typedef long long l2 __attribute__((ext_vector_type(2)));

// PACE = idle cycles between two consecutive loads of a wave: s_nop n waits n + 1 cycles (n <= 15); 64+ use s_sleep.
template <int PACE>
__device__ __forceinline__ void pace() {
    if constexpr (PACE >= 64) {
        __builtin_amdgcn_s_sleep(PACE / 64);
    } else {
        if constexpr (PACE >= 16) asm volatile("s_nop 15");
        if constexpr (PACE >= 32) asm volatile("s_nop 15");
        if constexpr (PACE >= 48) asm volatile("s_nop 15");
        if constexpr (PACE % 16 != 0) asm volatile("s_nop %0" ::"n"(PACE % 16 - 1));
    }
}

// MAP 0: tile loop. MAP 1: piece-interleaved (full rounds only; rows are a multiple of the round here).
// DRAIN 1: one wait for all loads, then the adds; DRAIN 0: whatever the compiler schedules (progressive waits).
template <int U, int PACE, int MAP, int DRAIN>
__global__ __launch_bounds__(256) void sum_kernel(const l2* __restrict__ a, size_t n_tiles, long long* __restrict__ out) {
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr size_t WAVE_VECS = (size_t)64 * U, TILE_VECS = WAVE_VECS * 4;
    l2 acc = {0, 0};
    if (MAP == 0) {
        for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
            const l2* p = a + t * TILE_VECS + wave * WAVE_VECS + lane;
            l2 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                v[u] = __builtin_nontemporal_load(p + (size_t)u * 64);
                if (u + 1 < U) pace<PACE>();
            }
            if (DRAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u];
        }
    } else {
        const size_t n_waves = (size_t)gridDim.x * 4, wave_id = (size_t)blockIdx.x * 4 + wave;
        const size_t n_pieces = n_tiles * 4 * U, round = n_waves * U;
        const l2* base = a + lane;
        for (size_t k = 0; k + round <= n_pieces; k += round) {
            l2 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                v[u] = __builtin_nontemporal_load(base + (k + (size_t)u * n_waves + wave_id) * 64);
                if (u + 1 < U) pace<PACE>();
            }
            if (DRAIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u];
        }
    }
    long long s = acc.x + acc.y;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (lane == 0) atomicAdd((unsigned long long*)out, (unsigned long long)s);
}

struct Variant {
    std::string name;
    void (*launch)(const l2*, size_t, int, long long*, hipStream_t);
    int grid;
    double best = 1e30;
};

template <int U, int PACE, int MAP, int DRAIN>
static void launch(const l2* a, size_t rows, int grid, long long* out, hipStream_t s) {
    size_t n_tiles = rows / ((size_t)2 * 64 * U * 4);
    hipLaunchKernelGGL((sum_kernel<U, PACE, MAP, DRAIN>), dim3(grid), dim3(256), 0, s, a, n_tiles, out);
}

#define ADD(U, PACE, MAP, DRAIN)                                                                                          \
    for (int bpc : {1, 2})                                                                                                \
        vars.push_back({std::string("U" #U " pace=" #PACE " map=" #MAP " drain=" #DRAIN " bpc=") + std::to_string(bpc),  \
                        launch<U, PACE, MAP, DRAIN>, cus * bpc});

int main(int argc, char** argv) {
    // a multiple of every round used below (U <= 16, grids of 256 / 512 workgroups): 2^30 rows
    size_t rows = argc > 1 ? strtoull(argv[1], nullptr, 10) : ((size_t)1 << 30);
    int rounds = argc > 2 ? atoi(argv[2]) : 4, reps = 10;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    l2* a;
    long long* out;
    CK(hipMalloc(&a, rows * 8));
    CK(hipMalloc(&out, 8));
    CK(hipMemset(a, 1, rows * 8));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<Variant> vars;
    ADD(8, 0, 0, 0) ADD(8, 0, 0, 1)
    ADD(8, 4, 0, 0) ADD(8, 4, 0, 1) ADD(8, 8, 0, 0) ADD(8, 8, 0, 1) ADD(8, 12, 0, 0) ADD(8, 12, 0, 1) ADD(8, 16, 0, 0) ADD(8, 16, 0, 1)
    ADD(8, 20, 0, 0) ADD(8, 20, 0, 1) ADD(8, 24, 0, 0) ADD(8, 24, 0, 1) ADD(8, 32, 0, 1) ADD(8, 64, 0, 1)
    ADD(8, 0, 1, 0) ADD(8, 8, 1, 1) ADD(8, 12, 1, 1) ADD(8, 16, 1, 1) ADD(8, 20, 1, 1) ADD(8, 24, 1, 1)
    ADD(16, 0, 0, 0) ADD(16, 8, 0, 1) ADD(16, 12, 0, 1) ADD(16, 16, 0, 1) ADD(4, 16, 0, 1) ADD(4, 32, 0, 1)
    for (int r = 0; r < rounds; ++r) {
        for (auto& v : vars) {
            CK(hipMemsetAsync(out, 0, 8, s));
            v.launch(a, rows, v.grid, out, s);
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < reps; ++i) v.launch(a, rows, v.grid, out, s);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            v.best = std::min<double>(v.best, ms / reps);
        }
    }
    std::sort(vars.begin(), vars.end(), [](const Variant& x, const Variant& y) { return x.best < y.best; });
    for (auto& v : vars)
        printf("%-44s %8.4f ms  %8.1f GB/s  %5.1f%% of 8 TB/s\n", v.name.c_str(), v.best, rows * 8.0 / v.best / 1e6,
               rows * 8.0 / v.best / 1e6 / 80.0);
    return 0;
}
