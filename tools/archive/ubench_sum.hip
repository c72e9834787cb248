// ubench_sum.hip — read-only stream experiments for the sum kernel (MI355X): single-buffered loop (what
// ma_reduce.hip does) vs a software-pipelined loop that issues the next tile's loads before consuming the current one.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_sum.hip -o /tmp/ubench_sum && /tmp/ubench_sum
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(x)                                                             \
    do {                                                                  \
        hipError_t e = (x);                                               \
        if (e != hipSuccess) {                                            \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));        \
            exit(1);                                                      \
        }                                                                 \
    } while (0)

typedef long long l2 __attribute__((ext_vector_type(2)));

struct Scratch {
    unsigned long long partials[4096 * 4];
    unsigned int ticket;
};
__device__ Scratch g_scratch;

template <int U, int BLOCK, bool PIPE, int PRIO, int MAP = 0>
__global__ __launch_bounds__(BLOCK) void sum_kernel(const l2* __restrict__ a, size_t n_tiles, long long* __restrict__ out) {
    constexpr int WAVES = BLOCK / 64;
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr size_t WAVE_VECS = (size_t)64 * U, TILE_VECS = WAVE_VECS * WAVES;
    if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
    l2 acc = {0, 0};
    if (MAP == 1) {
        // every wave instruction of the whole grid reads the next 1 KiB piece: piece = (iter * U + u) * n_waves + wave_id
        const size_t n_waves = (size_t)gridDim.x * WAVES, wave_id = (size_t)blockIdx.x * WAVES + wave;
        const size_t n_pieces = n_tiles * WAVES * U;  // 64-vector pieces
        for (size_t base = 0; base < n_pieces; base += n_waves * U) {
            l2 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                size_t piece = base + (size_t)u * n_waves + wave_id;
                v[u] = piece < n_pieces ? __builtin_nontemporal_load(a + piece * 64 + lane) : l2{0, 0};
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u];
        }
    } else if (MAP == 2) {
        // each workgroup owns one contiguous span of tiles
        size_t per = (n_tiles + gridDim.x - 1) / gridDim.x, t0 = (size_t)blockIdx.x * per;
        size_t t1 = t0 + per < n_tiles ? t0 + per : n_tiles;
        for (size_t t = t0; t < t1; ++t) {
            const l2* p = a + t * TILE_VECS + wave * WAVE_VECS + lane;
            l2 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(p + (size_t)u * 64);
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u];
        }
    } else if (MAP == 3) {
        // XCD-partitioned: workgroup b runs on XCD b % 8 (round-robin dispatch); XCD x owns the x-th contiguous eighth of
        // the tiles and its gridDim.x / 8 workgroups walk that eighth round-robin. (The default mapping, tile t ->
        // workgroup t % grid, interleaves the XCDs at tile granularity instead.)
        const size_t xcd = blockIdx.x & 7, local = blockIdx.x >> 3, per_xcd_blocks = gridDim.x >> 3;
        const size_t per = (n_tiles + 7) / 8, t0 = xcd * per, t1 = t0 + per < n_tiles ? t0 + per : n_tiles;
        for (size_t t = t0 + local; t < t1; t += per_xcd_blocks) {
            const l2* p = a + t * TILE_VECS + wave * WAVE_VECS + lane;
            l2 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(p + (size_t)u * 64);
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u];
        }
    } else if (!PIPE || MAP == 4) {
        for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
            const l2* p = a + t * TILE_VECS + wave * WAVE_VECS + lane;
            l2 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(p + (size_t)u * 64);
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u];
        }
    } else {
        size_t t = blockIdx.x;
        l2 v0[U], v1[U];
        if (t < n_tiles) {
            const l2* p = a + t * TILE_VECS + wave * WAVE_VECS + lane;
#pragma unroll
            for (int u = 0; u < U; ++u) v0[u] = __builtin_nontemporal_load(p + (size_t)u * 64);
        }
        while (t < n_tiles) {
            size_t tn = t + gridDim.x;
            if (tn < n_tiles) {
                const l2* p = a + tn * TILE_VECS + wave * WAVE_VECS + lane;
#pragma unroll
                for (int u = 0; u < U; ++u) v1[u] = __builtin_nontemporal_load(p + (size_t)u * 64);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v0[u];
            t = tn;
            size_t tn2 = t + gridDim.x;
            if (t < n_tiles) {
                if (tn2 < n_tiles) {
                    const l2* p = a + tn2 * TILE_VECS + wave * WAVE_VECS + lane;
#pragma unroll
                    for (int u = 0; u < U; ++u) v0[u] = __builtin_nontemporal_load(p + (size_t)u * 64);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) acc += v1[u];
                t = tn2;
            }
        }
    }
    long long s = acc.x + acc.y;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (MAP == 4) {
        // the library's epilogue (ma_reduce.hip): wave sums -> LDS -> one 32-byte partial per workgroup, agent-scope
        // release, ticket, and the workgroup that draws the last ticket folds every partial
        __shared__ long long lds[BLOCK / 64];
        __shared__ int is_last;
        if (lane == 0) lds[wave] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            long long w = 0;
            for (int i = 0; i < BLOCK / 64; ++i) w += lds[i];
            g_scratch.partials[blockIdx.x * 4] = (unsigned long long)w;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            unsigned int t = __hip_atomic_fetch_add(&g_scratch.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int last = t == gridDim.x - 1;
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            is_last = last;
        }
        __syncthreads();
        if (!is_last) return;
        long long tot = 0;
        for (unsigned i = threadIdx.x; i < gridDim.x; i += BLOCK)
            tot += (long long)__hip_atomic_load(&g_scratch.partials[i * 4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int o = 32; o > 0; o >>= 1) tot += __shfl_down(tot, o, 64);
        __syncthreads();
        if (lane == 0) lds[wave] = tot;
        __syncthreads();
        if (threadIdx.x == 0) {
            long long w = 0;
            for (int i = 0; i < BLOCK / 64; ++i) w += lds[i];
            *out = w;
            g_scratch.ticket = 0;
        }
        return;
    }
    if (lane == 0) atomicAdd((unsigned long long*)out, (unsigned long long)s);
}

__global__ void iota_kernel(long long* a, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = (long long)i;
}

struct Variant {
    std::string name;
    void (*launch)(const l2*, size_t, int, int, long long*, hipStream_t);
    int bpc;
    double best = 1e30;
    long long result = 0;
};

template <int U, int BLOCK, bool PIPE, int PRIO, int MAP = 0>
static void launch(const l2* a, size_t rows, int bpc, int cus, long long* out, hipStream_t s) {
    size_t tile_rows = (size_t)2 * 64 * U * (BLOCK / 64);
    size_t n_tiles = rows / tile_rows;
    int grid = (int)std::min<size_t>(n_tiles, (size_t)cus * bpc);
    hipLaunchKernelGGL((sum_kernel<U, BLOCK, PIPE, PRIO, MAP>), dim3(grid), dim3(BLOCK), 0, s, a, n_tiles, out);
}

#define ADDM(U, B, MAP)                                                                                     \
    for (int bpc : bpcs)                                                                                    \
        vars.push_back({std::string("U" #U " B" #B " map=" #MAP), launch<U, B, false, 0, MAP>, bpc});
#define ADD(U, B, PIPE, PRIO)                                                                               \
    for (int bpc : bpcs)                                                                                    \
        vars.push_back({std::string("U" #U " B" #B " pipe=" #PIPE " prio=" #PRIO), launch<U, B, PIPE, PRIO>, bpc});

int main(int argc, char** argv) {
    size_t rows = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1000000000ull;
    int rounds = argc > 2 ? atoi(argv[2]) : 4, reps = 10;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    l2* a;
    long long* out;
    CK(hipMalloc(&a, rows * 8));
    CK(hipMalloc(&out, 8));
    CK(hipMemset(a, 1, rows * 8));
    if (argc > 3 && std::string(argv[3]) == "iota") {  // the bench's data (v[i] = i) instead of a constant byte pattern
        hipLaunchKernelGGL(iota_kernel, dim3(4096), dim3(256), 0, 0, (long long*)a, rows);
        CK(hipDeviceSynchronize());
    }
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<Variant> vars;
    std::vector<int> bpcs = {1, 2};
    ADD(8, 256, false, 0) ADD(4, 256, false, 0) ADD(16, 256, false, 0)
    ADD(8, 256, true, 0) ADD(4, 256, true, 0) ADD(2, 256, true, 0) ADD(16, 256, true, 0)
    ADD(8, 256, false, 1) ADD(8, 256, true, 1)
    ADD(4, 512, false, 0) ADD(4, 512, true, 0) ADD(8, 512, false, 0) ADD(2, 1024, true, 0) ADD(4, 1024, false, 0)
    ADDM(8, 256, 1) ADDM(4, 256, 1) ADDM(16, 256, 1) ADDM(8, 256, 2) ADDM(4, 256, 2) ADDM(8, 512, 1) ADDM(8, 128, 1) ADDM(8, 256, 3) ADDM(4, 256, 3) ADDM(8, 256, 4)
    ADD(8, 128, false, 0) ADD(8, 128, true, 0) ADD(16, 128, false, 0) ADD(8, 64, true, 0) ADD(16, 64, false, 0) ADD(16, 64, true, 0)
    for (int r = 0; r < rounds; ++r) {
        for (auto& v : vars) {
            CK(hipMemsetAsync(out, 0, 8, s));
            v.launch(a, rows, v.bpc, cus, out, s);
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < reps; ++i) v.launch(a, rows, v.bpc, cus, out, s);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            v.best = std::min<double>(v.best, ms / reps);
        }
    }
    std::sort(vars.begin(), vars.end(), [](const Variant& x, const Variant& y) { return x.best < y.best; });
    for (auto& v : vars)
        printf("%-34s bpc=%d  %8.4f ms  %8.1f GB/s  %5.1f%% of 8 TB/s\n", v.name.c_str(), v.bpc, v.best, rows * 8.0 / v.best / 1e6,
               rows * 8.0 / v.best / 1e6 / 80.0);
    return 0;
}
