// Premise check for load balancing (round 4): if every workgroup got a share of the tiles in proportion to the speed it
// showed in the previous launch — no atomics, just a per-workgroup tile count — does the launch end when the AVERAGE workgroup
// would, i.e. is the ~4 % tail of tools/probe_epilogue.hip recoverable? Launch k records each workgroup's scan time; launch
// k + 1 deals the first `base` rounds round-robin as always and hands out the rest of the tiles (the tail region) by weight.
// Build: hipcc -O3 --offload-arch=gfx950 -Iminarrow_amd/csrc -Iinclude tools/probe_weighted.hip -o /tmp/probe_weighted
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>

#include "ma_acc.hpp"
#include "ma_device.hpp"

using namespace ma;

#define HIP(x)                                                      \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            exit(1);                                                \
        }                                                           \
    } while (0)

constexpr int UNROLL = 8, PACE = 24;

struct Args {
    const int64_t* data;
    size_t n_static;          // tiles dealt round-robin (multiple of the grid)
    const unsigned* extra0;   // per workgroup: first extra tile (index into the tail region) ...
    const unsigned* extra_n;  // ... and how many
    uint64_t* sums;           // per workgroup partial (checked on the host)
    uint64_t* stamps;         // per workgroup: t0, t2
};

__global__ __launch_bounds__(kBlock) void sum_w(Args a) {
    typedef Vec16<int64_t>::type V;
    constexpr size_t WAVE_ROWS = (size_t)64 * 2 * UNROLL, TILE_ROWS = WAVE_ROWS * kWaves;
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t t0 = wall_clock64();
    IntAcc acc[2];
    acc[0].init();
    acc[1].init();
    auto run = [&](size_t t) {
        const V* p = (const V*)(a.data + t * TILE_ROWS + (size_t)wave * WAVE_ROWS) + lane;
        V v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            v[u] = load16<V, true>(p + (size_t)u * 64);
            if (u + 1 < UNROLL) pace_loads<PACE>();
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            acc[0].add(v[u][0]);
            acc[1].add(v[u][1]);
        }
    };
    for (size_t t = blockIdx.x; t < a.n_static; t += gridDim.x) run(t);
    const unsigned e0 = a.extra0[blockIdx.x], en = a.extra_n[blockIdx.x];
    for (unsigned k = 0; k < en; ++k) run(a.n_static + e0 + k);
    const uint64_t t2 = wall_clock64();
    acc[0].merge(acc[1]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc[0].shfl_down_merge(off);
    __shared__ uint64_t lds[kWaves];
    if (lane == 0) lds[wave] = acc[0].s;
    __syncthreads();
    if (tid == 0) {
        a.sums[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
        a.stamps[blockIdx.x * 2] = t0;
        a.stamps[blockIdx.x * 2 + 1] = t2;
    }
}

__global__ void fill_k(int64_t* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (int64_t)i;
}

int main() {
    const size_t top = 1000000000;
    int64_t* data;
    HIP(hipMalloc(&data, top * 8));
    hipLaunchKernelGGL(fill_k, dim3(2048), dim3(256), 0, 0, data, top);
    const int G = 256;
    unsigned *d_e0, *d_en;
    uint64_t *d_sums, *d_stamps;
    HIP(hipMalloc(&d_e0, G * 4));
    HIP(hipMalloc(&d_en, G * 4));
    HIP(hipMalloc(&d_sums, G * 8));
    HIP(hipMalloc(&d_stamps, G * 16));
    hipStream_t s;
    HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    HIP(hipEventCreate(&e0));
    HIP(hipEventCreate(&e1));
    HIP(hipDeviceSynchronize());
    const size_t tile_rows = (size_t)64 * 2 * UNROLL * kWaves;
    for (size_t rows : {(size_t)1 << 24, (size_t)1 << 26, (size_t)125000000, (size_t)1 << 28, (size_t)1000000000}) {
        const size_t n_tiles = rows / tile_rows;
        const uint64_t want = (uint64_t)(n_tiles * tile_rows) * (uint64_t)(n_tiles * tile_rows - 1) / 2;
        std::vector<double> w(G, 1.0);  // relative speeds, refined launch after launch
        for (int iter = 0; iter < 5; ++iter) {
            // shares: tiles_b = n_tiles * w_b / sum(w); `base` full round-robin rounds, the rest from the tail region
            const double wsum = std::accumulate(w.begin(), w.end(), 0.0);
            std::vector<size_t> share(G);
            size_t given = 0;
            for (int b = 0; b < G; ++b) given += share[b] = (size_t)((double)n_tiles * w[b] / wsum);
            for (int b = 0; given < n_tiles; b = (b + 1) % G, ++given) ++share[b];
            const size_t base = *std::min_element(share.begin(), share.end());
            std::vector<unsigned> x0(G), xn(G);
            unsigned cur = 0;
            for (int b = 0; b < G; ++b) {
                x0[b] = cur;
                xn[b] = (unsigned)(share[b] - base);
                cur += xn[b];
            }
            HIP(hipMemcpy(d_e0, x0.data(), G * 4, hipMemcpyHostToDevice));
            HIP(hipMemcpy(d_en, xn.data(), G * 4, hipMemcpyHostToDevice));
            Args a{data, base * G, d_e0, d_en, d_sums, d_stamps};
            const int reps = rows >= ((size_t)1 << 28) ? 20 : 60;
            for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(sum_w, dim3(G), dim3(kBlock), 0, s, a);
            float best = 1e9f;
            for (int trial = 0; trial < 3; ++trial) {
                HIP(hipEventRecord(e0, s));
                for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(sum_w, dim3(G), dim3(kBlock), 0, s, a);
                HIP(hipEventRecord(e1, s));
                HIP(hipEventSynchronize(e1));
                float ms;
                HIP(hipEventElapsedTime(&ms, e0, e1));
                best = std::min(best, ms / reps);
            }
            std::vector<uint64_t> sums(G), st(G * 2);
            HIP(hipMemcpy(sums.data(), d_sums, G * 8, hipMemcpyDeviceToHost));
            HIP(hipMemcpy(st.data(), d_stamps, G * 16, hipMemcpyDeviceToHost));
            uint64_t got = 0;
            for (uint64_t v : sums) got += v;
            std::vector<double> dur(G);
            for (int b = 0; b < G; ++b) dur[b] = (double)(st[b * 2 + 1] - st[b * 2]) * 0.01;
            const double dmin = *std::min_element(dur.begin(), dur.end()), dmax = *std::max_element(dur.begin(), dur.end());
            printf("{\"rows\": %zu, \"iter\": %d, \"us\": %.2f, \"tbps\": %.3f, \"scan_us_min\": %.2f, \"scan_us_max\": %.2f, \"tail_tiles\": %u, \"ok\": %s}\n", rows,
                   iter, best * 1e3, (double)(n_tiles * tile_rows * 8) / best / 1e9, dmin, dmax, cur, got == want ? "true" : "false");
            fflush(stdout);
            // speed of workgroup b in the last launch = its tiles / its scan time; damped update
            for (int b = 0; b < G; ++b) {
                const double speed = (double)share[b] / std::max(dur[b] - 1.5, 1.0);
                w[b] = iter == 0 ? speed : 0.5 * w[b] / (std::accumulate(w.begin(), w.end(), 0.0) / G) + 0.5 * speed / 1.0;
            }
            // normalise to mean 1
            const double m = std::accumulate(w.begin(), w.end(), 0.0) / G;
            for (double& x : w) x /= m;
        }
    }
    return 0;
}
