// Static round-robin tiles leave the chip half idle for the last ~4 % of every scan: workgroups run at different speeds
// (tools/probe_epilogue.hip: scan-done stamps spread over 129 .. 139.5 us at 125 M rows, XCD medians 131.8 .. 135.7), and the
// launch ends with its slowest workgroup. This probe times the library's dense i64 sum shape against the same kernel with a
// DYNAMIC TAIL: the first `static_num/16` of the tiles dealt round-robin as before, the rest claimed by WAVES in pieces of
// `claim` x 8 KiB from 8 sharded counters (workgroup b starts on shard b mod 8 = its XCD and moves on when a shard runs dry),
// the next claim requested before the current piece's loads. Round 4.
// Build: hipcc -O3 --offload-arch=gfx950 -Iminarrow_amd/csrc -Iinclude tools/probe_dyntail.hip -o /tmp/probe_dyntail
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
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
constexpr unsigned kShards = 8, kShardWord0 = 64, kShardStride = 16, kShardFrom = 96;
constexpr unsigned kClaimWord0 = 256;  // 8 claim counters, 64 B apart, behind the ticket shards

struct Args {
    const int64_t* data;
    size_t n_tiles;        // full 32-KiB tiles
    size_t n_static;       // tiles dealt round-robin (a multiple of the grid)
    unsigned per_shard;    // dynamic wave-pieces (of `claim` x 8 KiB) per shard
    unsigned claim;        // 8-KiB wave runs per claim
    Partial* partials;
    unsigned* ticket;
    uint64_t* out;
};

template <bool DYN>
__global__ __launch_bounds__(kBlock) void sum_k(Args a) {
    typedef Vec16<int64_t>::type V;
    constexpr size_t WAVE_ROWS = (size_t)64 * 2 * UNROLL, TILE_ROWS = WAVE_ROWS * kWaves;
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    IntAcc acc[2];
    acc[0].init();
    acc[1].init();
    auto run = [&](const int64_t* base) {  // one wave run: 8 KiB
        const V* p = (const V*)base + lane;
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
    const size_t n_static = DYN ? a.n_static : a.n_tiles;
    for (size_t t = blockIdx.x; t < n_static; t += gridDim.x) run(a.data + t * TILE_ROWS + (size_t)wave * WAVE_ROWS);
    if constexpr (DYN) {
        // dynamic tail: shard s owns pieces [s * per_shard, (s + 1) * per_shard) of the region behind the static tiles
        const int64_t* region = a.data + a.n_static * TILE_ROWS;
        const size_t region_runs = (a.n_tiles - a.n_static) * kWaves;  // 8-KiB wave runs in the region
        unsigned shard = blockIdx.x & 7, tries = 0;
        unsigned* counters = a.ticket + kClaimWord0;
        unsigned idx = 0;
        if (lane == 0) idx = __hip_atomic_fetch_add(counters + shard * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        idx = __builtin_amdgcn_readfirstlane(idx);
        while (tries < 8) {
            if (idx < a.per_shard) {
                // request the next claim before this piece's loads
                unsigned nxt = 0;
                if (lane == 0) nxt = __hip_atomic_fetch_add(counters + shard * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const size_t first = ((size_t)shard * a.per_shard + idx) * a.claim;
                for (unsigned c = 0; c < a.claim; ++c)
                    if (first + c < region_runs) run(region + (first + c) * WAVE_ROWS);
                idx = __builtin_amdgcn_readfirstlane(nxt);
            } else {
                shard = (shard + 1) & 7;
                ++tries;
                if (tries < 8) {
                    unsigned nxt = 0;
                    if (lane == 0) nxt = __hip_atomic_fetch_add(counters + shard * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    idx = __builtin_amdgcn_readfirstlane(nxt);
                }
            }
        }
    }
    acc[0].merge(acc[1]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc[0].shfl_down_merge(off);
    __shared__ Partial lds[kWaves];
    __shared__ int is_last;
    if (lane == 0) acc[0].to_partial(lds[wave]);
    __syncthreads();
    if (tid == 0) {
        uint64_t s = lds[0].a + lds[1].a + lds[2].a + lds[3].a;
        uint64_t* q = (uint64_t*)&a.partials[blockIdx.x];
        store_agent(q, s);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int last;
        const unsigned sh = blockIdx.x & (kShards - 1);
        const unsigned members = (gridDim.x - sh + kShards - 1) / kShards;
        unsigned* shard = a.ticket + kShardWord0 + sh * kShardStride;
        last = 0;
        if (__hip_atomic_fetch_add(shard, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1) {
            __hip_atomic_store(shard, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kShards - 1;
        }
        is_last = last;
    }
    __syncthreads();
    if (!is_last) return;
    IntAcc tot;
    tot.init();
    for (unsigned i = tid; i < gridDim.x; i += kBlock) tot.s += load_agent((const uint64_t*)&a.partials[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tot.shfl_down_merge(off);
    __syncthreads();
    if (lane == 0) tot.to_partial(lds[wave]);
    __syncthreads();
    if (tid == 0) {
        *a.out = lds[0].a + lds[1].a + lds[2].a + lds[3].a;
        __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (DYN)
            for (int s = 0; s < 8; ++s) __hip_atomic_store(a.ticket + kClaimWord0 + s * 16, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
    Partial* partials;
    unsigned* ticket;
    uint64_t* out;
    HIP(hipMalloc(&partials, sizeof(Partial) * 16384));
    HIP(hipMalloc(&ticket, 4096));
    HIP(hipMemset(ticket, 0, 4096));
    HIP(hipMalloc(&out, 64));
    hipStream_t s;
    HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    HIP(hipEventCreate(&e0));
    HIP(hipEventCreate(&e1));
    HIP(hipDeviceSynchronize());
    const size_t tile_rows = (size_t)64 * 2 * UNROLL * kWaves;
    const size_t sizes[] = {(size_t)1 << 22, (size_t)1 << 24, (size_t)1 << 26, 125000000, (size_t)1 << 28, 1000000000};
    for (size_t rows : sizes) {
        const size_t n_tiles = rows / tile_rows;
        const uint64_t want = (uint64_t)(n_tiles * tile_rows) * (uint64_t)(n_tiles * tile_rows - 1) / 2;
        for (int grid : {256, 512, 768}) {
            if (rows > ((size_t)1 << 24) && grid != 256) continue;
            // static_num / 16 of the tiles static; 16 = all static (the baseline kernel)
            for (int static_num : {16, 15, 14, 12, 8, 0}) {
                for (unsigned claim : {1u, 2u, 4u}) {
                    if (static_num == 16 && claim != 1) continue;
                    Args a{};
                    a.data = data;
                    a.n_tiles = n_tiles;
                    a.n_static = (n_tiles * (size_t)static_num / 16) / (size_t)grid * (size_t)grid;
                    a.claim = claim;
                    const size_t region_runs = (n_tiles - a.n_static) * kWaves;
                    a.per_shard = (unsigned)((region_runs + (size_t)8 * claim - 1) / ((size_t)8 * claim));
                    a.partials = partials;
                    a.ticket = ticket;
                    a.out = out;
                    const int reps = rows >= ((size_t)1 << 28) ? 20 : 60;
                    auto launch = [&] {
                        if (static_num == 16) hipLaunchKernelGGL(sum_k<false>, dim3(grid), dim3(kBlock), 0, s, a);
                        else hipLaunchKernelGGL(sum_k<true>, dim3(grid), dim3(kBlock), 0, s, a);
                    };
                    for (int w = 0; w < 5; ++w) launch();
                    float best = 1e9f;
                    for (int trial = 0; trial < 3; ++trial) {
                        HIP(hipEventRecord(e0, s));
                        for (int r = 0; r < reps; ++r) launch();
                        HIP(hipEventRecord(e1, s));
                        HIP(hipEventSynchronize(e1));
                        float ms;
                        HIP(hipEventElapsedTime(&ms, e0, e1));
                        best = std::min(best, ms / reps);
                    }
                    uint64_t got;
                    HIP(hipMemcpy(&got, out, 8, hipMemcpyDeviceToHost));
                    printf("{\"rows\": %zu, \"grid\": %d, \"static_16ths\": %d, \"claim_runs\": %u, \"us\": %.2f, \"tbps\": %.3f, \"ok\": %s}\n", rows, grid,
                           static_num, claim, best * 1e3, (double)(n_tiles * tile_rows * 8) / best / 1e9, got == want ? "true" : "false");
                    fflush(stdout);
                }
            }
        }
    }
    return 0;
}
