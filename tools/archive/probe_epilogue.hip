// Where do the ~4.3 us that every sum launch costs beyond its bytes go? (round 4)
// The library's dense i64 sum kernel shape — 256 threads, 8 paced non-temporal 16-byte loads per lane per 32-KiB tile, tile t ->
// workgroup t mod grid, wave shuffles -> LDS -> one partial per workgroup published with sc1 stores, sharded tickets, the last
// arrival folds — with s_memrealtime stamps (100 MHz) taken by thread 0 of every workgroup:
//   t0 kernel entry | t1 first tile's data consumed | t2 scan loop done | t3 workgroup reduced (LDS) | t4 partial published
//   (vmcnt drained) | t5 ticket(s) answered | last workgroup only: t6 all partials loaded + folded | t7 final store issued
// Output: per size, the launch's wall time (events), and percentiles of the stamps relative to the earliest t0.
// Build: hipcc -O3 --offload-arch=gfx950 -Iminarrow_amd/csrc -Iinclude tools/probe_epilogue.hip -o /tmp/probe_epilogue
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

__device__ __forceinline__ uint64_t now() { return wall_clock64(); }

struct Args {
    const int64_t* data;
    size_t n_tiles;
    Partial* partials;
    unsigned* ticket;
    uint64_t* out;
    uint64_t* stamps;  // 8 per workgroup
    unsigned* hwid;    // per workgroup: HW_ID register (CU / SE) and XCC_ID
};

__global__ __launch_bounds__(kBlock) void sum_stamped(Args a) {
    typedef Vec16<int64_t>::type V;
    constexpr size_t WAVE_ROWS = (size_t)64 * 2 * UNROLL, TILE_ROWS = WAVE_ROWS * kWaves;
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint64_t t0 = now(), t1 = 0;
    IntAcc acc[2];
    acc[0].init();
    acc[1].init();
    for (size_t t = blockIdx.x; t < a.n_tiles; t += gridDim.x) {
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
        if (t1 == 0) {
            asm volatile("" ::"v"(acc[0].s));
            t1 = now();
        }
    }
    uint64_t t2 = now();
    acc[0].merge(acc[1]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc[0].shfl_down_merge(off);
    __shared__ Partial lds[kWaves];
    __shared__ int is_last;
    if (lane == 0) acc[0].to_partial(lds[wave]);
    __syncthreads();
    uint64_t t3 = 0, t4 = 0, t5 = 0;
    if (tid == 0) {
        uint64_t s = lds[0].a + lds[1].a + lds[2].a + lds[3].a;
        t3 = now();
        uint64_t* q = (uint64_t*)&a.partials[blockIdx.x];
        store_agent(q, s);
        store_agent(q + 1, 0);
        store_agent(q + 2, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        t4 = now();
        int last;
        if (gridDim.x <= kShardFrom) {
            last = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
        } else {
            const unsigned sh = blockIdx.x & (kShards - 1);
            const unsigned members = (gridDim.x - sh + kShards - 1) / kShards;
            unsigned* shard = a.ticket + kShardWord0 + sh * kShardStride;
            last = 0;
            if (__hip_atomic_fetch_add(shard, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1) {
                __hip_atomic_store(shard, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kShards - 1;
            }
        }
        t5 = now();
        is_last = last;
        uint64_t* st = a.stamps + (size_t)blockIdx.x * 8;
        st[0] = t0; st[1] = t1; st[2] = t2; st[3] = t3; st[4] = t4; st[5] = t5; st[6] = 0; st[7] = 0;
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        a.hwid[blockIdx.x * 2] = hw;
        a.hwid[blockIdx.x * 2 + 1] = xcc;
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
        uint64_t s = lds[0].a + lds[1].a + lds[2].a + lds[3].a;
        uint64_t t6 = now();
        *a.out = s;
        __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint64_t t7 = now();
        uint64_t* st = a.stamps + (size_t)blockIdx.x * 8;
        st[6] = t6;
        st[7] = t7;
    }
}

__global__ void fill_k(int64_t* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (int64_t)i;
}

int main() {
    const size_t top = (size_t)1 << 27;
    int64_t* data;
    HIP(hipMalloc(&data, top * 8));
    hipLaunchKernelGGL(fill_k, dim3(2048), dim3(256), 0, 0, data, top);
    Partial* partials;
    unsigned* ticket;
    uint64_t *out, *stamps;
    HIP(hipMalloc(&partials, sizeof(Partial) * 16384));
    HIP(hipMalloc(&ticket, 1024));
    HIP(hipMemset(ticket, 0, 1024));
    HIP(hipMalloc(&out, 64));
    HIP(hipMalloc(&stamps, 16384 * 64));
    unsigned* hwid;
    HIP(hipMalloc(&hwid, 16384 * 8));
    hipStream_t s;
    HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    HIP(hipEventCreate(&e0));
    HIP(hipEventCreate(&e1));
    HIP(hipDeviceSynchronize());
    const size_t tile_rows = (size_t)64 * 2 * UNROLL * kWaves;
    struct Case { size_t rows; int grid; };
    const Case cases[] = {{1 << 10, 1},        {(size_t)1 << 20, 256}, {(size_t)1 << 22, 768}, {(size_t)1 << 24, 768}, {(size_t)1 << 24, 512},
                          {(size_t)1 << 24, 256}, {(size_t)1 << 26, 256}, {125000000, 256},       {(size_t)1 << 27, 256}};
    for (const Case& c : cases) {
        Args a{data, c.rows / tile_rows, partials, ticket, out, stamps, hwid};
        const int grid = (int)std::min<size_t>((size_t)c.grid, std::max<size_t>(a.n_tiles, 1));
        for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(sum_stamped, dim3(grid), dim3(kBlock), 0, s, a);
        HIP(hipEventRecord(e0, s));
        const int reps = 50;
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(sum_stamped, dim3(grid), dim3(kBlock), 0, s, a);
        HIP(hipEventRecord(e1, s));
        HIP(hipEventSynchronize(e1));
        float ms;
        HIP(hipEventElapsedTime(&ms, e0, e1));
        std::vector<uint64_t> st((size_t)grid * 8);
        HIP(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));  // the LAST launch's stamps
        uint64_t base = ~0ull;
        for (int b = 0; b < grid; ++b) base = std::min(base, st[(size_t)b * 8]);
        auto col = [&](int k) {
            std::vector<double> v;
            for (int b = 0; b < grid; ++b)
                if (st[(size_t)b * 8 + k]) v.push_back((double)(st[(size_t)b * 8 + k] - base) * 0.01);  // 100 MHz -> us
            std::sort(v.begin(), v.end());
            return v;
        };
        printf("{\"rows\": %zu, \"grid\": %d, \"tiles\": %zu, \"period_us\": %.2f, \"ideal_us_at_7.3\": %.2f", c.rows, grid, a.n_tiles,
               ms * 1e3 / reps, (double)a.n_tiles * tile_rows * 8 / 7.3e6);
        const char* names[] = {"t0_entry", "t1_first_tile", "t2_scan_done", "t3_reduced", "t4_published", "t5_ticketed", "t6_folded", "t7_stored"};
        for (int k = 0; k < 8; ++k) {
            std::vector<double> v = col(k);
            if (v.empty()) continue;
            printf(", \"%s\": [%.2f, %.2f, %.2f]", names[k], v.front(), v[v.size() / 2], v.back());
        }
        // scan-done time by XCD (workgroup b runs on XCD b mod 8): median of t2 per XCD
        std::vector<unsigned> hw((size_t)grid * 2);
        HIP(hipMemcpy(hw.data(), hwid, hw.size() * 4, hipMemcpyDeviceToHost));
        printf(", \"t2_median_by_xcc\": [");
        for (unsigned x = 0; x < 8; ++x) {
            std::vector<double> v;
            for (int b = 0; b < grid; ++b)
                if ((hw[(size_t)b * 2 + 1] & 15u) == x) v.push_back((double)(st[(size_t)b * 8 + 2] - base) * 0.01);
            std::sort(v.begin(), v.end());
            printf("%s%.2f", x ? ", " : "", v.empty() ? 0.0 : v[v.size() / 2]);
        }
        printf("]}\n");
        if (c.rows == 125000000) {  // every workgroup of one launch: [xcc, se, cu, t1, t2]
            printf("{\"per_workgroup_125M\": [");
            for (int b = 0; b < grid; ++b) {
                const unsigned h = hw[(size_t)b * 2];
                printf("%s[%u, %u, %u, %.2f, %.2f]", b ? ", " : "", hw[(size_t)b * 2 + 1] & 15u, (h >> 13) & 7u, (h >> 8) & 15u,
                       (double)(st[(size_t)b * 8 + 1] - base) * 0.01, (double)(st[(size_t)b * 8 + 2] - base) * 0.01);
            }
            printf("]}\n");
        }
    }
    return 0;
}
