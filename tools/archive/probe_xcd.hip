// Does the read rate of a streaming scan depend on WHICH XCD reads WHICH 32-KiB tile? (round 4)
//
// tools/probe_proc.c showed the 7.3 / 7.1 / 6.9 TB/s levels of the 10^9-row sum changing between contexts of ONE process in
// a repeatable pattern. The sum kernel deals tile t to workgroup t mod 256, and the dispatcher deals workgroups to the
// 8 XCDs round-robin from wherever its pointer stands: XCD = (workgroup + start) mod 8, so "tile mod 8 -> XCD" is fixed for a
// launch but its PHASE (start) is whatever earlier dispatches left behind. If memory is interleaved over the HBM stacks in
// pieces of about a tile, every XCD then streams from ONE stack phase, and how far that stack is from the XCD depends on the phase.
//
//   A  a 1-workgroup dummy kernel in front of each measurement (rotates the pointer by one), the read kernel records the XCD
//      of every workgroup: rate against (XCD of workgroup 0)
//   B  the same kernel on a base pointer shifted by j x 32 KiB
//   C  tiles rotated per round (workgroup b takes tile (b + i) mod grid of round i): every XCD sweeps all phases
//   D  XCD-aware: a workgroup reads its XCC_ID, takes a rank among its XCD's workgroups, and picks tiles whose phase is
//      (xcc + shift) mod 8 — independent of the dispatcher's pointer; shift = 0..7
// Build: hipcc -O3 --offload-arch=gfx950 tools/probe_xcd.hip -o /tmp/probe_xcd
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define HIP(x)                                                                     \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

typedef unsigned int u4 __attribute__((ext_vector_type(4)));
constexpr int kBlock = 256, kUnroll = 8;
constexpr size_t kTileVec = (size_t)kBlock * kUnroll;  // 2048 x 16 B = 32 KiB per workgroup tile

__device__ __forceinline__ unsigned xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 15u;
}

__global__ void dummy_k(unsigned* p) {
    if (threadIdx.x == 0 && p) p[0] = xcc_id();
}

// mode 0: tile = b + i*G.  mode 1: tile = i*G + (b + i) mod G.  mode 3: XCD-aware (see header); counters[8] zeroed per launch.
__global__ __launch_bounds__(kBlock) void read_k(const u4* __restrict__ base, size_t n_tiles, int mode, int shift, unsigned* xcc_out,
                                                  unsigned* counters, unsigned long long* sink) {
    const unsigned G = gridDim.x;
    unsigned b = blockIdx.x;
    __shared__ unsigned vb_s;
    const unsigned xcc = xcc_id();
    if (threadIdx.x == 0 && xcc_out) xcc_out[blockIdx.x] = xcc;
    if (mode == 3) {
        if (threadIdx.x == 0) {
            const unsigned r = atomicAdd(&counters[xcc & 7], 1u);  // rank of this workgroup among its XCD's
            vb_s = r * 8 + ((xcc + (unsigned)shift) & 7);          // virtual workgroup id: phase = (xcc + shift) mod 8
        }
        __syncthreads();
        b = vb_s;
    }
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long acc = 0;
    size_t i = 0;
    for (size_t t0 = 0; t0 < n_tiles; t0 += G, ++i) {
        size_t t = mode == 1 ? t0 + (b + i) % G : t0 + b;
        if (t >= n_tiles || b >= G) continue;
        const u4* p = base + t * kTileVec + (size_t)wave * 64 * kUnroll + lane;
        u4 v[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) v[u] = __builtin_nontemporal_load(p + (size_t)u * 64);
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) acc += (unsigned long long)v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 0x1234567887654321ull) sink[0] = acc;  // keeps the loads alive
}

int main(int argc, char** argv) {
    const size_t bytes = argc > 1 ? strtoull(argv[1], nullptr, 10) : 8000000000ull;
    int dev = 0;
    hipDeviceProp_t prop;
    HIP(hipGetDeviceProperties(&prop, dev));
    const int G = prop.multiProcessorCount;  // one workgroup per CU, as the library's dense scan
    char* buf = nullptr;
    HIP(hipMalloc(&buf, bytes + (1 << 20)));
    HIP(hipMemset(buf, 1, bytes + (1 << 20)));
    unsigned *xcc_out, *counters, *dummy_out;
    unsigned long long* sink;
    HIP(hipMalloc(&xcc_out, 4096 * 4));
    HIP(hipMalloc(&counters, 64));
    HIP(hipMalloc(&dummy_out, 64));
    HIP(hipMalloc(&sink, 64));
    hipStream_t s;
    HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    HIP(hipEventCreate(&e0));
    HIP(hipEventCreate(&e1));
    std::vector<unsigned> xh(4096);
    auto measure = [&](const char* base, int mode, int shift, int reps) {
        const size_t n_tiles = bytes / (kTileVec * 16);
        float best = 1e9f;
        for (int trial = 0; trial < 3; ++trial) {
            HIP(hipEventRecord(e0, s));
            for (int r = 0; r < reps; ++r) {
                if (mode == 3) HIP(hipMemsetAsync(counters, 0, 64, s));
                hipLaunchKernelGGL(read_k, dim3(G), dim3(kBlock), 0, s, (const u4*)base, n_tiles, mode, shift, xcc_out, counters, sink);
            }
            HIP(hipEventRecord(e1, s));
            HIP(hipEventSynchronize(e1));
            float ms;
            HIP(hipEventElapsedTime(&ms, e0, e1));
            if (ms / reps < best) best = ms / reps;
        }
        HIP(hipMemcpy(xh.data(), xcc_out, 4096 * 4, hipMemcpyDeviceToHost));
        return (double)(n_tiles * kTileVec * 16) / best / 1e9;  // TB/s = bytes / ms / 1e9
    };
    printf("{\"cus\": %d, \"bytes\": %zu, \"base\": \"%p\"}\n", G, bytes, (void*)buf);
    measure(buf, 0, 0, 5);
    // A: rotate the dispatcher's pointer by one workgroup per step
    for (int k = 0; k < 18; ++k) {
        hipLaunchKernelGGL(dummy_k, dim3(1), dim3(64), 0, s, dummy_out);
        const double r = measure(buf, 0, 0, 10);
        unsigned d;
        HIP(hipMemcpy(&d, dummy_out, 4, hipMemcpyDeviceToHost));
        printf("{\"test\": \"A_rotate\", \"k\": %d, \"tbps\": %.3f, \"xcc_of_wg\": [%u, %u, %u, %u, %u, %u, %u, %u, %u], \"dummy_xcc\": %u}\n", k, r, xh[0],
               xh[1], xh[2], xh[3], xh[4], xh[5], xh[6], xh[7], xh[8], d);
        fflush(stdout);
    }
    // B: base pointer shifted by j tiles
    for (int j = 0; j < 9; ++j) {
        const double r = measure(buf + (size_t)j * 32768, 0, 0, 10);
        printf("{\"test\": \"B_base_shift\", \"tiles\": %d, \"tbps\": %.3f, \"xcc_of_wg0\": %u}\n", j, r, xh[0]);
    }
    for (int j = 1; j < 8; ++j) {  // finer: 4-KiB steps
        const double r = measure(buf + (size_t)j * 4096, 0, 0, 10);
        printf("{\"test\": \"B_base_shift_4k\", \"x4KiB\": %d, \"tbps\": %.3f, \"xcc_of_wg0\": %u}\n", j, r, xh[0]);
    }
    // C: rotated tiles
    for (int k = 0; k < 4; ++k) {
        hipLaunchKernelGGL(dummy_k, dim3(1), dim3(64), 0, s, dummy_out);
        const double r = measure(buf, 1, 0, 10);
        printf("{\"test\": \"C_rotated_tiles\", \"k\": %d, \"tbps\": %.3f, \"xcc_of_wg0\": %u}\n", k, r, xh[0]);
    }
    // D: XCD-aware placement, every phase shift, under two positions of the dispatcher's pointer
    for (int k = 0; k < 2; ++k) {
        if (k) hipLaunchKernelGGL(dummy_k, dim3(3), dim3(64), 0, s, dummy_out);
        for (int shift = 0; shift < 8; ++shift) {
            const double r = measure(buf, 3, shift, 10);
            printf("{\"test\": \"D_xcc_aware\", \"pointer_moved\": %d, \"shift\": %d, \"tbps\": %.3f, \"xcc_of_wg0\": %u}\n", k, shift, r, xh[0]);
        }
    }
    return 0;
}
