// ubench_rw2.hip — read+write kernels with a TIGHT store front (follow-up of ubench_w.hip: on most 8-GB buffers the
// write rate depends on how spread out the concurrently written addresses are; 1 MiB fronts from few waves write at
// 6.3-6.4 TB/s where the shipped 8-KiB-per-wave tiles from 6 workgroups per CU write at 5.5-5.7).
//   tile   shipped mapping: a wave owns 8 KiB per operand, tiles round-robin over workgroups
//   il     piece-interleaved: access u of round k of wave w touches piece (k*U + u) * n_waves + w (1 KiB pieces), for
//          loads and stores alike: every wave instruction of the grid lands next to its neighbours'
// copy (1R 1W), add (2R 1W), fma (3R 1W); several output buffers (slow and fast ones).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/ubench_rw2.hip -o /tmp/ubench_rw2 && /tmp/ubench_rw2 [n_out=4]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
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
constexpr int BLOCK = 256, WAVES = 4;

template <bool NT>
__device__ __forceinline__ void st(d2* p, d2 v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}
__device__ __forceinline__ d2 ld(const d2* p) { return __builtin_nontemporal_load(p); }

template <int MODE>  // 2 copy, 3 add, 4 fma
__device__ __forceinline__ d2 op(d2 x, d2 y, d2 z) {
    if (MODE == 2) return x;
    if (MODE == 3) return x + y;
    return d2{fma(x[0], y[0], z[0]), fma(x[1], y[1], z[1])};
}

template <int MODE, int U, bool NTS>
__global__ __launch_bounds__(BLOCK) void k_tile(const d2* __restrict__ a, const d2* __restrict__ b, const d2* __restrict__ c,
                                                d2* __restrict__ out, size_t n_vecs) {
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr size_t WAVE_VECS = 64 * U, TILE_VECS = WAVE_VECS * WAVES;
    const size_t n_tiles = n_vecs / TILE_VECS;
    for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const size_t v0 = t * TILE_VECS + wave * WAVE_VECS + lane;
        d2 x[U], y[U], z[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = ld(a + v0 + (size_t)u * 64);
        if (MODE >= 3) {
#pragma unroll
            for (int u = 0; u < U; ++u) y[u] = ld(b + v0 + (size_t)u * 64);
        }
        if (MODE == 4) {
#pragma unroll
            for (int u = 0; u < U; ++u) z[u] = ld(c + v0 + (size_t)u * 64);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) st<NTS>(out + v0 + (size_t)u * 64, op<MODE>(x[u], y[u], z[u]));
    }
}

template <int MODE, int U, bool NTS>
__global__ __launch_bounds__(BLOCK) void k_il(const d2* __restrict__ a, const d2* __restrict__ b, const d2* __restrict__ c,
                                              d2* __restrict__ out, size_t n_vecs) {
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t n_waves = (size_t)gridDim.x * WAVES, wave_id = (size_t)blockIdx.x * WAVES + wave;
    const size_t n_pieces = n_vecs / 64, round = n_waves * U;
    for (size_t k = 0; k + round <= n_pieces; k += round) {
        d2 x[U], y[U], z[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = ld(a + (k + (size_t)u * n_waves + wave_id) * 64 + lane);
        if (MODE >= 3) {
#pragma unroll
            for (int u = 0; u < U; ++u) y[u] = ld(b + (k + (size_t)u * n_waves + wave_id) * 64 + lane);
        }
        if (MODE == 4) {
#pragma unroll
            for (int u = 0; u < U; ++u) z[u] = ld(c + (k + (size_t)u * n_waves + wave_id) * 64 + lane);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) st<NTS>(out + (k + (size_t)u * n_waves + wave_id) * 64 + lane, op<MODE>(x[u], y[u], z[u]));
    }
}

struct Var {
    std::string name;
    double bytes_per_vec;
    std::function<void(d2*, hipStream_t)> run;
};

int main(int argc, char** argv) {
    const int n_out = argc > 1 ? atoi(argv[1]) : 4;
    const size_t bytes = 8000000000ull, n_vecs = bytes / 16;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    d2 *a, *b, *c;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&b, bytes));
    CK(hipMalloc(&c, bytes));
    CK(hipMemset(a, 0x11, bytes));
    CK(hipMemset(b, 0x22, bytes));
    CK(hipMemset(c, 0x33, bytes));
    std::vector<char*> outs(n_out);
    for (auto& o : outs) CK(hipMalloc(&o, bytes));

    std::vector<Var> vars;
#define V(NAME, BPV, KERN, GRID) vars.push_back({NAME, BPV, [=](d2* o, hipStream_t st_) { hipLaunchKernelGGL(KERN, dim3(GRID), dim3(BLOCK), 0, st_, a, b, c, o, n_vecs); }})
    vars.push_back({"write: hipMemsetAsync", 16, [=](d2* o, hipStream_t st_) { CK(hipMemsetAsync(o, 0x5a, bytes, st_)); }});
    V("copy tile U8 nt    bpc6 (shipped)", 32, (k_tile<2, 8, true>), cus * 6);
    V("copy il   U8 plain bpc1", 32, (k_il<2, 8, false>), cus * 1);
    V("copy il   U8 nt    bpc1", 32, (k_il<2, 8, true>), cus * 1);
    V("copy il   U8 plain bpc2", 32, (k_il<2, 8, false>), cus * 2);
    V("copy il   U4 plain bpc2", 32, (k_il<2, 4, false>), cus * 2);
    V("copy il   U4 plain bpc4", 32, (k_il<2, 4, false>), cus * 4);
    V("copy il   U16 plain bpc1", 32, (k_il<2, 16, false>), cus * 1);
    V("add  tile U8 nt    bpc6 (shipped)", 48, (k_tile<3, 8, true>), cus * 6);
    V("add  il   U8 plain bpc1", 48, (k_il<3, 8, false>), cus * 1);
    V("add  il   U8 nt    bpc1", 48, (k_il<3, 8, true>), cus * 1);
    V("add  il   U8 plain bpc2", 48, (k_il<3, 8, false>), cus * 2);
    V("add  il   U4 plain bpc2", 48, (k_il<3, 4, false>), cus * 2);
    V("add  il   U4 plain bpc4", 48, (k_il<3, 4, false>), cus * 4);
    V("add  il   U8 plain bpc3", 48, (k_il<3, 8, false>), cus * 3);
    V("fma  tile U8 nt    bpc6 (shipped)", 64, (k_tile<4, 8, true>), cus * 6);
    V("fma  il   U8 plain bpc1", 64, (k_il<4, 8, false>), cus * 1);
    V("fma  il   U8 plain bpc2", 64, (k_il<4, 8, false>), cus * 2);
    V("fma  il   U4 plain bpc2", 64, (k_il<4, 4, false>), cus * 2);

    printf("%-36s", "kernel");
    for (int o = 0; o < n_out; ++o) printf("   out%d GB/s", o);
    printf("\n");
    for (auto& v : vars) {
        printf("%-36s", v.name.c_str());
        for (int o = 0; o < n_out; ++o) {
            float best = 1e30f;
            for (int round = 0; round < 2; ++round) {
                v.run((d2*)outs[o], s);
                CK(hipEventRecord(e0, s));
                for (int i = 0; i < 3; ++i) v.run((d2*)outs[o], s);
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                best = std::min(best, ms / 3);
            }
            printf("   %9.0f", n_vecs * v.bytes_per_vec / best / 1e6);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
