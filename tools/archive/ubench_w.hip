// ubench_w.hip — which WRITE pattern suits the slow-writing regions of HBM?
// hipMemsetAsync reaches 6.3 TB/s on buffers where the library's tiled store pattern reaches 5.5-5.9. Patterns:
//   tile   the shipped one: a wave owns 8 KiB (8 x 1-KiB wave stores), tiles dealt round-robin to workgroups
//   il     piece-interleaved: all waves of the grid sweep memory as ONE front, 1 KiB per wave instruction
//   gs     plain grid-stride: thread i stores vector i, i + threads, ... (what a fill kernel does)
// each with nt / plain stores and several grid sizes, on several separately allocated 8-GB buffers (slow and fast ones).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_w.hip -o /tmp/ubench_w && /tmp/ubench_w [n_buffers=4]
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

template <int U, bool NT>
__global__ __launch_bounds__(BLOCK) void w_tile(d2* __restrict__ out, size_t n_vecs) {
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr size_t WAVE_VECS = 64 * U, TILE_VECS = WAVE_VECS * WAVES;
    const size_t n_tiles = n_vecs / TILE_VECS;
    for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const size_t v0 = t * TILE_VECS + wave * WAVE_VECS + lane;
#pragma unroll
        for (int u = 0; u < U; ++u) st<NT>(out + v0 + (size_t)u * 64, d2{1.5, 2.5});
    }
}

template <int U, bool NT>
__global__ __launch_bounds__(BLOCK) void w_il(d2* __restrict__ out, size_t n_vecs) {
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t n_waves = (size_t)gridDim.x * WAVES, wave_id = (size_t)blockIdx.x * WAVES + wave;
    const size_t n_pieces = n_vecs / 64, round = n_waves * U;
    for (size_t k = 0; k + round <= n_pieces; k += round) {
#pragma unroll
        for (int u = 0; u < U; ++u) st<NT>(out + (k + (size_t)u * n_waves + wave_id) * 64 + lane, d2{1.5, 2.5});
    }
}

template <bool NT>
__global__ __launch_bounds__(BLOCK) void w_gs(d2* __restrict__ out, size_t n_vecs) {
    const size_t stride = (size_t)gridDim.x * BLOCK;
    for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n_vecs; i += stride) st<NT>(out + i, d2{1.5, 2.5});
}

struct Var {
    std::string name;
    std::function<void(d2*, hipStream_t)> run;
};

int main(int argc, char** argv) {
    const int n_buf = argc > 1 ? atoi(argv[1]) : 4;
    const size_t bytes = 8000000000ull, n_vecs = bytes / 16;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<char*> bufs(n_buf);
    for (auto& b : bufs) CK(hipMalloc(&b, bytes));

    std::vector<Var> vars;
    vars.push_back({"hipMemsetAsync", [=](d2* p, hipStream_t st_) { CK(hipMemsetAsync(p, 0x5a, bytes, st_)); }});
#define V(NAME, KERN, GRID) vars.push_back({NAME, [=](d2* p, hipStream_t st_) { hipLaunchKernelGGL(KERN, dim3(GRID), dim3(BLOCK), 0, st_, p, n_vecs); }})
    V("tile U8 nt    bpc6", (w_tile<8, true>), cus * 6);
    V("tile U8 plain bpc6", (w_tile<8, false>), cus * 6);
    V("tile U8 plain bpc2", (w_tile<8, false>), cus * 2);
    V("tile U2 plain bpc8", (w_tile<2, false>), cus * 8);
    V("tile U1 plain bpc8", (w_tile<1, false>), cus * 8);
    V("il   U8 nt    bpc6", (w_il<8, true>), cus * 6);
    V("il   U8 plain bpc6", (w_il<8, false>), cus * 6);
    V("il   U8 plain bpc2", (w_il<8, false>), cus * 2);
    V("il   U4 plain bpc4", (w_il<4, false>), cus * 4);
    V("il   U1 plain bpc8", (w_il<1, false>), cus * 8);
    V("tile U8 plain bpc1", (w_tile<8, false>), cus * 1);
    V("tile U8 nt    bpc1", (w_tile<8, true>), cus * 1);
    V("tile U2 plain bpc1", (w_tile<2, false>), cus * 1);
    V("tile U1 plain bpc1", (w_tile<1, false>), cus * 1);
    V("il   U8 plain bpc1", (w_il<8, false>), cus * 1);
    V("il   U2 plain bpc1", (w_il<2, false>), cus * 1);
    V("il   U1 plain bpc1", (w_il<1, false>), cus * 1);
    V("gs      plain bpc1", (w_gs<false>), cus * 1);
    V("gs      nt    bpc1", (w_gs<true>), cus * 1);
    V("gs      plain 128 WGs", (w_gs<false>), cus / 2);
    V("gs      plain bpc8", (w_gs<false>), cus * 8);
    V("gs      nt    bpc8", (w_gs<true>), cus * 8);
    V("gs      plain bpc2", (w_gs<false>), cus * 2);
    V("gs      plain bpc32", (w_gs<false>), cus * 32);

    printf("%-22s", "pattern");
    for (int b = 0; b < n_buf; ++b) printf("  buf%d GB/s", b);
    printf("\n");
    for (auto& v : vars) {
        printf("%-22s", v.name.c_str());
        for (int b = 0; b < n_buf; ++b) {
            float best = 1e30f;
            for (int round = 0; round < 2; ++round) {
                v.run((d2*)bufs[b], s);
                CK(hipEventRecord(e0, s));
                for (int i = 0; i < 3; ++i) v.run((d2*)bufs[b], s);
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                best = std::min(best, ms / 3);
            }
            printf("  %9.0f", bytes / best / 1e6);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
