// ubench_rw.hip — where does the read+write ceiling of MI355X sit, and does operand placement move it?
// Round-2 follow-up to tools/ubench_stream.hip (launch shapes, cache policies: all within 3 %). One process, interleaved
// rounds, best-of; 16-byte accesses, 8 per operand in flight per lane, 6 workgroups per CU (the shipped shape).
//   R      read-only (sum-like)                       8 B/row
//   W      write-only (nt / plain)                    8 B/row
//   C      copy kernel (1R 1W), hipMemcpyAsync D2D   16 B/row
//   A      a + b -> out (2R 1W)                      24 B/row
//   F      a * b + c -> out (3R 1W)                  32 B/row
//   A@k    the same add with b and out shifted by k bytes against 2-MiB-aligned bases: if the three streams of a tile
//          collide on channels / banks when they share their low address bits, a shift changes the rate.
//   A.lds  add with the stores of a tile issued only after EVERY wave of the workgroup has loaded (barrier): load burst,
//          then store burst, per workgroup.
// Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/ubench_rw.hip -o /tmp/ubench_rw && /tmp/ubench_rw
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
typedef d2 d2u __attribute__((aligned(1)));
constexpr int U = 8, BLOCK = 256, WAVES = 4;
constexpr size_t WAVE_VECS = 64 * U, TILE_VECS = WAVE_VECS * WAVES;

__device__ __forceinline__ d2 ld(const d2* p) { return __builtin_nontemporal_load((const d2u*)p); }
template <bool NT>
__device__ __forceinline__ void st(d2* p, d2 v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// MODE: 0 read, 1 write, 2 copy, 3 add, 4 fma. BAR: workgroup barrier between the loads and the stores of a tile.
template <int MODE, bool NTS, bool BAR>
__global__ __launch_bounds__(BLOCK) void rw_kernel(const d2* __restrict__ a, const d2* __restrict__ b,
                                                   const d2* __restrict__ c, d2* __restrict__ out, size_t n_tiles,
                                                   double* sink) {
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    d2 acc = {0.0, 0.0};
    for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const size_t v0 = t * TILE_VECS + wave * WAVE_VECS + lane;
        d2 x[U], y[U], z[U];
        if (MODE != 1) {
#pragma unroll
            for (int u = 0; u < U; ++u) x[u] = ld(a + v0 + (size_t)u * 64);
        }
        if (MODE >= 3) {
#pragma unroll
            for (int u = 0; u < U; ++u) y[u] = ld(b + v0 + (size_t)u * 64);
        }
        if (MODE == 4) {
#pragma unroll
            for (int u = 0; u < U; ++u) z[u] = ld(c + v0 + (size_t)u * 64);
        }
        if (BAR) __syncthreads();
#pragma unroll
        for (int u = 0; u < U; ++u) {
            d2 r;
            if (MODE == 0) { acc += x[u]; continue; }
            else if (MODE == 1) r = d2{1.5, 2.5};
            else if (MODE == 2) r = x[u];
            else if (MODE == 3) r = x[u] + y[u];
            else r = d2{fma(x[u][0], y[u][0], z[u][0]), fma(x[u][1], y[u][1], z[u][1])};
            st<NTS>(out + v0 + (size_t)u * 64, r);
        }
    }
    if (MODE == 0 && acc[0] + acc[1] == 123.456) *sink = acc[0];
}

struct Var {
    std::string name;
    double bytes_per_row;
    std::function<void(hipStream_t)> run;
    double best_ms = 1e30;
};

int main(int argc, char** argv) {
    const size_t rows = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1000000000ull;
    const int rounds = argc > 2 ? atoi(argv[2]) : 4;
    const int reps = 5;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t pad = (size_t)8 << 20;
    const size_t bytes = rows * 8;
    char *a, *b, *c, *out;
    double* sink;
    CK(hipMalloc(&a, bytes + pad));
    CK(hipMalloc(&b, bytes + pad));
    CK(hipMalloc(&c, bytes + pad));
    CK(hipMalloc(&out, bytes + pad));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(a, 0x11, bytes + pad));
    CK(hipMemset(b, 0x22, bytes + pad));
    CK(hipMemset(c, 0x33, bytes + pad));
    printf("bases: a %p b %p c %p out %p (low 21 bits: %lx %lx %lx %lx)\n", (void*)a, (void*)b, (void*)c, (void*)out,
           (unsigned long)((uintptr_t)a & 0x1fffff), (unsigned long)((uintptr_t)b & 0x1fffff),
           (unsigned long)((uintptr_t)c & 0x1fffff), (unsigned long)((uintptr_t)out & 0x1fffff));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const size_t n_tiles = rows * 8 / 16 / TILE_VECS;

    std::vector<Var> vars;
    auto grid_of = [&](int bpc) { return (int)std::min<size_t>(n_tiles, (size_t)cus * bpc); };
#define LAUNCH(MODE, NTS, BAR, bpc, A, B, C, O)                                                                    \
    [=](hipStream_t st_) {                                                                                          \
        hipLaunchKernelGGL((rw_kernel<MODE, NTS, BAR>), dim3(grid_of(bpc)), dim3(BLOCK), 0, st_, (const d2*)(A),   \
                           (const d2*)(B), (const d2*)(C), (d2*)(O), n_tiles, sink);                                \
    }
    for (int bpc : {1, 2, 6}) vars.push_back({"R   read-only            bpc=" + std::to_string(bpc), 8, LAUNCH(0, true, false, bpc, a, b, c, out)});
    for (int bpc : {2, 6, 12}) {
        vars.push_back({"W   write-only nt        bpc=" + std::to_string(bpc), 8, LAUNCH(1, true, false, bpc, a, b, c, out)});
        vars.push_back({"W   write-only plain     bpc=" + std::to_string(bpc), 8, LAUNCH(1, false, false, bpc, a, b, c, out)});
    }
    vars.push_back({"W   hipMemsetAsync", 8, [=](hipStream_t st_) { CK(hipMemsetAsync(out, 0x5a, bytes, st_)); }});
    for (int bpc : {2, 6}) vars.push_back({"C   copy kernel nt       bpc=" + std::to_string(bpc), 16, LAUNCH(2, true, false, bpc, a, b, c, out)});
    vars.push_back({"C   hipMemcpyAsync D2D", 16, [=](hipStream_t st_) { CK(hipMemcpyAsync(out, a, bytes, hipMemcpyDeviceToDevice, st_)); }});
    for (int bpc : {4, 6, 8}) vars.push_back({"A   add nt               bpc=" + std::to_string(bpc), 24, LAUNCH(3, true, false, bpc, a, b, c, out)});
    vars.push_back({"A   add plain stores     bpc=6", 24, LAUNCH(3, false, false, 6, a, b, c, out)});
    vars.push_back({"A.bar add, barrier       bpc=6", 24, LAUNCH(3, true, true, 6, a, b, c, out)});
    vars.push_back({"A.bar add, barrier       bpc=2", 24, LAUNCH(3, true, true, 2, a, b, c, out)});
    vars.push_back({"F   fma nt               bpc=6", 32, LAUNCH(4, true, false, 6, a, b, c, out)});
    vars.push_back({"F   fma nt               bpc=4", 32, LAUNCH(4, true, false, 4, a, b, c, out)});
    // placement: shift b and out against a
    const size_t shifts[][2] = {{256, 512}, {4096, 8192}, {65536 + 4096, 131072 + 8192}, {(1 << 20) + 4096, (2 << 20) + 8192 + 65536},
                                {2048, 1024 + 65536}, {16384, 32768}};
    for (auto& sh : shifts) {
        char nm[96];
        snprintf(nm, sizeof(nm), "A@  add b+%zu out+%zu  bpc=6", sh[0], sh[1]);
        vars.push_back({nm, 24, LAUNCH(3, true, false, 6, a, b + sh[0], c, out + sh[1])});
    }
    vars.push_back({"C@  copy out+4096+64K    bpc=6", 16, LAUNCH(2, true, false, 6, a, b, c, out + 4096 + 65536)});
    vars.push_back({"C   copy kernel nt       bpc=6 (again)", 16, LAUNCH(2, true, false, 6, a, b, c, out)});

    for (int r = 0; r < rounds; ++r) {
        for (auto& v : vars) {
            v.run(s);
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < reps; ++i) v.run(s);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            v.best_ms = std::min<double>(v.best_ms, ms / reps);
        }
    }
    for (auto& v : vars)
        printf("%-44s %8.4f ms  %8.1f GB/s  %5.1f %% of 8 TB/s\n", v.name.c_str(), v.best_ms,
               rows * v.bytes_per_row / v.best_ms / 1e6, rows * v.bytes_per_row / v.best_ms / 1e6 / 80.0);
    return 0;
}
