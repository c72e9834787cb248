// ubench_phase.hip — can a copy run as chip-wide READ and WRITE phases? (a read-only stream runs at 7.2 TB/s,
// a tight-front write-only stream at 6.4, their fine-grained mix at 5.4-6.3.) One workgroup per CU, every lane holds U
// 16-byte vectors: a workgroup reads BLOCK * U * 16 bytes in one burst, then writes them in one burst; all workgroups
// start together and share the bandwidth, so their phases may stay aligned without any barrier. Tile mapping il-style:
// access u of a round is one contiguous front over the whole grid.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_phase.hip -o /tmp/ubench_phase && /tmp/ubench_phase [n_out=3]
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

template <int BLOCK, int U, bool NTS>
__global__ __launch_bounds__(BLOCK) void k_phase(const d2* __restrict__ a, d2* __restrict__ out, size_t n_vecs) {
    const size_t n_threads = (size_t)gridDim.x * BLOCK, tid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const size_t round = n_threads * U;
    for (size_t k = 0; k + round <= n_vecs; k += round) {
        d2 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = __builtin_nontemporal_load(a + k + (size_t)u * n_threads + tid);
        // every load has landed before the first store is issued: a read burst, then a write burst
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NTS) __builtin_nontemporal_store(x[u], out + k + (size_t)u * n_threads + tid);
            else out[k + (size_t)u * n_threads + tid] = x[u];
        }
    }
}

struct Var {
    std::string name;
    std::function<void(d2*, hipStream_t)> run;
};

int main(int argc, char** argv) {
    const int n_out = argc > 1 ? atoi(argv[1]) : 3;
    const size_t bytes = 8000000000ull, n_vecs = bytes / 16;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    d2* a;
    CK(hipMalloc(&a, bytes));
    CK(hipMemset(a, 0x11, bytes));
    std::vector<char*> outs(n_out);
    for (auto& o : outs) CK(hipMalloc(&o, bytes));
    std::vector<Var> vars;
#define V(NAME, B, U_, NTS, GRID) vars.push_back({NAME, [=](d2* o, hipStream_t st_) { hipLaunchKernelGGL((k_phase<B, U_, NTS>), dim3(GRID), dim3(B), 0, st_, a, o, n_vecs); }})
    V("B256  U8   plain 1/CU  (32 KiB per CU per phase)", 256, 8, false, cus);
    V("B256  U32  plain 1/CU  (128 KiB)", 256, 32, false, cus);
    V("B256  U64  plain 1/CU  (256 KiB)", 256, 64, false, cus);
    V("B256  U96  plain 1/CU  (384 KiB)", 256, 96, false, cus);
    V("B1024 U8   plain 1/CU  (128 KiB)", 1024, 8, false, cus);
    V("B1024 U16  plain 1/CU  (256 KiB)", 1024, 16, false, cus);
    V("B1024 U24  plain 1/CU  (384 KiB)", 1024, 24, false, cus);
    V("B1024 U24  nt    1/CU  (384 KiB)", 1024, 24, true, cus);
    V("B256  U8   nt    6/CU  (the shipped amount in flight)", 256, 8, true, cus * 6);
    printf("%-58s", "copy kernel");
    for (int o = 0; o < n_out; ++o) printf("  out%d GB/s", o);
    printf("\n");
    for (auto& v : vars) {
        printf("%-58s", v.name.c_str());
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
            printf("  %9.0f", 2.0 * bytes / best / 1e6);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
