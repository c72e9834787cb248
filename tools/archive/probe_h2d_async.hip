// probe_h2d_async.hip — how long does hipMemcpyAsync(host -> device) from PINNED memory hold the calling thread, by size?
// (the descriptor table of a 60 000-chunk SuperArray call is 6.7 MB)
//   hipcc -O3 --offload-arch=gfx950 tools/probe_h2d_async.hip -o /tmp/h2d && /tmp/h2d
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>

#define CK(x)                                                             \
    do {                                                                  \
        hipError_t e = (x);                                               \
        if (e != hipSuccess) {                                            \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e)); \
            return 1;                                                     \
        }                                                                 \
    } while (0)

int main() {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const size_t max_bytes = 64 << 20;
    void *pinned = nullptr, *dev = nullptr;
    CK(hipHostMalloc(&pinned, max_bytes, hipHostMallocPortable));
    CK(hipMalloc(&dev, max_bytes));
    memset(pinned, 1, max_bytes);
    printf("%12s %14s %14s\n", "bytes", "call returns us", "complete us");
    for (size_t bytes : {(size_t)4096, (size_t)65536, (size_t)(1 << 20), (size_t)(6700000), (size_t)(32 << 20)}) {
        double best_call = 1e9, best_done = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipStreamSynchronize(s));
            auto t0 = std::chrono::steady_clock::now();
            CK(hipMemcpyAsync(dev, pinned, bytes, hipMemcpyHostToDevice, s));
            auto t1 = std::chrono::steady_clock::now();
            CK(hipStreamSynchronize(s));
            auto t2 = std::chrono::steady_clock::now();
            best_call = std::min(best_call, std::chrono::duration<double, std::micro>(t1 - t0).count());
            best_done = std::min(best_done, std::chrono::duration<double, std::micro>(t2 - t0).count());
        }
        printf("%12zu %14.1f %14.1f\n", bytes, best_call, best_done);
    }
    return 0;
}
