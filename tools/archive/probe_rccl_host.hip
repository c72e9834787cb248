// Host time of enqueueing a tiny RCCL collective on a one-rank communicator (round 4): what does the issuing thread pay per
// ncclAllGather of 64 bytes, compared with a 64-byte hipMemcpyAsync and a kernel launch? And does the collective replay
// from a hipGraph (stream capture) — at what host cost per replay?
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/probe_rccl_host tools/probe_rccl_host.hip -ldl && /tmp/probe_rccl_host
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;
#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            printf("{\"error\": \"%s at line %d\"}\n", hipGetErrorString(e_), __LINE__); \
            return 1;                                                             \
        }                                                                         \
    } while (0)

__global__ void tiny(unsigned long long* p) { p[0] += 1; }

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
    void* h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { printf("{\"error\": \"no librccl\"}\n"); return 1; }
    auto InitAll = (ncclResult_t(*)(ncclComm_t*, int, const int*))dlsym(h, "ncclCommInitAll");
    auto AllGather = (ncclResult_t(*)(const void*, void*, size_t, int, ncclComm_t, hipStream_t))dlsym(h, "ncclAllGather");
    auto AllReduce = (ncclResult_t(*)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t))dlsym(h, "ncclAllReduce");
    auto GroupStart = (ncclResult_t(*)())dlsym(h, "ncclGroupStart");
    auto GroupEnd = (ncclResult_t(*)())dlsym(h, "ncclGroupEnd");
    auto Destroy = (ncclResult_t(*)(ncclComm_t))dlsym(h, "ncclCommDestroy");
    CK(hipSetDevice(0));
    ncclComm_t comm;
    int dev = 0;
    if (InitAll(&comm, 1, &dev) != 0) { printf("{\"error\": \"ncclCommInitAll\"}\n"); return 1; }
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned long long *a, *b;
    CK(hipMalloc(&a, 4096));
    CK(hipMalloc(&b, 4096));
    CK(hipMemset(a, 0, 4096));
    const int N = 300;
    auto timeit = [&](const char* name, auto fn) {
        for (int i = 0; i < 20; ++i) fn();
        (void)hipStreamSynchronize(s);
        const double t0 = now_us();
        for (int i = 0; i < N; ++i) fn();
        const double t1 = now_us();
        (void)hipStreamSynchronize(s);
        const double t2 = now_us();
        printf("{\"what\": \"%s\", \"host_us_per_call\": %.2f, \"us_per_call_incl_drain\": %.2f}\n", name, (t1 - t0) / N, (t2 - t0) / N);
        fflush(stdout);
    };
    timeit("kernel launch", [&] { hipLaunchKernelGGL(tiny, dim3(1), dim3(1), 0, s, a); });
    timeit("hipMemcpyAsync D2D 64 B", [&] { (void)hipMemcpyAsync(b, a, 64, hipMemcpyDeviceToDevice, s); });
    timeit("ncclAllGather 64 B, one rank", [&] { AllGather(a, b, 64, 0 /*ncclChar*/, comm, s); });
    timeit("ncclAllGather 64 B in place, one rank", [&] { AllGather(b, b, 64, 0, comm, s); });
    timeit("ncclAllReduce 8 x i64, one rank", [&] { AllReduce(a, b, 8, 4 /*ncclInt64*/, 0 /*sum*/, comm, s); });
    timeit("ncclGroupStart + ncclAllGather + ncclGroupEnd", [&] { GroupStart(); AllGather(a, b, 64, 0, comm, s); GroupEnd(); });
    // stream capture of [all-gather + kernel], replayed
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    int rc = AllGather(a, b, 64, 0, comm, s);
    hipLaunchKernelGGL(tiny, dim3(1), dim3(1), 0, s, b);
    hipError_t e2 = hipStreamEndCapture(s, &g);
    if (e != hipSuccess || e2 != hipSuccess || rc != 0 || !g) {
        (void)hipGetLastError();
        printf("{\"what\": \"capture of ncclAllGather\", \"ok\": false, \"begin\": %d, \"nccl\": %d, \"end\": %d}\n", (int)e, rc, (int)e2);
    } else {
        size_t nodes = 0;
        (void)hipGraphGetNodes(g, nullptr, &nodes);
        if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) {
            printf("{\"what\": \"instantiate\", \"ok\": false}\n");
        } else {
            printf("{\"what\": \"capture of ncclAllGather\", \"ok\": true, \"nodes\": %zu}\n", nodes);
            timeit("hipGraphLaunch of [ncclAllGather + kernel]", [&] { (void)hipGraphLaunch(ge, s); });
        }
    }
    (void)hipStreamSynchronize(s);
    Destroy(comm);
    return 0;
}
