// What is hipMallocSignalMemory on this runtime? Size rule, pointer attributes, host visibility, and whether a stream wait on
// it (hipStreamWaitValue64) is released by (a) a host store, (b) hipStreamWriteValue64 on another stream of normal / high
// priority while several normal-priority streams exist (HW-queue sharing). Answers feed ma_stamp_alloc and the abort path.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL %s -> %s\n", #x, hipGetErrorString(e_)); } } while (0)
__global__ void store(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
static bool done_within(hipStream_t s, double ms) {
    auto t0 = std::chrono::steady_clock::now();
    while (hipStreamQuery(s) != hipSuccess) {
        (void)hipGetLastError();
        if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > ms) return false;
        std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
    return true;
}
int main() {
    void* p64 = nullptr; void* p8 = nullptr;
    printf("signal 64 B: %s\n", hipGetErrorString(hipExtMallocWithFlags(&p64, 64, hipMallocSignalMemory))); (void)hipGetLastError();
    hipError_t e8 = hipExtMallocWithFlags(&p8, 8, hipMallocSignalMemory);
    printf("signal  8 B: %s ptr %p\n", hipGetErrorString(e8), p8);
    if (e8 != hipSuccess) return 0;
    hipPointerAttribute_t a{};
    CK(hipPointerGetAttributes(&a, p8));
    printf("attributes: type %d device %d devicePointer %p hostPointer %p isManaged %d\n", (int)a.type, a.device, a.devicePointer, a.hostPointer, a.isManaged);
    printf("hipMemset: %s\n", hipGetErrorString(hipMemset(p8, 0, 8)));
    uint64_t v = 0;
    printf("hipMemcpy D2H: %s value %llu\n", hipGetErrorString(hipMemcpy(&v, p8, 8, hipMemcpyDeviceToHost)), (unsigned long long)v);
    // many normal streams first, as a long-lived host process has
    std::vector<hipStream_t> crowd(6);
    for (auto& s : crowd) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    printf("priority range: least %d greatest %d\n", lo, hi);
    hipStream_t waiter, normal, high;
    CK(hipStreamCreateWithFlags(&waiter, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&normal, hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&high, hipStreamNonBlocking, hi));
    void* words[2] = {p8, nullptr};
    CK(hipMalloc(&words[1], 64));
    CK(hipMemset(words[1], 0, 64));
    const char* names[2] = {"signal memory", "device memory"};
    for (int w = 0; w < 2; ++w) {
        uint64_t* word = (uint64_t*)words[w];
        uint64_t seq = 0;
        // (k) a kernel store releases it
        CK(hipStreamWaitValue64(waiter, word, ++seq, hipStreamWaitValueGte, ~0ull));
        hipLaunchKernelGGL(store, dim3(1), dim3(1), 0, normal, word, seq);
        printf("%s: released by a kernel's store on another stream: %d\n", names[w], (int)done_within(waiter, 2000));
        // (a) hipStreamWriteValue64 on each crowd stream in turn: does any share the waiter's HW queue?
        int stuck = 0;
        for (size_t c = 0; c < crowd.size(); ++c) {
            CK(hipStreamWaitValue64(waiter, word, ++seq, hipStreamWaitValueGte, ~0ull));
            CK(hipStreamWriteValue64(crowd[c], word, seq, 0));
            if (!done_within(waiter, 300)) {
                ++stuck;
                printf("%s: write through crowd stream %zu does NOT release the waiter (shared HW queue?)\n", names[w], c);
                CK(hipStreamWriteValue64(high, word, seq, 0));
                printf("%s:   ... a HIGH-priority stream's write releases it: %d\n", names[w], (int)done_within(waiter, 2000));
                (void)done_within(crowd[c], 2000);
            }
        }
        printf("%s: %d of %zu normal-priority writers were stuck behind the waiter\n", names[w], stuck, crowd.size());
        CK(hipStreamWaitValue64(waiter, word, ++seq, hipStreamWaitValueGte, ~0ull));
        CK(hipStreamWriteValue64(high, word, seq, 0));
        printf("%s: released by a high-priority stream's write: %d\n", names[w], (int)done_within(waiter, 2000));
        if (w == 0 && a.type == hipMemoryTypeHost) {  // host store only where the runtime says the memory is the host's
            CK(hipStreamWaitValue64(waiter, word, ++seq, hipStreamWaitValueGte, ~0ull));
            __atomic_store_n(word, seq, __ATOMIC_RELEASE);
            printf("%s: released by a plain host store: %d\n", names[w], (int)done_within(waiter, 2000));
        }
    }
    // pinned host word
    uint64_t* hw = nullptr;
    CK(hipHostMalloc((void**)&hw, 64, hipHostMallocMapped | hipHostMallocPortable));
    *hw = 0;
    hipError_t ew = hipStreamWaitValue64(waiter, hw, 1, hipStreamWaitValueGte, ~0ull);
    printf("pinned host word: hipStreamWaitValue64 -> %s\n", hipGetErrorString(ew));
    if (ew == hipSuccess) {
        printf("pinned host word: stuck before the store: %d\n", (int)!done_within(waiter, 100));
        __atomic_store_n(hw, 1, __ATOMIC_RELEASE);
        printf("pinned host word: released by a plain host store: %d\n", (int)done_within(waiter, 2000));
        CK(hipStreamWaitValue64(waiter, hw, 2, hipStreamWaitValueGte, ~0ull));
        hipLaunchKernelGGL(store, dim3(1), dim3(1), 0, normal, hw, 2);
        printf("pinned host word: released by a kernel's store: %d\n", (int)done_within(waiter, 2000));
    }
    return 0;
}
