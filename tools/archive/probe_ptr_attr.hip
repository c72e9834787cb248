// probe_ptr_attr.hip — cost of hipPointerGetAttributes / hipMemGetAddressRange per pointer kind (every ABI call classifies
// its operands and its scalar outputs).
//   hipcc -O3 --offload-arch=gfx950 tools/probe_ptr_attr.hip -o /tmp/ptrattr && /tmp/ptrattr
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

int main() {
    void *dev = nullptr, *pinned = nullptr;
    (void)hipMalloc(&dev, 1 << 20);
    (void)hipHostMalloc(&pinned, 1 << 20, hipHostMallocDefault);
    void* heap = malloc(1 << 20);
    long on_stack = 0;
    struct Case { const char* name; void* p; } cases[] = {{"device", dev}, {"pinned", pinned}, {"heap (pageable)", heap}, {"stack (pageable)", &on_stack}};
    for (auto& c : cases) {
        hipPointerAttribute_t attr;
        const int reps = 20000;
        auto t0 = std::chrono::steady_clock::now();
        int ok = 0;
        for (int i = 0; i < reps; ++i) {
            hipError_t e = hipPointerGetAttributes(&attr, (char*)c.p + (i & 1023));
            ok += e == hipSuccess;
            if (e != hipSuccess) (void)hipGetLastError();
        }
        auto t1 = std::chrono::steady_clock::now();
        printf("%-18s hipPointerGetAttributes %7.3f us per call (%s)\n", c.name, std::chrono::duration<double, std::micro>(t1 - t0).count() / reps,
               ok ? "success" : "error path");
    }
    hipDeviceptr_t base;
    size_t size;
    const int reps = 20000;
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; ++i) (void)hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)((char*)dev + (i & 1023)));
    auto t1 = std::chrono::steady_clock::now();
    printf("%-18s hipMemGetAddressRange   %7.3f us per call\n", "device", std::chrono::duration<double, std::micro>(t1 - t0).count() / reps);
    return 0;
}
