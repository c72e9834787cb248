// probe_rcp_f64.hip — accuracy of v_rcp_f64 on gfx950 (the seed of the f32 Power path's ln): max |d*r - 1| raw and after one Newton step.
//   hipcc -O3 --offload-arch=gfx950 tools/probe_rcp_f64.hip -o /tmp/rcp && /tmp/rcp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(double* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    double worst0 = 0, worst1 = 0;
    for (int j = 0; j < 4096; ++j) {
        double d = 1.7 + (double)((i * 4096 + j) % 1000003) * (0.72 / 1000003.0) + 1e-9 * j;
        double r0 = __builtin_amdgcn_rcp(d);
        double r1 = fma(fma(-d, r0, 1.0), r0, r0);
        double e0 = fabs(fma(d, r0, -1.0)), e1 = fabs(fma(d, r1, -1.0));
        worst0 = fmax(worst0, e0);
        worst1 = fmax(worst1, e1);
    }
    out[2 * i] = worst0;
    out[2 * i + 1] = worst1;
}
int main() {
    const int n = 256 * 64;
    double* d;
    (void)hipMalloc(&d, n * 16);
    hipLaunchKernelGGL(k, dim3(64), dim3(256), 0, 0, d, n);
    double* h = new double[2 * n];
    (void)hipMemcpy(h, d, n * 16, hipMemcpyDeviceToHost);
    double w0 = 0, w1 = 0;
    for (int i = 0; i < n; ++i) { w0 = fmax(w0, h[2 * i]); w1 = fmax(w1, h[2 * i + 1]); }
    printf("v_rcp_f64 max |d*r-1| = %.3e (2^%.1f); after one Newton step %.3e (2^%.1f)\n", w0, log2(w0), w1, log2(w1));
    return 0;
}
