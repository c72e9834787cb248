// Test-only entry points: what they expose is internal (the series behind float Power), declared in the header's
// "testing hooks" section so that tests/ can hold them to their documented accuracy through the same C ABI as everything
// else. Nothing in the library calls them.
#include "ma_binary.hpp"

namespace ma {

__global__ __launch_bounds__(kBlock) void pow_series_kernel(int which, const void* __restrict__ in, double* __restrict__ out,
                                                            size_t n) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        double r;
        if (which == 0) r = pow_f64_ln(((const double*)in)[i]);
        else if (which == 1) r = pow_f32_ln((double)((const float*)in)[i]);
        else r = pow_f32_exp((double)((const float*)in)[i]);
        out[i] = r;
    }
}

}  // namespace ma

using namespace ma;

extern "C" ma_status ma_test_pow_series(ma_ctx* ctx, int32_t which, const void* in, double* out, size_t n) {
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    MA_REQUIRE(which >= 0 && which <= 2, MA_ERR_INVALID_ARGUMENT, "which = %d (0: f64 ln, 1: f32-path ln, 2: f32-path exp)", which);
    if (n == 0) return MA_OK;
    MA_REQUIRE(in != nullptr && out != nullptr, MA_ERR_INVALID_ARGUMENT, "NULL buffer");
    MA_ENTER(ctx);
    MA_HIP(hipSetDevice(ctx->device));
    CallScope scope(ctx);
    const void* din = nullptr;
    void* dout = nullptr;
    MA_TRY(scope.in(in, n * (which == 0 ? 8 : 4), &din));
    MA_TRY(scope.out(out, n * 8, &dout));
    const int grid = grid_for(ctx, (n + kBlock - 1) / kBlock, 4);
    hipLaunchKernelGGL(pow_series_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, (int)which, din, (double*)dout, n);
    MA_HIP(hipGetLastError());
    return end_call(ctx, scope);
}
