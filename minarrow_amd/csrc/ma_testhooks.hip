// Test-only entry points (include/minarrow_hip_testing.h): the switch that keeps every fault hook of the library inert unless
// MINARROW_HIP_TEST_HOOKS=1 was in the environment when the library was LOADED, and what exposes an internal to the tests (the series
// behind float Power) through the same C ABI as everything else. Nothing in the library calls them.
#include "minarrow_hip_testing.h"

#include "ma_binary.hpp"

namespace ma {

namespace {
// read once, at load: a host that never set the variable cannot have the hooks switched on behind its back later
const bool g_hooks_live = [] {
    const char* e = getenv("MINARROW_HIP_TEST_HOOKS");
    return e && e[0] == '1';
}();
}  // namespace

ma_status test_hooks_enabled() {
    if (g_hooks_live) return MA_OK;
    set_error("test hooks are disabled: they act only when MINARROW_HIP_TEST_HOOKS=1 was set when libminarrow_hip.so was loaded "
              "(include/minarrow_hip_testing.h)");
    return MA_ERR_UNSUPPORTED;
}

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

extern "C" int32_t ma_test_hooks_enabled(void) { return g_hooks_live ? 1 : 0; }

extern "C" ma_status ma_test_pow_series(ma_ctx* ctx, int32_t which, const void* in, double* out, size_t n) {
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    MA_TRY(test_hooks_enabled());
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
