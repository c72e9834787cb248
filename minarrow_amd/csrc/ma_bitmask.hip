// Bitmask kernels for gfx950 — src/kernels/bitmask/{mod,std,simd,dispatch}.rs.
// A bitmap window is (bits, bit offset, bit length) = BitmaskVT (src/aliases.rs:172). One u64 word = 64 rows
// = one wave64, so these are plain word-wise kernels: one thread per output word, coalesced 8-byte accesses.
#include "ma_device.hpp"

namespace ma {

// Funnel-shifted read of the window word `j` (bits [bit_off + 64 j, +64)), zero beyond `last_word`.
__device__ __forceinline__ uint64_t window_word(const uint64_t* __restrict__ words, size_t bit_off, size_t last_word,
                                                size_t j) {
    const size_t b = bit_off + (j << 6);
    const size_t w = b >> 6;
    const unsigned sh = (unsigned)(b & 63);
    uint64_t lo = w <= last_word ? words[w] : 0;
    if (sh == 0) return lo;
    uint64_t hi = (w + 1) <= last_word ? words[w + 1] : 0;
    return (lo >> sh) | (hi << (64 - sh));
}

__global__ __launch_bounds__(kBlock) void mask_copy_kernel(const uint64_t* __restrict__ words, size_t bit_off,
                                                           size_t last_word, size_t n,
                                                           uint64_t* __restrict__ out_words) {
    const size_t n_words = (n + 63) >> 6;
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x; j < n_words; j += stride) {
        uint64_t w = window_word(words, bit_off, last_word, j);
        if (j == n_words - 1 && (n & 63)) w &= (((uint64_t)1) << (n & 63)) - 1;  // bits >= n are zero
        out_words[j] = w;
    }
}

ma_status launch_mask_copy(ma_ctx* ctx, const uint64_t* words, size_t bit_off, size_t n, uint64_t* out_words) {
    if (n == 0) return MA_OK;
    const size_t n_words = (n + 63) >> 6;
    const size_t last_word = (bit_off + n - 1) >> 6;
    int grid = grid_for(ctx, (n_words + kBlock - 1) / kBlock, 8);
    hipLaunchKernelGGL(mask_copy_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, words, bit_off, last_word, n,
                       out_words);
    MA_HIP(hipGetLastError());
    return MA_OK;
}

}  // namespace ma
