// Batched chunk fan-out: SuperArray (op) SuperArray in ONE launch.
//
// route_super_array_broadcast (src/kernels/broadcast/super_array.rs:180-251) loops over chunk pairs sequentially
// ("// TODO: Parallelise", :193). SuperArrays are rechunked to 8192 rows by default
// (RechunkStrategy::Auto, src/structs/chunked/super_array.rs:51-59): a 10^9-row column is ~122 000 chunks, and a
// launch per chunk would be launch-bound by three orders of magnitude. Here a descriptor table with per-chunk tile
// prefix sums is uploaded once; every workgroup binary-searches its tile's chunk (like concat_kernel) and runs the
// same 16-byte vector body as the single-array kernels (inputs on any element phase), or a row body for ragged tiles.
// Validity: the common mask of a chunk is lhs | rhs (Bitmask::union, :224) or whichever side has one; a second
// launch assembles every chunk's output bitmap word by word.
#include <vector>

#include "ma_binary.hpp"

namespace ma {

struct PairDesc {
    const void* lhs;
    const void* rhs;
    void* out;
    const uint64_t* lw;   // lhs validity words or nullptr
    size_t lo, l_last;
    const uint64_t* rw;   // rhs validity words or nullptr
    size_t ro, r_last;
    uint64_t* ow;         // output validity words or nullptr
    size_t len;
    size_t tile0;         // first tile of this chunk
    size_t word0;         // first output-bitmap word of this chunk in the global word numbering
    unsigned head;        // rows before the first 16-byte boundary of `out`
};

__device__ __forceinline__ int find_pair_by_tile(const PairDesc* __restrict__ d, int n, size_t tile) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (d[mid].tile0 <= tile) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}
__device__ __forceinline__ int find_pair_by_word(const PairDesc* __restrict__ d, int n, size_t word) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (d[mid].word0 <= word) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

__device__ __forceinline__ uint64_t window_word_at(const uint64_t* __restrict__ words, size_t bit_off, size_t last_word,
                                                   size_t j) {
    const size_t b = bit_off + (j << 6);
    const size_t w = b >> 6;
    const unsigned sh = (unsigned)(b & 63);
    uint64_t lo = w <= last_word ? words[w] : 0;
    if (sh == 0) return lo;
    uint64_t hi = (w + 1) <= last_word ? words[w + 1] : 0;
    return (lo >> sh) | (hi << (64 - sh));
}

// common validity of chunk row `row` (64-row word containing it, shifted so that bit 0 = row)
__device__ __forceinline__ unsigned pair_row_valid(const PairDesc& d, size_t row) {
    unsigned v = 0;
    bool any = false;
    if (d.lw) {
        v |= row_bit(d.lw, d.lo + row);
        any = true;
    }
    if (d.rw) {
        v |= row_bit(d.rw, d.ro + row);
        any = true;
    }
    return any ? v : 1u;
}

// FUSE_MASK: the wave that computes a run of rows also writes the run's words of the chunk's output bitmap (the common
// mask it has in registers anyway) — possible when every masked chunk's `out` starts on a 16-byte boundary (head == 0),
// so that runs start on validity-word boundaries; otherwise batched_mask_kernel assembles the bitmaps in a second launch.
template <typename T, int UNROLL, bool FUSE_MASK>
__global__ __launch_bounds__(kBlock) void batched_binary_kernel(const PairDesc* __restrict__ descs, int n_chunks,
                                                                size_t n_tiles, int op, uint32_t* flags) {
    typedef typename Vec16<T>::type V;
    constexpr int R = 16 / (int)sizeof(T);
    constexpr int WPT = R * UNROLL;
    constexpr size_t WAVE_ROWS = (size_t)64 * R * UNROLL;
    constexpr size_t TILE_ROWS = WAVE_ROWS * kWaves;
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bool dz = false;
    for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const int c = find_pair_by_tile(descs, n_chunks, t);
        const PairDesc d = descs[c];
        const T* __restrict__ lhs = (const T*)d.lhs;
        const T* __restrict__ rhs = (const T*)d.rhs;
        T* __restrict__ out = (T*)d.out;
        const bool masked = d.lw != nullptr || d.rw != nullptr;
        const size_t lt = t - d.tile0;
        const size_t r0 = d.head + lt * TILE_ROWS;
        const size_t r1 = r0 + TILE_ROWS < d.len ? r0 + TILE_ROWS : d.len;
        if (lt == 0) {
            for (size_t i = threadIdx.x; i < d.head && i < d.len; i += kBlock) {
                T v = Elem<T>::apply_rt(op, lhs[i], rhs[i], dz);
                if (masked) v = pair_row_valid(d, i) ? v : (T)0;
                out[i] = v;
            }
        }
        const size_t w0 = r0 + (size_t)wave * WAVE_ROWS;  // this wave's run of the tile
        if (w0 >= r1) continue;
        // Stores are 16-byte aligned by construction; inputs may sit on any element phase (load16u). A ragged last tile
        // runs the same vector body over its whole vectors (loads and stores guarded per vector) and finishes the < R
        // rows that remain one by one; whole runs take the unguarded body.
        const size_t run_rows = r1 - w0 < WAVE_ROWS ? r1 - w0 : WAVE_ROWS;
        const V* __restrict__ p = (const V*)(lhs + w0) + lane;
        const V* __restrict__ q = (const V*)(rhs + w0) + lane;
        V* __restrict__ o = (V*)(out + w0) + lane;
        uint64_t aw = ~(uint64_t)0;
        if (masked) {
            aw = 0;
            if (d.lw) aw |= load_run_words<WPT>(d.lw, d.lo + w0, d.l_last, lane);
            if (d.rw) aw |= load_run_words<WPT>(d.rw, d.ro + w0, d.r_last, lane);
            if (FUSE_MASK && lane < (unsigned)WPT) {  // lane k holds run word k = word w0 / 64 + k of the chunk's bitmap
                const size_t j = (w0 >> 6) + lane;
                const size_t first = j << 6;
                if (first < d.len) {
                    uint64_t w = aw;
                    if (d.len - first < 64) w &= (((uint64_t)1) << (d.len - first)) - 1;
                    d.ow[j] = w;
                }
            }
        }
        if (run_rows == WAVE_ROWS) {
            V va[UNROLL], vb[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) va[u] = load16u<V, true>(p + (size_t)u * 64);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) vb[u] = load16u<V, true>(q + (size_t)u * 64);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                unsigned bits = ~0u;
                if (masked) bits = lane_bits<R>(aw, u, lane);
                V r;
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    T v = Elem<T>::apply_rt(op, (T)va[u][k], (T)vb[u][k], dz);
                    v = ((bits >> k) & 1u) ? v : (T)0;
                    r[k] = v;
                }
                store16<V, true>(o + (size_t)u * 64, r);
            }
        } else {
            const unsigned n_vec = (unsigned)(run_rows / R);
            for (int u = 0; u < UNROLL; ++u) {
                unsigned bits = ~0u;
                if (masked) bits = lane_bits<R>(aw, u, lane);  // wave-wide shuffle: outside the per-lane guard
                if ((unsigned)u * 64 + lane < n_vec) {
                    const V a = load16u<V, true>(p + (size_t)u * 64);
                    const V b = load16u<V, true>(q + (size_t)u * 64);
                    V r;
#pragma unroll
                    for (int k = 0; k < R; ++k) {
                        T v = Elem<T>::apply_rt(op, (T)a[k], (T)b[k], dz);
                        v = ((bits >> k) & 1u) ? v : (T)0;
                        r[k] = v;
                    }
                    store16<V, true>(o + (size_t)u * 64, r);
                }
            }
            const size_t tail0 = w0 + (size_t)n_vec * R;
            if (tail0 + lane < w0 + run_rows) {
                const size_t i = tail0 + lane;
                T v = Elem<T>::apply_rt(op, lhs[i], rhs[i], dz);
                if (masked) v = pair_row_valid(d, i) ? v : (T)0;
                out[i] = v;
            }
        }
    }
    if constexpr (std::is_integral<T>::value) {
        // only dense chunks can latch (masked integer division is routed chunk by chunk on the host)
        if (__any(dz) && lane == 0) atomicOr(flags, 1u);
    }
}

// One thread per output validity word over all chunks: word = lhs window | rhs window (or the single one).
__global__ __launch_bounds__(kBlock) void batched_mask_kernel(const PairDesc* __restrict__ descs, int n_chunks,
                                                              size_t n_words) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t g = (size_t)blockIdx.x * kBlock + threadIdx.x; g < n_words; g += stride) {
        const int c = find_pair_by_word(descs, n_chunks, g);
        const PairDesc& d = descs[c];
        if (d.ow == nullptr) continue;
        const size_t j = g - d.word0;
        const size_t chunk_words = (d.len + 63) >> 6;
        if (j >= chunk_words) continue;
        uint64_t w = 0;
        if (d.lw) w |= window_word_at(d.lw, d.lo, d.l_last, j);
        if (d.rw) w |= window_word_at(d.rw, d.ro, d.r_last, j);
        if (j == chunk_words - 1 && (d.len & 63)) w &= (((uint64_t)1) << (d.len & 63)) - 1;
        d.ow[j] = w;
    }
}

template <typename T, int U>
static ma_status batched_impl_u(ma_ctx* ctx, int op, size_t n_chunks, const void* const* lhs_data, const size_t* lens,
                                const uint8_t* const* lhs_masks, const void* const* rhs_data,
                                const uint8_t* const* rhs_masks, const uint8_t* override_mask, void* const* out_data,
                                uint8_t* const* out_masks, int32_t* out_has_mask) {
    constexpr int R = 16 / (int)sizeof(T);
    constexpr size_t TILE_ROWS = (size_t)64 * R * U * kWaves;
    MA_ENTER(ctx);
    MA_NO_CAPTURE(ctx, "route_super_array_broadcast (descriptor upload)");
    MA_HIP(hipSetDevice(ctx->device));
    CallScope scope(ctx);
    PairDesc* descs = nullptr;  // built in the context's pinned staging buffer: no second copy of a multi-megabyte table
    MA_TRY(table_begin(ctx, sizeof(PairDesc) * n_chunks, (void**)&descs));
    size_t n_tiles = 0, n_words = 0;
    bool any_mask = false, masked_head = false;
    DeviceRange lhs_role, rhs_role, out_role, lm_role, rm_role, om_role;
    for (size_t i = 0; i < n_chunks; ++i) {
        PairDesc& d = descs[i];
        memset(&d, 0, sizeof(d));
        const size_t n = lens[i];
        d.len = n;
        d.tile0 = n_tiles;
        d.word0 = n_words;
        const uint8_t* lm = override_mask ? override_mask : (lhs_masks ? lhs_masks[i] : nullptr);
        const uint8_t* rm = override_mask ? nullptr : (rhs_masks ? rhs_masks[i] : nullptr);
        if (out_has_mask) out_has_mask[i] = (lm || rm) ? 1 : 0;
        if (n == 0) continue;
        MA_REQUIRE(lhs_data[i] && rhs_data[i] && out_data[i], MA_ERR_INVALID_ARGUMENT, "chunk %zu: NULL buffer", i);
        // A chunked column's pointers run through a few allocations: each operand role remembers the device range its
        // last pointer fell into, so the common case is two compares per pointer (no classification call — six of those
        // per chunk pair were most of the 38 ns a pair cost the host, as long as the kernel itself at 60 000 pairs).
        if (lhs_role.holds(lhs_data[i])) {
            d.lhs = lhs_data[i];
        } else {
            const void* p = nullptr;
            MA_TRY(scope.in(lhs_data[i], n * sizeof(T), &p));
            d.lhs = p;
            lhs_role.learn(lhs_data[i]);
        }
        if (rhs_role.holds(rhs_data[i])) {
            d.rhs = rhs_data[i];
        } else {
            const void* p = nullptr;
            MA_TRY(scope.in(rhs_data[i], n * sizeof(T), &p));
            d.rhs = p;
            rhs_role.learn(rhs_data[i]);
        }
        if (out_role.holds(out_data[i])) {
            d.out = out_data[i];
        } else {
            void* po = nullptr;
            MA_TRY(scope.out(out_data[i], n * sizeof(T), &po));
            d.out = po;
            out_role.learn(out_data[i]);
        }
        if (lm) {
            if (lm_role.holds(lm)) {
                const uintptr_t addr = (uintptr_t)lm, base = addr & ~(uintptr_t)7;  // CallScope::in_mask's re-basing
                d.lw = (const uint64_t*)base;
                d.lo = (size_t)(addr - base) * 8;
            } else {
                MA_TRY(scope.in_mask(lm, 0, n, &d.lw, &d.lo));
                lm_role.learn(lm);
            }
            d.l_last = (d.lo + n - 1) >> 6;
        }
        if (rm) {
            if (rm_role.holds(rm)) {
                const uintptr_t addr = (uintptr_t)rm, base = addr & ~(uintptr_t)7;
                d.rw = (const uint64_t*)base;
                d.ro = (size_t)(addr - base) * 8;
            } else {
                MA_TRY(scope.in_mask(rm, 0, n, &d.rw, &d.ro));
                rm_role.learn(rm);
            }
            d.r_last = (d.ro + n - 1) >> 6;
        }
        if (lm || rm) {
            MA_REQUIRE(out_masks && out_masks[i], MA_ERR_INVALID_ARGUMENT, "chunk %zu carries nulls but has no output bitmap", i);
            if (om_role.holds(out_masks[i]) && ((uintptr_t)out_masks[i] & 7) == 0) {
                d.ow = (uint64_t*)out_masks[i];
            } else {
                MA_TRY(scope.out_mask(out_masks[i], n, &d.ow));
                om_role.learn(out_masks[i]);
            }
            any_mask = true;
        }
        const uintptr_t mis = (uintptr_t)d.out & 15;
        d.head = mis ? (unsigned)((16 - mis) / sizeof(T)) : 0;
        if (d.head && (lm || rm)) masked_head = true;
        n_tiles += n > d.head ? (n - d.head + TILE_ROWS - 1) / TILE_ROWS : 1;
        n_words += (n + 63) >> 6;
    }
    if (n_tiles == 0) return MA_OK;
    void* ddesc = nullptr;
    MA_TRY(ctx_scratch(ctx, sizeof(PairDesc) * n_chunks, &ddesc));
    MA_TRY(table_commit(ctx, descs, sizeof(PairDesc) * n_chunks, ddesc));
    const PairDesc* dd = (const PairDesc*)ddesc;
    const bool fuse = any_mask && !masked_head && !(ctx->variant & 64);  // variant bit 64: always the separate bitmap launch
    if (any_mask && !fuse) {
        int grid = grid_for(ctx, (n_words + kBlock - 1) / kBlock, 8);
        hipLaunchKernelGGL(batched_mask_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, dd, (int)n_chunks, n_words);
        MA_HIP(hipGetLastError());
    }
    {
        int grid = grid_for(ctx, n_tiles, 6);
        if (fuse)
            hipLaunchKernelGGL((batched_binary_kernel<T, U, true>), dim3(grid), dim3(kBlock), 0, ctx->stream, dd, (int)n_chunks,
                               n_tiles, op, ctx->dev_flags);
        else
            hipLaunchKernelGGL((batched_binary_kernel<T, U, false>), dim3(grid), dim3(kBlock), 0, ctx->stream, dd, (int)n_chunks,
                               n_tiles, op, ctx->dev_flags);
        MA_HIP(hipGetLastError());
    }
    MA_TRY(end_call(ctx, scope));
    const bool int_div = std::is_integral<T>::value && (op == MA_OP_DIVIDE || op == MA_OP_REMAINDER || op == MA_OP_FLOORDIV);
    if (int_div && is_async(ctx) && !scope.staged()) {
        ctx->pending_flags = true;  // like the dense kernels of an async context: reported by the next synchronize
        return MA_OK;
    }
    if (int_div) {
        uint32_t f = 0;
        MA_HIP(hipMemcpyAsync(&f, ctx->dev_flags, sizeof(f), hipMemcpyDeviceToHost, ctx->stream));
        MA_HIP(hipStreamSynchronize(ctx->stream));
        if (f & 1u) {
            MA_HIP(hipMemsetAsync(ctx->dev_flags, 0, sizeof(f), ctx->stream));
            MA_HIP(hipStreamSynchronize(ctx->stream));
            set_error("Super Array broadcasting error - division by zero in a dense integer chunk");
            return MA_ERR_DIVIDE_BY_ZERO;
        }
    }
    return MA_OK;
}

// Rows per wave step: 8 x 16 bytes per lane (the single-array kernels' shape) for chunks long enough to fill such tiles,
// 4 x 16 bytes when the chunks are short (RechunkStrategy::Auto's 8192-row chunks: fewer ragged tiles). variant bit 16
// forces 4, bit 32 forces 8 (tuning).
template <typename T>
static ma_status batched_impl(ma_ctx* ctx, int op, size_t n_chunks, const void* const* lhs_data, const size_t* lens,
                              const uint8_t* const* lhs_masks, const void* const* rhs_data,
                              const uint8_t* const* rhs_masks, const uint8_t* override_mask, void* const* out_data,
                              uint8_t* const* out_masks, int32_t* out_has_mask) {
    size_t total = 0;
    for (size_t i = 0; i < n_chunks; ++i) total += lens[i];
    constexpr size_t kWideTileRows = (size_t)64 * (16 / sizeof(T)) * 8 * kWaves;
    bool wide = n_chunks && total / n_chunks >= 16 * kWideTileRows;
    if (ctx->variant & 16) wide = false;
    if (ctx->variant & 32) wide = true;
    if (wide)
        return batched_impl_u<T, 8>(ctx, op, n_chunks, lhs_data, lens, lhs_masks, rhs_data, rhs_masks, override_mask, out_data, out_masks, out_has_mask);
    return batched_impl_u<T, 4>(ctx, op, n_chunks, lhs_data, lens, lhs_masks, rhs_data, rhs_masks, override_mask, out_data, out_masks, out_has_mask);
}

}  // namespace ma

namespace ma {

// Batched form of route_super_array_broadcast for same-type chunk pairs (internal: declared in ma_common.hpp). The
// caller (ma_route_super_array_broadcast in ma_arrow.hip) decides which calls need the per-chunk path instead.
ma_status route_batched(ma_ctx* ctx, int32_t format_code, int32_t op, size_t n_chunks, const void* const* lhs_data,
                        const size_t* lens, const uint8_t* const* lhs_masks, const void* const* rhs_data,
                        const uint8_t* const* rhs_masks, const uint8_t* override_mask, void* const* out_data,
                        uint8_t* const* out_masks, int32_t* out_has_mask) {
    switch (format_code) {
        case 'i': return batched_impl<int32_t>(ctx, op, n_chunks, lhs_data, lens, lhs_masks, rhs_data, rhs_masks, override_mask, out_data, out_masks, out_has_mask);
        case 'I': return batched_impl<uint32_t>(ctx, op, n_chunks, lhs_data, lens, lhs_masks, rhs_data, rhs_masks, override_mask, out_data, out_masks, out_has_mask);
        case 'l': return batched_impl<int64_t>(ctx, op, n_chunks, lhs_data, lens, lhs_masks, rhs_data, rhs_masks, override_mask, out_data, out_masks, out_has_mask);
        case 'L': return batched_impl<uint64_t>(ctx, op, n_chunks, lhs_data, lens, lhs_masks, rhs_data, rhs_masks, override_mask, out_data, out_masks, out_has_mask);
        case 'f': return batched_impl<float>(ctx, op, n_chunks, lhs_data, lens, lhs_masks, rhs_data, rhs_masks, override_mask, out_data, out_masks, out_has_mask);
        case 'g': return batched_impl<double>(ctx, op, n_chunks, lhs_data, lens, lhs_masks, rhs_data, rhs_masks, override_mask, out_data, out_masks, out_has_mask);
        default:
            set_error("unsupported element format '%c'", (char)format_code);
            return MA_ERR_UNSUPPORTED;
    }
}

}  // namespace ma
