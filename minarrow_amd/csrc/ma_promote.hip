// Mixed-type arithmetic with the promotion fused into the kernel.
//
// arithmetic_dispatch promotes (Int32, Float64) / (Float64, Int32) to f64 and (Int32, Float32) / (Float32, Int32)
// to f32 by MATERIALISING two casted Vec64s and calling apply_float_* (src/kernels/routing/arithmetic.rs:244-269,
// 342-373). Here the cast happens in registers: 4 + 8 + 8 = 20 B/row instead of 12 + 16 (cast passes) + 24.
// `x as f64` is exact; `x as f32` rounds to nearest-even in Rust and on gfx950 alike.
// Results are bit-identical to casting first and calling ma_apply_float_* (tests/test_gpu_promote.py).
#include "ma_binary.hpp"

namespace ma {

template <typename LT, typename RT, typename OT>
struct PromoteArgs {
    const LT* lhs;
    const RT* rhs;
    OT* out;
    OT scalar;              // the scalar side, already promoted
    size_t n, head, n_tiles;
    const uint64_t* words;
    size_t bit_off, last_word;
    int op, kind;
};

template <typename T, int R>
struct VecN {
    typedef T type __attribute__((ext_vector_type(R)));
};

template <typename LT, typename RT, typename OT, bool MASKED, int UNROLL>
__global__ __launch_bounds__(kBlock) void promote_kernel(PromoteArgs<LT, RT, OT> a) {
    constexpr int R = 16 / (int)sizeof(OT);  // rows per lane per step: the OUTPUT moves 16 bytes
    typedef typename VecN<LT, R>::type VL;
    typedef typename VecN<RT, R>::type VR;
    typedef typename VecN<OT, R>::type VO;
    constexpr int WPT = R * UNROLL;
    constexpr size_t WAVE_ROWS = (size_t)64 * R * UNROLL;
    constexpr size_t TILE_ROWS = WAVE_ROWS * kWaves;
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bool dz = false;
    for (size_t t = blockIdx.x; t < a.n_tiles; t += gridDim.x) {
        const size_t row0 = a.head + t * TILE_ROWS + (size_t)wave * WAVE_ROWS;
        VL vl[UNROLL];
        VR vr[UNROLL];
        if (a.kind != kSA) {
            const VL* __restrict__ p = (const VL*)(a.lhs + row0) + lane;
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) vl[u] = load16u<VL, true>(p + (size_t)u * 64);
        }
        if (a.kind != kAS) {
            const VR* __restrict__ q = (const VR*)(a.rhs + row0) + lane;
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) vr[u] = load16u<VR, true>(q + (size_t)u * 64);
        }
        uint64_t aw = 0;
        if constexpr (MASKED) aw = load_run_words<WPT>(a.words, a.bit_off + row0, a.last_word, lane);
        VO* __restrict__ o = (VO*)(a.out + row0) + lane;
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            unsigned bits = ~0u;
            if constexpr (MASKED) bits = lane_bits<R>(aw, u, lane);
            VO r;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                OT x = a.kind == kSA ? a.scalar : (OT)vl[u][k];
                OT y = a.kind == kAS ? a.scalar : (OT)vr[u][k];
                OT v = Elem<OT>::apply_rt(a.op, x, y, dz);
                if constexpr (MASKED) v = ((bits >> k) & 1u) ? v : (OT)0;
                r[k] = v;
            }
            __builtin_nontemporal_store(r, o + (size_t)u * 64);
        }
    }
    // ragged rows: the unaligned head and the tail after the last full tile (last workgroup), or — when the operands
    // do not share a vector phase and there are no tiles at all — every row, grid-strided over all workgroups.
    const size_t tail_start = a.head + a.n_tiles * TILE_ROWS;
    const size_t n_ragged = a.head + (a.n - tail_start);
    const bool everyone = a.n_tiles == 0;
    if (everyone || blockIdx.x == gridDim.x - 1) {
        const size_t first = everyone ? (size_t)blockIdx.x * kBlock + threadIdx.x : threadIdx.x;
        const size_t step = everyone ? (size_t)gridDim.x * kBlock : kBlock;
        for (size_t i = first; i < n_ragged; i += step) {
            size_t row = i < a.head ? i : tail_start + (i - a.head);
            OT x = a.kind == kSA ? a.scalar : (OT)a.lhs[row];
            OT y = a.kind == kAS ? a.scalar : (OT)a.rhs[row];
            OT v = Elem<OT>::apply_rt(a.op, x, y, dz);
            if constexpr (MASKED) v = row_bit(a.words, a.bit_off + row) ? v : (OT)0;
            a.out[row] = v;
        }
    }
}

// Enqueues one promote launch over device-reachable operands (a.lhs / a.rhs / a.out / a.n and, when masked,
// a.words + a.bit_off set by the caller).
template <typename LT, typename RT, typename OT>
static ma_status enqueue_promote(ma_ctx* ctx, PromoteArgs<LT, RT, OT> a, bool masked) {
    const size_t n = a.n;
    if (masked) a.last_word = (a.bit_off + n - 1) >> 6;
    constexpr int R = 16 / (int)sizeof(OT);
    constexpr int U = 8;  // launch shape of the elementwise kernels: 8 accesses per operand in flight, 6 workgroups per CU
    const size_t tile_rows = (size_t)64 * R * U * kWaves;
    // vector path: `head` rows are peeled so that the output stores are 16-byte aligned; inputs are read with
    // element-aligned vector loads (load16u)
    const uintptr_t mis = (uintptr_t)a.out & 15;
    size_t head = mis ? (16 - mis) / sizeof(OT) : 0;
    if (head > n) head = n;
    a.head = head;
    a.n_tiles = (n - head) / tile_rows;
    int grid = a.n_tiles ? grid_for(ctx, a.n_tiles, 6) : grid_for(ctx, (n + kBlock - 1) / kBlock, 8);
    if (masked) hipLaunchKernelGGL((promote_kernel<LT, RT, OT, true, U>), dim3(grid), dim3(kBlock), 0, ctx->stream, a);
    else hipLaunchKernelGGL((promote_kernel<LT, RT, OT, false, U>), dim3(grid), dim3(kBlock), 0, ctx->stream, a);
    MA_HIP(hipGetLastError());
    return MA_OK;
}

// One tile of a host-resident call (run_tiled, ma_pipeline.hip): operand order lhs, rhs, out.
template <typename LT, typename RT, typename OT>
struct PromoteTile {
    ma_ctx* ctx;
    PromoteArgs<LT, RT, OT> base;
    bool masked;
    static ma_status run(void* user, size_t row0, size_t rows, void* const* ptrs) {
        const PromoteTile& t = *(const PromoteTile*)user;
        PromoteArgs<LT, RT, OT> a = t.base;
        a.lhs = (const LT*)ptrs[0];
        a.rhs = (const RT*)ptrs[1];
        a.out = (OT*)ptrs[2];
        a.n = rows;
        if (t.masked) {
            const size_t bit = t.base.bit_off + row0;
            a.words = t.base.words + (bit >> 6);
            a.bit_off = bit & 63;
        }
        return enqueue_promote<LT, RT, OT>(t.ctx, a, t.masked);
    }
};

template <typename LT, typename RT, typename OT>
static ma_status promote_impl(ma_ctx* ctx, int kind, const LT* lhs, size_t lhs_len, const RT* rhs, size_t rhs_len, OT scalar,
                              int op, const uint8_t* mask_bits, size_t mask_bit_offset, OT* out, uint8_t* out_mask_bits) {
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    MA_REQUIRE(op >= MA_OP_ADD && op <= MA_OP_FLOORDIV, MA_ERR_INVALID_ARGUMENT, "unknown ArithmeticOperator code %d", op);
    if (kind == kAA && lhs_len != rhs_len) {
        set_error("arithmetic_dispatch => Length mismatch: LHS %zu RHS %zu", lhs_len, rhs_len);  // arithmetic.rs:235-241
        return MA_ERR_LENGTH_MISMATCH;
    }
    const size_t n = kind == kSA ? rhs_len : lhs_len;
    if (n == 0) return MA_OK;
    const bool masked = mask_bits != nullptr;
    MA_REQUIRE(out != nullptr && (kind == kSA || lhs) && (kind == kAS || rhs), MA_ERR_INVALID_ARGUMENT, "NULL buffer");
    MA_REQUIRE(!masked || out_mask_bits, MA_ERR_INVALID_ARGUMENT, "a masked call needs an output bitmap");
    MA_REQUIRE(((uintptr_t)lhs % sizeof(LT)) == 0 && ((uintptr_t)rhs % sizeof(RT)) == 0 && ((uintptr_t)out % sizeof(OT)) == 0,
               MA_ERR_INVALID_ARGUMENT, "a data pointer is not aligned to its element size");
    MA_ENTER(ctx);
    MA_HIP(hipSetDevice(ctx->device));
    CallScope scope(ctx);
    PromoteArgs<LT, RT, OT> a{};
    a.scalar = scalar;
    a.n = n;
    a.op = op;
    a.kind = kind;
    uint64_t* out_words = nullptr;

    // host-resident columns cross PCIe in tiles (ma_pipeline.hip); the tile is sized by the widest operand
    const size_t tile_rows = (ctx->staging_tile_bytes / sizeof(OT)) & ~(size_t)32767;
    if (tile_rows && n >= 2 * tile_rows && !ctx->capturing) {
        PipeOperand ops[3] = {{kind != kSA ? lhs : nullptr, nullptr, sizeof(LT), false},
                              {kind != kAS ? rhs : nullptr, nullptr, sizeof(RT), false},
                              {nullptr, out, sizeof(OT), false}};
        bool any = false;
        for (auto& o : ops) {
            const void* q = o.out ? o.out : o.in;
            o.staged = q != nullptr && crosses_in_tiles(ctx, pointer_kind(q));
            any = any || o.staged;
        }
        if (any) {
            if (masked) {
                MA_TRY(scope.in_mask(mask_bits, mask_bit_offset, n, &a.words, &a.bit_off));
                MA_TRY(scope.out_mask(out_mask_bits, n, &out_words));
                MA_TRY(launch_mask_copy(ctx, a.words, a.bit_off, n, out_words));
            }
            PromoteTile<LT, RT, OT> call{ctx, a, masked};
            MA_TRY(run_tiled(ctx, n, tile_rows, ops, 3, &PromoteTile<LT, RT, OT>::run, &call));
            return scope.finish();
        }
    }

    const void* p = nullptr;
    if (kind != kSA) {
        MA_TRY(scope.in(lhs, n * sizeof(LT), &p));
        a.lhs = (const LT*)p;
    }
    if (kind != kAS) {
        MA_TRY(scope.in(rhs, n * sizeof(RT), &p));
        a.rhs = (const RT*)p;
    }
    void* po = nullptr;
    MA_TRY(scope.out(out, n * sizeof(OT), &po));
    a.out = (OT*)po;
    if (masked) {
        MA_TRY(scope.in_mask(mask_bits, mask_bit_offset, n, &a.words, &a.bit_off));
        MA_TRY(scope.out_mask(out_mask_bits, n, &out_words));
        MA_TRY(launch_mask_copy(ctx, a.words, a.bit_off, n, out_words));  // float ops: validity out == validity in
    }
    MA_TRY((enqueue_promote<LT, RT, OT>(ctx, a, masked)));
    return end_call(ctx, scope);
}

}  // namespace ma

using namespace ma;

#define MA_DEFINE_PROMOTE(LTAG, LT, RTAG, RT, OT)                                                                       \
    extern "C" ma_status ma_apply_promote_##LTAG##_##RTAG(ma_ctx* ctx, const LT* lhs, size_t lhs_len, const RT* rhs,    \
                                                          size_t rhs_len, int32_t op, const uint8_t* mask_bits,         \
                                                          size_t mask_bit_offset, OT* out, uint8_t* out_mask_bits) {    \
        return promote_impl<LT, RT, OT>(ctx, kAA, lhs, lhs_len, rhs, rhs_len, (OT)0, op, mask_bits, mask_bit_offset,    \
                                        out, out_mask_bits);                                                            \
    }                                                                                                                   \
    extern "C" ma_status ma_apply_promote_##LTAG##_##RTAG##_scalar_rhs(ma_ctx* ctx, const LT* lhs, size_t lhs_len,      \
                                                                       RT scalar, int32_t op,                          \
                                                                       const uint8_t* mask_bits,                       \
                                                                       size_t mask_bit_offset, OT* out,                \
                                                                       uint8_t* out_mask_bits) {                       \
        return promote_impl<LT, RT, OT>(ctx, kAS, lhs, lhs_len, nullptr, 0, (OT)scalar, op, mask_bits, mask_bit_offset, \
                                        out, out_mask_bits);                                                            \
    }                                                                                                                   \
    extern "C" ma_status ma_apply_promote_##LTAG##_##RTAG##_scalar_lhs(ma_ctx* ctx, LT scalar, const RT* rhs,           \
                                                                       size_t rhs_len, int32_t op,                     \
                                                                       const uint8_t* mask_bits,                       \
                                                                       size_t mask_bit_offset, OT* out,                \
                                                                       uint8_t* out_mask_bits) {                       \
        return promote_impl<LT, RT, OT>(ctx, kSA, nullptr, 0, rhs, rhs_len, (OT)scalar, op, mask_bits, mask_bit_offset, \
                                        out, out_mask_bits);                                                            \
    }

MA_DEFINE_PROMOTE(i32, int32_t, f64, double, double)
MA_DEFINE_PROMOTE(f64, double, i32, int32_t, double)
MA_DEFINE_PROMOTE(i32, int32_t, f32, float, float)
MA_DEFINE_PROMOTE(f32, float, i32, int32_t, float)
