// apply_datetime_{i32,u32,i64,u64} — src/kernels/arithmetic/dispatch.rs:309-372, :420-427.
// "All other ops delegate directly to the integer kernels": the result validity is merge_bitmasks_to_new(lhs mask,
// rhs mask, len) (per-row AND from bit 0 — the reference does NOT window the masks by the view offset, only the data:
// dispatch.rs:321-324), then the masked or dense integer kernel runs on data[off .. off+len].
// Here the merge is fused: the integer kernel reads BOTH validity bitmaps and ANDs them in registers (BinArgs::words2,
// ma_binary.hpp); the output validity is one word-wise AND launch (data-dependent validity — Div/Rem/FloorDiv — comes
// out of the kernel itself). No merged temporary, no allocation.
#include "ma_common.hpp"

using namespace ma;

namespace ma {
#define MA_DECLARE_TWO_MASKS(TAG, T)                                                                                \
    ma_status apply_int_two_masks_##TAG(ma_ctx* ctx, const T* lhs, size_t lhs_len, const T* rhs, size_t rhs_len,    \
                                        int32_t op, const uint8_t* mask1, size_t off1, const uint8_t* mask2,        \
                                        size_t off2, bool combine_and, T* out, uint8_t* out_mask_bits);
MA_DECLARE_TWO_MASKS(i32, int32_t)
MA_DECLARE_TWO_MASKS(u32, uint32_t)
MA_DECLARE_TWO_MASKS(i64, int64_t)
MA_DECLARE_TWO_MASKS(u64, uint64_t)
}  // namespace ma

#define MA_DEFINE_DATETIME(TAG, T)                                                                                    \
    extern "C" ma_status ma_apply_datetime_##TAG(ma_ctx* ctx, const T* lhs_data, size_t lhs_offset, size_t lhs_len,    \
                                                 const uint8_t* lhs_mask_bits, const T* rhs_data, size_t rhs_offset,   \
                                                 size_t rhs_len, const uint8_t* rhs_mask_bits, int32_t op, T* out,     \
                                                 uint8_t* out_mask_bits, int32_t* out_has_mask) {                      \
        MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");                                            \
        if (out_has_mask) *out_has_mask = (lhs_mask_bits || rhs_mask_bits) ? 1 : 0;                                    \
        if (lhs_len != rhs_len) {                                                                                     \
            set_error("apply_datetime: length mismatch (lhs: %zu, rhs: %zu)", lhs_len, rhs_len);                       \
            return MA_ERR_LENGTH_MISMATCH;                                                                            \
        }                                                                                                             \
        const T* l = lhs_data ? lhs_data + lhs_offset : nullptr;                                                      \
        const T* r = rhs_data ? rhs_data + rhs_offset : nullptr;                                                      \
        if (!lhs_mask_bits && !rhs_mask_bits)                                                                         \
            return ma_apply_int_##TAG(ctx, l, lhs_len, r, rhs_len, op, nullptr, 0, out, nullptr);                      \
        MA_REQUIRE(out_mask_bits != nullptr || lhs_len == 0, MA_ERR_INVALID_ARGUMENT,                                  \
                   "an input carries nulls but out_mask_bits is NULL");                                               \
        if (lhs_len == 0) return MA_OK;                                                                               \
        if (lhs_mask_bits && rhs_mask_bits)                                                                           \
            return apply_int_two_masks_##TAG(ctx, l, lhs_len, r, rhs_len, op, lhs_mask_bits, 0, rhs_mask_bits, 0,      \
                                             true, out, out_mask_bits);                                               \
        return ma_apply_int_##TAG(ctx, l, lhs_len, r, rhs_len, op, lhs_mask_bits ? lhs_mask_bits : rhs_mask_bits, 0,   \
                                  out, out_mask_bits);                                                                \
    }

MA_DEFINE_DATETIME(i32, int32_t)
MA_DEFINE_DATETIME(u32, uint32_t)
MA_DEFINE_DATETIME(i64, int64_t)
MA_DEFINE_DATETIME(u64, uint64_t)
