// Arrow C Data Interface entry points — the boundary format of the reference
// (src/ffi/arrow_c_ffi.rs: #[repr(C)] ArrowArray / ArrowSchema; checked from C by tests/c_inspect_arrow.c:17-41).
//
// Numeric primitive arrays only: buffers[0] = validity bitmap (may be NULL), buffers[1] = values;
// format "i" i32, "I" u32, "l" i64, "L" u64, "f" f32, "g" f64 (tests/arrow_c_integration.rs:62-80).
// The reference always exports offset 0 (src/ffi/arrow_c_ffi.rs:1773) and ignores `offset` on import
// (arrow_c_ffi.rs:1098-1111); other producers (PyArrow slices) set it, so it is honoured here for both buffers:
// values start at element `offset`, validity at bit `offset`.
// The library never calls `release`: the producer keeps ownership (arrow_c_ffi.rs:193-262).
#include "ma_common.hpp"

using namespace ma;

namespace {

struct Prim {
    char code;
    size_t size;
};

// allow_narrow: also the 1- and 2-byte integers ('c' 'C' 's' 'S': the reference's extended_numeric_types) — for the
// reductions; the arithmetic type matrix (arithmetic_dispatch, routing/arithmetic.rs:280-339) does not route them.
ma_status parse_primitive(const ArrowArray* array, const ArrowSchema* schema, Prim* out, bool allow_narrow = false) {
    MA_REQUIRE(array != nullptr && schema != nullptr, MA_ERR_INVALID_ARGUMENT, "ArrowArray or ArrowSchema is NULL");
    MA_REQUIRE(schema->format != nullptr, MA_ERR_INVALID_ARGUMENT, "ArrowSchema.format is NULL");
    MA_REQUIRE(array->length >= 0 && array->offset >= 0, MA_ERR_INVALID_ARGUMENT, "negative length or offset");
    MA_REQUIRE(schema->dictionary == nullptr && array->dictionary == nullptr, MA_ERR_UNSUPPORTED,
               "dictionary-encoded arrays are not numeric scans");
    const char* f = schema->format;
    MA_REQUIRE(f[0] != 0 && f[1] == 0, MA_ERR_UNSUPPORTED, "unsupported Arrow format \"%s\" (numeric primitives only)", f);
    switch (f[0]) {
        case 'i': case 'I': case 'f': *out = {f[0], 4}; break;
        case 'l': case 'L': case 'g': *out = {f[0], 8}; break;
        case 'c': case 'C': case 's': case 'S':
            if (allow_narrow) {
                *out = {f[0], (f[0] == 'c' || f[0] == 'C') ? (size_t)1 : (size_t)2};
                break;
            }
            [[fallthrough]];
        default:
            set_error("unsupported Arrow format \"%s\" (numeric primitives only)", f);
            return MA_ERR_UNSUPPORTED;
    }
    MA_REQUIRE(array->n_buffers == 2 && array->buffers != nullptr, MA_ERR_INVALID_ARGUMENT,
               "a primitive array has exactly 2 buffers (validity, values); got %lld", (long long)array->n_buffers);
    MA_REQUIRE(array->length == 0 || array->buffers[1] != nullptr, MA_ERR_INVALID_ARGUMENT, "values buffer is NULL");
    return MA_OK;
}

// One element of a (host or device) values buffer -> host.
ma_status fetch_scalar(ma_ctx* ctx, const void* src, size_t size, void* dst) {
    if (pointer_kind(src) == kPageable || pointer_kind(src) == kPinned) {
        memcpy(dst, src, size);
        return MA_OK;
    }
    return ma_dev_download(ctx, dst, src, size);
}

}  // namespace

extern "C" {

ma_status ma_sum_arrow(ma_ctx* ctx, const struct ArrowArray* array, const struct ArrowSchema* schema,
                       double* out_sum_f64, int64_t* out_sum_i64, uint64_t* out_valid_count) {
    Prim p{};
    MA_TRY(parse_primitive(array, schema, &p, true));
    const size_t n = (size_t)array->length, off = (size_t)array->offset;
    const uint8_t* validity = (const uint8_t*)array->buffers[0];
    const char* values = (const char*)array->buffers[1] + off * p.size;
    // null_count: -1 = unknown; 0 = take the dense kernel even if a bitmap is attached
    const int64_t nc = array->null_count;
    if (out_sum_i64) *out_sum_i64 = 0;
    if (out_sum_f64) *out_sum_f64 = 0.0;
    switch (p.code) {
        case 'l': {
            int64_t s = 0;
            MA_TRY(ma_i64_sum(ctx, (const int64_t*)values, n, validity, off, nc, &s, out_valid_count));
            if (out_sum_i64) *out_sum_i64 = s;
            if (out_sum_f64) *out_sum_f64 = (double)s;
            return MA_OK;
        }
        case 'L': {
            uint64_t s = 0;
            MA_TRY(ma_u64_sum(ctx, (const uint64_t*)values, n, validity, off, nc, &s, out_valid_count));
            if (out_sum_i64) *out_sum_i64 = (int64_t)s;
            if (out_sum_f64) *out_sum_f64 = (double)s;
            return MA_OK;
        }
        case 'i': {
            int64_t s = 0;
            MA_TRY(ma_i32_sum(ctx, (const int32_t*)values, n, validity, off, nc, &s, out_valid_count));
            if (out_sum_i64) *out_sum_i64 = s;
            if (out_sum_f64) *out_sum_f64 = (double)s;
            return MA_OK;
        }
        case 'I': {
            uint64_t s = 0;
            MA_TRY(ma_u32_sum(ctx, (const uint32_t*)values, n, validity, off, nc, &s, out_valid_count));
            if (out_sum_i64) *out_sum_i64 = (int64_t)s;
            if (out_sum_f64) *out_sum_f64 = (double)s;
            return MA_OK;
        }
#define MA_NARROW_SUM(CODE, TAG, T, S)                                                              \
    case CODE: {                                                                                    \
        S s = 0;                                                                                    \
        MA_TRY(ma_##TAG##_sum(ctx, (const T*)values, n, validity, off, nc, &s, out_valid_count));   \
        if (out_sum_i64) *out_sum_i64 = (int64_t)s;                                                 \
        if (out_sum_f64) *out_sum_f64 = (double)s;                                                  \
        return MA_OK;                                                                               \
    }
            MA_NARROW_SUM('c', i8, int8_t, int64_t)
            MA_NARROW_SUM('C', u8, uint8_t, uint64_t)
            MA_NARROW_SUM('s', i16, int16_t, int64_t)
            MA_NARROW_SUM('S', u16, uint16_t, uint64_t)
#undef MA_NARROW_SUM
        case 'f':
            return ma_f32_sum(ctx, (const float*)values, n, validity, off, nc, out_sum_f64, out_valid_count);
        default:
            return ma_f64_sum(ctx, (const double*)values, n, validity, off, nc, out_sum_f64, out_valid_count);
    }
}

ma_status ma_mean_arrow(ma_ctx* ctx, const struct ArrowArray* array, const struct ArrowSchema* schema, double* out_mean,
                        uint64_t* out_valid_count) {
    Prim p{};
    MA_TRY(parse_primitive(array, schema, &p, true));
    const size_t n = (size_t)array->length, off = (size_t)array->offset;
    const uint8_t* validity = (const uint8_t*)array->buffers[0];
    const char* values = (const char*)array->buffers[1] + off * p.size;
    const int64_t nc = array->null_count;
    switch (p.code) {
        case 'c': return ma_i8_mean(ctx, (const int8_t*)values, n, validity, off, nc, out_mean, out_valid_count);
        case 'C': return ma_u8_mean(ctx, (const uint8_t*)values, n, validity, off, nc, out_mean, out_valid_count);
        case 's': return ma_i16_mean(ctx, (const int16_t*)values, n, validity, off, nc, out_mean, out_valid_count);
        case 'S': return ma_u16_mean(ctx, (const uint16_t*)values, n, validity, off, nc, out_mean, out_valid_count);
        case 'l': return ma_i64_mean(ctx, (const int64_t*)values, n, validity, off, nc, out_mean, out_valid_count);
        case 'L': return ma_u64_mean(ctx, (const uint64_t*)values, n, validity, off, nc, out_mean, out_valid_count);
        case 'i': return ma_i32_mean(ctx, (const int32_t*)values, n, validity, off, nc, out_mean, out_valid_count);
        case 'I': return ma_u32_mean(ctx, (const uint32_t*)values, n, validity, off, nc, out_mean, out_valid_count);
        case 'f': return ma_f32_mean(ctx, (const float*)values, n, validity, off, nc, out_mean, out_valid_count);
        default: return ma_f64_mean(ctx, (const double*)values, n, validity, off, nc, out_mean, out_valid_count);
    }
}

// lhs (op) rhs for two primitive arrays, routed like resolve_binary_arithmetic
// (src/kernels/routing/arithmetic.rs:214-222): equal lengths, or one side of length 1 which is broadcast
// (routing/broadcast.rs:87-112, fused here). Validity: none attached -> dense kernel, `out_validity` untouched and
// *out_has_validity = 0; otherwise the AND of the attached bitmaps gates the rows (merge_bitmasks_to_new,
// src/kernels/bitmask/mod.rs:171-196 — the rule apply_datetime_* uses, dispatch.rs:321-322).
ma_status ma_apply_arrow(ma_ctx* ctx, int32_t op, const struct ArrowArray* lhs, const struct ArrowSchema* lhs_schema,
                         const struct ArrowArray* rhs, const struct ArrowSchema* rhs_schema, void* out_values,
                         uint8_t* out_validity, int32_t* out_has_validity) {
    Prim pl{}, pr{};
    MA_TRY(parse_primitive(lhs, lhs_schema, &pl));
    MA_TRY(parse_primitive(rhs, rhs_schema, &pr));
    // Type matrix of arithmetic_dispatch (src/kernels/routing/arithmetic.rs:278-406): same-type pairs, plus
    // Int32 <-> Float64 / Float32 promotions (:342-373); everything else is `UnsupportedType` (:403-405).
    const bool promote = (pl.code == 'i' && (pr.code == 'g' || pr.code == 'f')) ||
                         (pr.code == 'i' && (pl.code == 'g' || pl.code == 'f'));
    if (pl.code != pr.code && !promote) {
        set_error("Unsupported array type combination for arithmetic operations (\"%s\" vs \"%s\")", lhs_schema->format,
                  rhs_schema->format);
        return MA_ERR_UNSUPPORTED;
    }
    const size_t nl = (size_t)lhs->length, nr = (size_t)rhs->length;
    const size_t lo = (size_t)lhs->offset, ro = (size_t)rhs->offset;
    if (out_has_validity) *out_has_validity = 0;
    if (nl != nr && nl != 1 && nr != 1) {
        set_error("cannot broadcast arrays of length %zu and %zu", nl, nr);
        return MA_ERR_LENGTH_MISMATCH;
    }
    const size_t n = nl == nr ? nl : (nl == 1 ? nr : nl);
    const uint8_t* lv = lhs->null_count == 0 ? nullptr : (const uint8_t*)lhs->buffers[0];
    const uint8_t* rv = rhs->null_count == 0 ? nullptr : (const uint8_t*)rhs->buffers[0];
    const bool scalar_l = nl == 1 && nr != 1, scalar_r = nr == 1 && nl != 1;
    // A broadcast length-1 operand contributes no row validity (broadcast_length_1_array builds a mask-free
    // array, routing/broadcast.rs:30-45).
    if (scalar_l) lv = nullptr;
    if (scalar_r) rv = nullptr;
    const uint8_t* mask = nullptr;
    size_t mask_off = 0;
    if (lv && rv) {
        MA_REQUIRE(out_validity != nullptr, MA_ERR_INVALID_ARGUMENT, "out_validity is NULL but both inputs carry nulls");
        // AND of the two windows, written to out_validity at bit 0, then used as the gate.
        uint8_t* la = nullptr;  // left window re-based to bit 0 (scratch = out_validity itself)
        MA_TRY(ma_bitmask_slice(ctx, lv, lo, n, out_validity));
        la = out_validity;
        // rhs window -> temporary device bitmap, AND into out_validity
        void* tmp = nullptr;
        MA_TRY(ma_dev_alloc(ctx, ((n + 63) / 64) * 8 + 8, &tmp));
        ma_status s = ma_bitmask_slice(ctx, rv, ro, n, (uint8_t*)tmp);
        if (s == MA_OK) s = ma_and_masks(ctx, la, 0, (const uint8_t*)tmp, 0, n, out_validity);
        (void)ma_dev_free(ctx, tmp);
        MA_TRY(s);
        mask = out_validity;
        mask_off = 0;
    } else if (lv) {
        mask = lv;
        mask_off = lo;
    } else if (rv) {
        mask = rv;
        mask_off = ro;
    }
    MA_REQUIRE(mask == nullptr || out_validity != nullptr, MA_ERR_INVALID_ARGUMENT,
               "out_validity is NULL but an input carries nulls");
    if (out_has_validity) *out_has_validity = mask ? 1 : 0;
    const char* a = (const char*)lhs->buffers[1] + lo * pl.size;
    const char* b = (const char*)rhs->buffers[1] + ro * pr.size;

#define MA_ROUTE(FAMILY, TAG, T)                                                                                      \
    do {                                                                                                              \
        if (scalar_l) {                                                                                               \
            T sc;                                                                                                     \
            MA_TRY(fetch_scalar(ctx, a, sizeof(T), &sc));                                                          \
            return ma_apply_##FAMILY##_##TAG##_scalar_lhs(ctx, sc, (const T*)b, n, op, mask, mask_off, (T*)out_values, \
                                                          out_validity);                                              \
        }                                                                                                             \
        if (scalar_r) {                                                                                               \
            T sc;                                                                                                     \
            MA_TRY(fetch_scalar(ctx, b, sizeof(T), &sc));                                                          \
            return ma_apply_##FAMILY##_##TAG##_scalar_rhs(ctx, (const T*)a, n, sc, op, mask, mask_off, (T*)out_values, \
                                                          out_validity);                                              \
        }                                                                                                             \
        return ma_apply_##FAMILY##_##TAG(ctx, (const T*)a, nl, (const T*)b, nr, op, mask, mask_off, (T*)out_values,    \
                                         out_validity);                                                               \
    } while (0)

#define MA_ROUTE_PROMOTE(LTAG, LT, RTAG, RT, OT)                                                                        \
    do {                                                                                                              \
        if (scalar_l) {                                                                                               \
            LT sc;                                                                                                    \
            MA_TRY(fetch_scalar(ctx, a, sizeof(LT), &sc));                                                            \
            return ma_apply_promote_##LTAG##_##RTAG##_scalar_lhs(ctx, sc, (const RT*)b, n, op, mask, mask_off,         \
                                                                 (OT*)out_values, out_validity);                      \
        }                                                                                                             \
        if (scalar_r) {                                                                                               \
            RT sc;                                                                                                    \
            MA_TRY(fetch_scalar(ctx, b, sizeof(RT), &sc));                                                            \
            return ma_apply_promote_##LTAG##_##RTAG##_scalar_rhs(ctx, (const LT*)a, n, sc, op, mask, mask_off,         \
                                                                 (OT*)out_values, out_validity);                      \
        }                                                                                                             \
        return ma_apply_promote_##LTAG##_##RTAG(ctx, (const LT*)a, nl, (const RT*)b, nr, op, mask, mask_off,           \
                                                (OT*)out_values, out_validity);                                       \
    } while (0)

    if (promote) {
        if (pl.code == 'i' && pr.code == 'g') MA_ROUTE_PROMOTE(i32, int32_t, f64, double, double);
        if (pl.code == 'g' && pr.code == 'i') MA_ROUTE_PROMOTE(f64, double, i32, int32_t, double);
        if (pl.code == 'i' && pr.code == 'f') MA_ROUTE_PROMOTE(i32, int32_t, f32, float, float);
        MA_ROUTE_PROMOTE(f32, float, i32, int32_t, float);
    }
#undef MA_ROUTE_PROMOTE

    switch (pl.code) {
        case 'i': MA_ROUTE(int, i32, int32_t);
        case 'I': MA_ROUTE(int, u32, uint32_t);
        case 'l': MA_ROUTE(int, i64, int64_t);
        case 'L': MA_ROUTE(int, u64, uint64_t);
        case 'f': MA_ROUTE(float, f32, float);
        default: MA_ROUTE(float, f64, double);
    }
#undef MA_ROUTE
}

}  // extern "C"

namespace ma {
// the two-mask integer kernels (ma_binary_{i32,u32,i64,u64}.hip): validity = mask1 (&|) mask2, formed in registers
#define MA_DECL_TWO_MASKS(TAG, T)                                                                                    \
    ma_status apply_int_two_masks_##TAG(ma_ctx* ctx, const T* lhs, size_t lhs_len, const T* rhs, size_t rhs_len,    \
                                        int32_t op, const uint8_t* mask1, size_t off1, const uint8_t* mask2,         \
                                        size_t off2, bool combine_and, T* out, uint8_t* out_mask_bits);
MA_DECL_TWO_MASKS(i32, int32_t)
MA_DECL_TWO_MASKS(u32, uint32_t)
MA_DECL_TWO_MASKS(i64, int64_t)
MA_DECL_TWO_MASKS(u64, uint64_t)
#undef MA_DECL_TWO_MASKS
}  // namespace ma

// ------------------------------------------------------------------------------------------------
// route_super_array_broadcast — src/kernels/broadcast/super_array.rs:180-251.
// SuperArray (op) SuperArray, chunk by chunk. The reference's loop is sequential with a literal
// `// TODO: Parallelise` (:193); here ALL chunk pairs run in one launch (descriptor table + per-tile binary search,
// ma_superarray.hip), and a host that owns several contexts (one per GPU) hands each a subset of the chunks.
//   * chunk i: len(lhs_i) != len(rhs_i) -> MA_ERR_LENGTH_MISMATCH ("Super Array broadcasting error", :202-212)
//   * common mask (:215-229): neither has one -> dense kernel; one has -> that one; both -> lhs.union(rhs), i.e.
//     bitwise OR (src/structs/bitmask.rs:661) — NOT the AND of merge_bitmasks_to_new. `null_mask_override`, when
//     given, replaces the common mask of every chunk (:231).
//   * resolve_binary_arithmetic(op, lhs_i, rhs_i, mask) (:236) = the same-type kernels.
// ------------------------------------------------------------------------------------------------
extern "C" ma_status ma_route_super_array_broadcast(ma_ctx* ctx, int32_t format_code, int32_t op, size_t n_chunks,
                                                    const void* const* lhs_data, const size_t* lhs_lens,
                                                    const uint8_t* const* lhs_masks, const void* const* rhs_data,
                                                    const size_t* rhs_lens, const uint8_t* const* rhs_masks,
                                                    const uint8_t* null_mask_override, void* const* out_data,
                                                    uint8_t* const* out_masks, int32_t* out_has_mask) {
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    MA_REQUIRE(n_chunks == 0 || (lhs_data && lhs_lens && rhs_data && rhs_lens && out_data), MA_ERR_INVALID_ARGUMENT,
               "NULL chunk table");
    for (size_t i = 0; i < n_chunks; ++i) {
        if (lhs_lens[i] != rhs_lens[i]) {
            set_error("Super Array broadcasting error - Chunk %zu: LHS %zu RHS %zu", i, lhs_lens[i], rhs_lens[i]);
            return MA_ERR_LENGTH_MISMATCH;
        }
    }
    // One launch for all chunks (ma_superarray.hip). Masked integer Div/Rem/FloorDiv makes the output validity depend on
    // the data — a zero divisor clears the row's bit (simd.rs:319-326) — ...
    {
        const bool is_int = format_code == 'i' || format_code == 'I' || format_code == 'l' || format_code == 'L';
        const bool divlike = op == MA_OP_DIVIDE || op == MA_OP_REMAINDER || op == MA_OP_FLOORDIV;
        bool any_mask = null_mask_override != nullptr;
        for (size_t i = 0; i < n_chunks && !any_mask; ++i)
            any_mask = (lhs_masks && lhs_masks[i]) || (rhs_masks && rhs_masks[i]);
        // ... which the batched kernels do themselves (the computing wave packs the result bits: pair_tile's DV form) as long
        // as every masked chunk's output starts on a 16-byte boundary; only otherwise chunk by chunk
        bool masked_mid_vector = false;
        if (is_int && divlike && any_mask)
            for (size_t i = 0; i < n_chunks && !masked_mid_vector; ++i)
                masked_mid_vector = lhs_lens[i] != 0 && (((uintptr_t)out_data[i]) & 15) != 0 &&
                                    (null_mask_override || (lhs_masks && lhs_masks[i]) || (rhs_masks && rhs_masks[i]));
        if (!(is_int && divlike && any_mask && masked_mid_vector)) {
            MA_REQUIRE(op >= MA_OP_ADD && op <= MA_OP_FLOORDIV, MA_ERR_INVALID_ARGUMENT, "unknown ArithmeticOperator code %d", op);
            return route_batched(ctx, format_code, op, n_chunks, lhs_data, lhs_lens, lhs_masks, rhs_data, rhs_masks,
                                             null_mask_override, out_data, out_masks, out_has_mask);
        }
    }
    // Only reached by masked integer Div / Rem / FloorDiv. Both chunks carrying nulls: lhs.union(rhs) (Bitmask::union, OR)
    // is formed in registers by the two-mask kernels — no merged bitmap, no allocation, nothing to wait for.
    for (size_t i = 0; i < n_chunks; ++i) {
        const size_t n = lhs_lens[i];
        const uint8_t* lm = null_mask_override ? null_mask_override : (lhs_masks ? lhs_masks[i] : nullptr);
        const uint8_t* rm = null_mask_override ? nullptr : (rhs_masks ? rhs_masks[i] : nullptr);
        uint8_t* om = out_masks ? out_masks[i] : nullptr;
        if (out_has_mask) out_has_mask[i] = (lm || rm) ? 1 : 0;
        ma_status st;
        if (lm && rm) {
            switch (format_code) {
                case 'i': st = ma::apply_int_two_masks_i32(ctx, (const int32_t*)lhs_data[i], n, (const int32_t*)rhs_data[i], n, op, lm, 0, rm, 0, false, (int32_t*)out_data[i], om); break;
                case 'I': st = ma::apply_int_two_masks_u32(ctx, (const uint32_t*)lhs_data[i], n, (const uint32_t*)rhs_data[i], n, op, lm, 0, rm, 0, false, (uint32_t*)out_data[i], om); break;
                case 'l': st = ma::apply_int_two_masks_i64(ctx, (const int64_t*)lhs_data[i], n, (const int64_t*)rhs_data[i], n, op, lm, 0, rm, 0, false, (int64_t*)out_data[i], om); break;
                default: st = ma::apply_int_two_masks_u64(ctx, (const uint64_t*)lhs_data[i], n, (const uint64_t*)rhs_data[i], n, op, lm, 0, rm, 0, false, (uint64_t*)out_data[i], om); break;
            }
        } else {
            const uint8_t* mask = lm ? lm : rm;
            switch (format_code) {
                case 'i': st = ma_apply_int_i32(ctx, (const int32_t*)lhs_data[i], n, (const int32_t*)rhs_data[i], n, op, mask, 0, (int32_t*)out_data[i], om); break;
                case 'I': st = ma_apply_int_u32(ctx, (const uint32_t*)lhs_data[i], n, (const uint32_t*)rhs_data[i], n, op, mask, 0, (uint32_t*)out_data[i], om); break;
                case 'l': st = ma_apply_int_i64(ctx, (const int64_t*)lhs_data[i], n, (const int64_t*)rhs_data[i], n, op, mask, 0, (int64_t*)out_data[i], om); break;
                default: st = ma_apply_int_u64(ctx, (const uint64_t*)lhs_data[i], n, (const uint64_t*)rhs_data[i], n, op, mask, 0, (uint64_t*)out_data[i], om); break;
            }
        }
        if (st != MA_OK) return st;
    }
    return MA_OK;
}

// SuperArray (op) Scalar and Scalar (op) SuperArray: the reference maps `broadcast_value(chunk, scalar)` over the chunks
// (src/kernels/broadcast/super_array.rs:87-116, scalar.rs:214-243; the views' twins :120-148, :247-276), each of which ends
// in the array kernels with a length-1 operand (maybe_broadcast_scalar_array). Here all chunks go in one launch (a few for
// very long lists) with the scalar in a kernel argument; a chunk's result carries the chunk's own validity.
extern "C" ma_status ma_broadcast_super_array_scalar(ma_ctx* ctx, int32_t format_code, int32_t op, int32_t scalar_is_lhs,
                                                     const void* scalar, size_t n_chunks, const void* const* chunk_data,
                                                     const size_t* chunk_lens, const uint8_t* const* chunk_masks,
                                                     void* const* out_data, uint8_t* const* out_masks,
                                                     int32_t* out_has_mask) {
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    MA_REQUIRE(scalar != nullptr, MA_ERR_INVALID_ARGUMENT, "scalar is NULL");
    MA_REQUIRE(n_chunks == 0 || (chunk_data && chunk_lens && out_data), MA_ERR_INVALID_ARGUMENT, "NULL chunk table");
    MA_REQUIRE(op >= MA_OP_ADD && op <= MA_OP_FLOORDIV, MA_ERR_INVALID_ARGUMENT, "unknown ArithmeticOperator code %d", op);
    size_t elem = 0;
    switch (format_code) {
        case 'i': case 'I': case 'f': elem = 4; break;
        case 'l': case 'L': case 'g': elem = 8; break;
        default:
            set_error("unsupported element format '%c'", (char)format_code);
            return MA_ERR_UNSUPPORTED;
    }
    uint64_t sbits = 0;
    memcpy(&sbits, scalar, elem);
    const bool is_int = format_code == 'i' || format_code == 'I' || format_code == 'l' || format_code == 'L';
    const bool divlike = op == MA_OP_DIVIDE || op == MA_OP_REMAINDER || op == MA_OP_FLOORDIV;
    bool any_mask = false;
    for (size_t i = 0; i < n_chunks && !any_mask; ++i) any_mask = chunk_masks && chunk_masks[i] && chunk_lens[i];
    bool masked_mid_vector = false;  // a masked chunk whose output starts off a 16-byte boundary (see ma_route_super_array_broadcast)
    if (is_int && divlike && any_mask)
        for (size_t i = 0; i < n_chunks && !masked_mid_vector; ++i)
            masked_mid_vector = chunk_lens[i] != 0 && chunk_masks[i] && (((uintptr_t)out_data[i]) & 15) != 0;
    if (!(is_int && divlike && any_mask && masked_mid_vector)) {
        const int smode = scalar_is_lhs ? 1 : 2;
        return route_batched(ctx, format_code, op, n_chunks, scalar_is_lhs ? nullptr : chunk_data, chunk_lens,
                             scalar_is_lhs ? nullptr : chunk_masks, scalar_is_lhs ? chunk_data : nullptr,
                             scalar_is_lhs ? chunk_masks : nullptr, nullptr, out_data, out_masks, out_has_mask, smode, sbits);
    }
    // masked integer Div / Rem / FloorDiv: the output validity depends on the data (a zero divisor clears the row's bit,
    // simd.rs:319-326) — chunk by chunk through the scalar forms of the array kernels
    for (size_t i = 0; i < n_chunks; ++i) {
        const size_t n = chunk_lens[i];
        const uint8_t* m = chunk_masks ? chunk_masks[i] : nullptr;
        uint8_t* om = out_masks ? out_masks[i] : nullptr;
        if (out_has_mask) out_has_mask[i] = m ? 1 : 0;
        if (n == 0) continue;
        const void* d = chunk_data[i];
        ma_status st;
#define MA_SCALAR_CHUNK(T, TAG)                                                                                            \
    {                                                                                                                     \
        T sv;                                                                                                             \
        memcpy(&sv, scalar, sizeof(T));                                                                                   \
        st = scalar_is_lhs ? ma_apply_int_##TAG##_scalar_lhs(ctx, sv, (const T*)d, n, op, m, 0, (T*)out_data[i], om)      \
                           : ma_apply_int_##TAG##_scalar_rhs(ctx, (const T*)d, n, sv, op, m, 0, (T*)out_data[i], om);     \
    }
        switch (format_code) {
            case 'i': MA_SCALAR_CHUNK(int32_t, i32) break;
            case 'I': MA_SCALAR_CHUNK(uint32_t, u32) break;
            case 'l': MA_SCALAR_CHUNK(int64_t, i64) break;
            default: MA_SCALAR_CHUNK(uint64_t, u64) break;
        }
#undef MA_SCALAR_CHUNK
        if (st != MA_OK) return st;
    }
    return MA_OK;
}
