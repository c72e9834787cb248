// minarrow_hip_routing.hpp — the reference's enum-dispatch layer on the host, above the typed kernels of
// minarrow_hip.hpp. Header-only C++17; mirrors (names, argument order, error behaviour):
//
//   NumericArray                 src/enums/collections/numeric_array.rs:81-99   enum of Arc<IntegerArray/FloatArray>
//   NumericArrayV (ArrayV)       src/structs/views/array_view.rs                (array, offset, len)
//   Scalar                       src/enums/scalar.rs                            numeric alternatives only
//   resolve_binary_arithmetic    src/kernels/routing/arithmetic.rs:214-222
//   arithmetic_dispatch          src/kernels/routing/arithmetic.rs:225-407      type matrix + Int32<->Float promotion
//   maybe_broadcast_scalar_array src/kernels/routing/broadcast.rs:87-112        a length-1 side is broadcast
//   broadcast_array_to_scalar    src/kernels/broadcast/array.rs:139-184
//   broadcast_scalar_to_array    src/kernels/broadcast/scalar.rs:169-210
//   a + b, a - b, a * b, a / b   src/kernels/arithmetic/types.rs:51-57 -> broadcast_value (broadcast/mod.rs:152-169)
//
// What the host does NOT do here, unlike the reference: materialise `vec64![x; n]` for the length-1 side
// (routing/broadcast.rs:30-45) or two casted Vec64s for a promotion (routing/arithmetic.rs:244-269). Both are
// fused into the device kernels (ma_apply_*_scalar_{lhs,rhs}, ma_apply_promote_*); results are identical.
#pragma once

#include <limits>
#include <memory>
#include <variant>

#include "minarrow_hip.hpp"

namespace ma {

enum class NumericType { Int32, Int64, UInt32, UInt64, Float32, Float64 };

inline const char* numeric_type_name(NumericType t) {
    switch (t) {
        case NumericType::Int32: return "Int32";
        case NumericType::Int64: return "Int64";
        case NumericType::UInt32: return "UInt32";
        case NumericType::UInt64: return "UInt64";
        case NumericType::Float32: return "Float32";
        default: return "Float64";
    }
}

// enum NumericArray { Int32(Arc<IntegerArray<i32>>), ... } — cheap to clone, shared ownership.
class NumericArray {
  public:
    using Storage = std::variant<std::shared_ptr<const IntegerArray<int32_t>>, std::shared_ptr<const IntegerArray<int64_t>>,
                                 std::shared_ptr<const IntegerArray<uint32_t>>, std::shared_ptr<const IntegerArray<uint64_t>>,
                                 std::shared_ptr<const FloatArray<float>>, std::shared_ptr<const FloatArray<double>>>;
    Storage v;

    NumericArray() = default;
    // Array::from_int32 ... Array::from_float64 (src/enums/array.rs)
    static NumericArray from_int32(IntegerArray<int32_t> a) { return wrap<0>(std::move(a)); }
    static NumericArray from_int64(IntegerArray<int64_t> a) { return wrap<1>(std::move(a)); }
    static NumericArray from_uint32(IntegerArray<uint32_t> a) { return wrap<2>(std::move(a)); }
    static NumericArray from_uint64(IntegerArray<uint64_t> a) { return wrap<3>(std::move(a)); }
    static NumericArray from_float32(FloatArray<float> a) { return wrap<4>(std::move(a)); }
    static NumericArray from_float64(FloatArray<double> a) { return wrap<5>(std::move(a)); }

    NumericType type() const { return (NumericType)v.index(); }
    size_t len() const {
        return std::visit([](const auto& p) { return p->data.size(); }, v);
    }
    const std::optional<Bitmask>& null_mask() const {
        return std::visit([](const auto& p) -> const std::optional<Bitmask>& { return p->null_mask; }, v);
    }
    // try_i64_ref / try_f64_ref ... (src/enums/array.rs:345,397): nullptr when the variant differs
    const IntegerArray<int32_t>* try_i32_ref() const { return get<0>(); }
    const IntegerArray<int64_t>* try_i64_ref() const { return get<1>(); }
    const IntegerArray<uint32_t>* try_u32_ref() const { return get<2>(); }
    const IntegerArray<uint64_t>* try_u64_ref() const { return get<3>(); }
    const FloatArray<float>* try_f32_ref() const { return get<4>(); }
    const FloatArray<double>* try_f64_ref() const { return get<5>(); }

  private:
    template <size_t I, typename A>
    static NumericArray wrap(A a) {
        NumericArray n;
        n.v.template emplace<I>(std::make_shared<const A>(std::move(a)));
        return n;
    }
    template <size_t I>
    auto get() const -> decltype(std::get<I>(v).get()) {
        return v.index() == I ? std::get<I>(v).get() : nullptr;
    }
};

// ArrayV: a window over a shared array.
struct NumericArrayV {
    NumericArray array;
    size_t offset = 0;
    size_t len_ = 0;
    NumericArrayV() = default;
    NumericArrayV(NumericArray a) : array(std::move(a)), offset(0), len_(array.len()) {}  // impl From<Array> for ArrayV
    NumericArrayV(NumericArray a, size_t off, size_t n) : array(std::move(a)), offset(off), len_(n) {}
    size_t len() const { return len_; }
};

// Scalar — the numeric alternatives broadcast_array_to_scalar accepts (broadcast/array.rs:146-163).
using Scalar = std::variant<int32_t, int64_t, uint32_t, uint64_t, float, double>;

namespace detail {

template <typename T>
struct Family;
#define MA_ROUTING_FAMILY(T, TAG, FAM, ARRAY, WRAP)                                                              \
    template <>                                                                                                  \
    struct Family<T> {                                                                                           \
        using Array = ARRAY<T>;                                                                                  \
        static Array aa(Slice<T> l, Slice<T> r, ArithmeticOperator op, const Bitmask* m) {                       \
            return apply_##FAM##_##TAG(l, r, op, m);                                                             \
        }                                                                                                        \
        static Array as(Slice<T> l, T s, ArithmeticOperator op, const Bitmask* m) {                              \
            return apply_##FAM##_##TAG##_scalar_rhs(l, s, op, m);                                                \
        }                                                                                                        \
        static Array sa(T s, Slice<T> r, ArithmeticOperator op, const Bitmask* m) {                              \
            return apply_##FAM##_##TAG##_scalar_lhs(s, r, op, m);                                                \
        }                                                                                                        \
        static NumericArray wrap(Array a) { return NumericArray::WRAP(std::move(a)); }                           \
    };
MA_ROUTING_FAMILY(int32_t, i32, int, IntegerArray, from_int32)
MA_ROUTING_FAMILY(int64_t, i64, int, IntegerArray, from_int64)
MA_ROUTING_FAMILY(uint32_t, u32, int, IntegerArray, from_uint32)
MA_ROUTING_FAMILY(uint64_t, u64, int, IntegerArray, from_uint64)
MA_ROUTING_FAMILY(float, f32, float, FloatArray, from_float32)
MA_ROUTING_FAMILY(double, f64, float, FloatArray, from_float64)
#undef MA_ROUTING_FAMILY

// Same-type pair. `l1` / `r1`: that side is a length-1 array being broadcast — its value is data[0], as
// broadcast_length_1_array reads it (routing/broadcast.rs:30-45: `a.data[0]`, whatever the view's offset).
template <typename T, typename A>
NumericArray same_type(ArithmeticOperator op, const A& l, size_t lo, size_t ln, bool l1, const A& r, size_t ro, size_t rn,
                       bool r1, const Bitmask* mask) {
    using F = Family<T>;
    if (l1) return F::wrap(F::sa(l.data[0], Slice<T>(r.data.data() + ro, rn), op, mask));
    if (r1) return F::wrap(F::as(Slice<T>(l.data.data() + lo, ln), r.data[0], op, mask));
    return F::wrap(F::aa(Slice<T>(l.data.data() + lo, ln), Slice<T>(r.data.data() + ro, rn), op, mask));
}

template <typename OT>
FloatArray<OT> promoted_out(size_t n, const Bitmask* mask) {
    FloatArray<OT> out;
    out.data = Vec64<OT>::with_capacity(n);
    out.data.set_len(n);
    if (mask) out.null_mask = Bitmask::new_set_all(n, true);
    return out;
}

#define MA_ROUTING_PROMOTE(NAME, LT, RT, OT, LTAG, RTAG)                                                             \
    inline FloatArray<OT> NAME(ArithmeticOperator op, const LT* l, size_t ln, bool l1, const RT* r, size_t rn, bool r1,  \
                               const Bitmask* mask) {                                                                \
        const size_t n = l1 ? rn : ln;                                                                               \
        FloatArray<OT> out = promoted_out<OT>(n, mask);                                                              \
        const uint8_t* mb = mask ? mask->bits.data() : nullptr;                                                      \
        uint8_t* ob = mask ? out.null_mask->bits.data() : nullptr;                                                   \
        ma_ctx* ctx = Context::global().get();                                                                       \
        if (l1) check(ma_apply_promote_##LTAG##_##RTAG##_scalar_lhs(ctx, l[0], r, rn, (int32_t)op, mb, 0, out.data.data(), ob)); \
        else if (r1) check(ma_apply_promote_##LTAG##_##RTAG##_scalar_rhs(ctx, l, ln, r[0], (int32_t)op, mb, 0, out.data.data(), ob)); \
        else check(ma_apply_promote_##LTAG##_##RTAG(ctx, l, ln, r, rn, (int32_t)op, mb, 0, out.data.data(), ob));    \
        return out;                                                                                                  \
    }
MA_ROUTING_PROMOTE(promote_i32_f64, int32_t, double, double, i32, f64)
MA_ROUTING_PROMOTE(promote_f64_i32, double, int32_t, double, f64, i32)
MA_ROUTING_PROMOTE(promote_i32_f32, int32_t, float, float, i32, f32)
MA_ROUTING_PROMOTE(promote_f32_i32, float, int32_t, float, f32, i32)
#undef MA_ROUTING_PROMOTE

}  // namespace detail

// resolve_binary_arithmetic(op, lhs, rhs, null_mask) — routing/arithmetic.rs:214-222.
//   1. maybe_broadcast_scalar_array: equal lengths pass; exactly one side of length 1 is broadcast; otherwise
//      LengthMismatch("cannot broadcast arrays of length {l} and {r}")            (routing/broadcast.rs:87-112)
//   2. arithmetic_dispatch: same-type pairs of Int32/Int64/UInt32/UInt64/Float32/Float64 (:278-339);
//      Int32 with Float64 or Float32 in either order is promoted to the float type (:342-373);
//      anything else: UnsupportedType("Unsupported array type combination for arithmetic operations") (:403-405)
//   The views are sliced [offset, offset+len) (:273-285); `null_mask` is the CALLER's mask, read from bit 0 —
//   the arrays' own null masks are not consulted (:214-222 passes it straight down).
inline NumericArray resolve_binary_arithmetic(ArithmeticOperator op, const NumericArrayV& lhs, const NumericArrayV& rhs,
                                              const Bitmask* null_mask = nullptr) {
    const size_t l = lhs.len(), r = rhs.len();
    const bool l1 = l != r && l == 1, r1 = l != r && r == 1;
    if (l != r && !l1 && !r1)
        throw KernelError(KernelError::LengthMismatch,
                          "cannot broadcast arrays of length " + std::to_string(l) + " and " + std::to_string(r));
    const NumericArray& a = lhs.array;
    const NumericArray& b = rhs.array;
    // a broadcast side becomes ArrayV::new(materialised, 0, other.len()) in the reference: its view offset is dropped
    const size_t lo = lhs.offset, ro = rhs.offset;
    using NT = NumericType;
    if (a.type() == b.type()) {
        switch (a.type()) {
            case NT::Int32: return detail::same_type<int32_t>(op, *a.try_i32_ref(), lo, l, l1, *b.try_i32_ref(), ro, r, r1, null_mask);
            case NT::Int64: return detail::same_type<int64_t>(op, *a.try_i64_ref(), lo, l, l1, *b.try_i64_ref(), ro, r, r1, null_mask);
            case NT::UInt32: return detail::same_type<uint32_t>(op, *a.try_u32_ref(), lo, l, l1, *b.try_u32_ref(), ro, r, r1, null_mask);
            case NT::UInt64: return detail::same_type<uint64_t>(op, *a.try_u64_ref(), lo, l, l1, *b.try_u64_ref(), ro, r, r1, null_mask);
            case NT::Float32: return detail::same_type<float>(op, *a.try_f32_ref(), lo, l, l1, *b.try_f32_ref(), ro, r, r1, null_mask);
            default: return detail::same_type<double>(op, *a.try_f64_ref(), lo, l, l1, *b.try_f64_ref(), ro, r, r1, null_mask);
        }
    }
    // a length-1 side is read at data[0]; a full side at its view offset
    auto at = [](const auto* arr, size_t off, bool one) { return arr->data.data() + (one ? 0 : off); };
    if (a.type() == NT::Int32 && b.type() == NT::Float64)
        return NumericArray::from_float64(detail::promote_i32_f64(op, at(a.try_i32_ref(), lo, l1), l, l1, at(b.try_f64_ref(), ro, r1), r, r1, null_mask));
    if (a.type() == NT::Float64 && b.type() == NT::Int32)
        return NumericArray::from_float64(detail::promote_f64_i32(op, at(a.try_f64_ref(), lo, l1), l, l1, at(b.try_i32_ref(), ro, r1), r, r1, null_mask));
    if (a.type() == NT::Int32 && b.type() == NT::Float32)
        return NumericArray::from_float32(detail::promote_i32_f32(op, at(a.try_i32_ref(), lo, l1), l, l1, at(b.try_f32_ref(), ro, r1), r, r1, null_mask));
    if (a.type() == NT::Float32 && b.type() == NT::Int32)
        return NumericArray::from_float32(detail::promote_f32_i32(op, at(a.try_f32_ref(), lo, l1), l, l1, at(b.try_i32_ref(), ro, r1), r, r1, null_mask));
    throw KernelError(KernelError::UnsupportedType, "Unsupported array type combination for arithmetic operations");
}

namespace detail {
inline NumericArray scalar_array(const Scalar& s) {  // IntegerArray::from_slice(&[*val]) — broadcast/array.rs:146-163
    return std::visit(
        [](auto val) -> NumericArray {
            using T = decltype(val);
            typename Family<T>::Array one;
            one.data = Vec64<T>{val};
            return Family<T>::wrap(std::move(one));
        },
        s);
}
}  // namespace detail

// Array (op) Scalar — broadcast/array.rs:139-184: the scalar becomes a 1-element array of ITS OWN type, then the
// ordinary routing applies (so Int64 array with a Float64 scalar is UnsupportedType, Int32 with Float64 promotes).
inline NumericArray broadcast_array_to_scalar(ArithmeticOperator op, const NumericArray& array, const Scalar& scalar) {
    return resolve_binary_arithmetic(op, NumericArrayV(array), NumericArrayV(detail::scalar_array(scalar)), nullptr);
}
// Scalar (op) Array — broadcast/scalar.rs:169-210
inline NumericArray broadcast_scalar_to_array(ArithmeticOperator op, const Scalar& scalar, const NumericArray& array) {
    return resolve_binary_arithmetic(op, NumericArrayV(detail::scalar_array(scalar)), NumericArrayV(array), nullptr);
}

// Value + Value — src/kernels/arithmetic/types.rs:51-57 -> broadcast_value(op, l, r) (broadcast/mod.rs:152-169);
// the (Array, Array) arm passes null_mask = None.
inline NumericArray operator+(const NumericArray& l, const NumericArray& r) { return resolve_binary_arithmetic(ArithmeticOperator::Add, l, r); }
inline NumericArray operator-(const NumericArray& l, const NumericArray& r) { return resolve_binary_arithmetic(ArithmeticOperator::Subtract, l, r); }
inline NumericArray operator*(const NumericArray& l, const NumericArray& r) { return resolve_binary_arithmetic(ArithmeticOperator::Multiply, l, r); }
inline NumericArray operator/(const NumericArray& l, const NumericArray& r) { return resolve_binary_arithmetic(ArithmeticOperator::Divide, l, r); }
inline NumericArray operator%(const NumericArray& l, const NumericArray& r) { return resolve_binary_arithmetic(ArithmeticOperator::Remainder, l, r); }
inline NumericArray operator+(const NumericArray& l, const Scalar& r) { return broadcast_array_to_scalar(ArithmeticOperator::Add, l, r); }
inline NumericArray operator-(const NumericArray& l, const Scalar& r) { return broadcast_array_to_scalar(ArithmeticOperator::Subtract, l, r); }
inline NumericArray operator*(const NumericArray& l, const Scalar& r) { return broadcast_array_to_scalar(ArithmeticOperator::Multiply, l, r); }
inline NumericArray operator/(const NumericArray& l, const Scalar& r) { return broadcast_array_to_scalar(ArithmeticOperator::Divide, l, r); }
inline NumericArray operator+(const Scalar& l, const NumericArray& r) { return broadcast_scalar_to_array(ArithmeticOperator::Add, l, r); }
inline NumericArray operator-(const Scalar& l, const NumericArray& r) { return broadcast_scalar_to_array(ArithmeticOperator::Subtract, l, r); }
inline NumericArray operator*(const Scalar& l, const NumericArray& r) { return broadcast_scalar_to_array(ArithmeticOperator::Multiply, l, r); }
inline NumericArray operator/(const Scalar& l, const NumericArray& r) { return broadcast_scalar_to_array(ArithmeticOperator::Divide, l, r); }

// ---- Table: named columns of equal length (src/structs/table.rs) — only what the broadcast layer touches ----------
struct FieldArray {
    std::string name;
    NumericArray array;
};
struct Table {
    std::vector<FieldArray> cols;
    std::string name;
    size_t n_cols() const { return cols.size(); }
    size_t n_rows() const { return cols.empty() ? 0 : cols[0].array.len(); }
};

// broadcast_table_with_operator — src/kernels/broadcast/table.rs:31-63: column counts must match (ShapeError
// "Table column count mismatch: {} vs {}"), column i = resolve_binary_arithmetic(op, l.cols[i], r.cols[i], None)
// under the LEFT table's field (:55-57); the result carries the left table's name (:62).
inline Table broadcast_table_with_operator(ArithmeticOperator op, const Table& l, const Table& r) {
    if (l.n_cols() != r.n_cols())
        throw KernelError(KernelError::Broadcasting, "Table column count mismatch: " + std::to_string(l.n_cols()) + " vs " +
                                                         std::to_string(r.n_cols()));
    Table out;
    out.name = l.name;
    for (size_t i = 0; i < l.n_cols(); ++i)
        out.cols.push_back({l.cols[i].name, resolve_binary_arithmetic(op, l.cols[i].array, r.cols[i].array, nullptr)});
    return out;
}
// broadcast_table_add — table.rs:69-127: additionally checks the row counts ("Table row count mismatch: LHS {} rows,
// RHS {} rows") and passes the caller's null mask to every column.
inline Table broadcast_table_add(const Table& l, const Table& r, const Bitmask* null_mask = nullptr) {
    if (l.n_cols() != r.n_cols())
        throw KernelError(KernelError::Broadcasting, "Table column count mismatch: LHS " + std::to_string(l.n_cols()) +
                                                         " cols, RHS " + std::to_string(r.n_cols()) + " cols");
    if (l.n_rows() != r.n_rows())
        throw KernelError(KernelError::Broadcasting, "Table row count mismatch: LHS " + std::to_string(l.n_rows()) +
                                                         " rows, RHS " + std::to_string(r.n_rows()) + " rows");
    Table out;
    out.name = l.name;
    for (size_t i = 0; i < l.n_cols(); ++i)
        out.cols.push_back({l.cols[i].name, resolve_binary_arithmetic(ArithmeticOperator::Add, l.cols[i].array, r.cols[i].array, null_mask)});
    return out;
}
// broadcast_array_to_table — broadcast/array.rs:187-235: array (op) column, for every column.
inline Table broadcast_array_to_table(ArithmeticOperator op, const NumericArray& array, const Table& table) {
    Table out;
    out.name = table.name;
    for (const FieldArray& c : table.cols) out.cols.push_back({c.name, resolve_binary_arithmetic(op, array, c.array, nullptr)});
    return out;
}
// broadcast_table_to_array — table.rs:179-228: column (op) array.
inline Table broadcast_table_to_array(ArithmeticOperator op, const Table& table, const NumericArray& array) {
    Table out;
    out.name = table.name;
    for (const FieldArray& c : table.cols) out.cols.push_back({c.name, resolve_binary_arithmetic(op, c.array, array, nullptr)});
    return out;
}
// broadcast_table_to_scalar — table.rs:230-279: column (op) scalar.
inline Table broadcast_table_to_scalar(ArithmeticOperator op, const Table& table, const Scalar& scalar) {
    Table out;
    out.name = table.name;
    for (const FieldArray& c : table.cols) out.cols.push_back({c.name, broadcast_array_to_scalar(op, c.array, scalar)});
    return out;
}

// ---- chunked containers (src/structs/chunked/super_array.rs, super_table.rs) ----------------------------------------
struct SuperArray {
    std::vector<NumericArray> chunks_;
    SuperArray() = default;
    explicit SuperArray(std::vector<NumericArray> c) : chunks_(std::move(c)) {}
    void push(NumericArray a) { chunks_.push_back(std::move(a)); }
    const std::vector<NumericArray>& chunks() const { return chunks_; }
    size_t n_chunks() const { return chunks_.size(); }
    size_t len() const {
        size_t n = 0;
        for (const NumericArray& c : chunks_) n += c.len();
        return n;
    }
};

namespace detail {
inline const void* chunk_data(const NumericArray& a) {
    return std::visit([](const auto& p) -> const void* { return p->data.data(); }, a.v);
}
inline int32_t format_code(NumericType t) {
    switch (t) {
        case NumericType::Int32: return 'i';
        case NumericType::Int64: return 'l';
        case NumericType::UInt32: return 'I';
        case NumericType::UInt64: return 'L';
        case NumericType::Float32: return 'f';
        default: return 'g';
    }
}
inline size_t elem_size(NumericType t) {
    return (t == NumericType::Int64 || t == NumericType::UInt64 || t == NumericType::Float64) ? 8 : 4;
}
template <typename T>
NumericArray make_chunk(size_t n, bool with_mask) {
    typename Family<T>::Array a;
    a.data = Vec64<T>::with_capacity(n);
    a.data.set_len(n);
    if (with_mask) a.null_mask = Bitmask::new_set_all(n, true);
    return Family<T>::wrap(std::move(a));
}
inline NumericArray make_chunk_of(NumericType t, size_t n, bool with_mask) {
    switch (t) {
        case NumericType::Int32: return make_chunk<int32_t>(n, with_mask);
        case NumericType::Int64: return make_chunk<int64_t>(n, with_mask);
        case NumericType::UInt32: return make_chunk<uint32_t>(n, with_mask);
        case NumericType::UInt64: return make_chunk<uint64_t>(n, with_mask);
        case NumericType::Float32: return make_chunk<float>(n, with_mask);
        default: return make_chunk<double>(n, with_mask);
    }
}
inline void* mutable_data(const NumericArray& a) {  // freshly made, uniquely owned output chunks only
    return const_cast<void*>(std::visit([](const auto& p) -> const void* { return p->data.data(); }, a.v));
}
inline uint8_t* mutable_mask(const NumericArray& a) {
    const std::optional<Bitmask>& m = a.null_mask();
    return m ? const_cast<uint8_t*>(m->bits.data()) : nullptr;
}
}  // namespace detail

// route_super_array_broadcast — src/kernels/broadcast/super_array.rs:180-251. Chunk i of the result is
// resolve_binary_arithmetic(op, lhs_i, rhs_i, mask_i) where mask_i = null_mask_override, or the common mask of the
// chunks' OWN null masks: none / the one present / lhs.union(rhs) = bitwise OR (:215-229). Chunk lengths must agree
// pairwise (ShapeError "Super Array broadcasting error ..."). The reference loops sequentially ("// TODO: Parallelise",
// :193); when every chunk pair has one common element type ALL pairs run in one launch (ma_route_super_array_broadcast).
inline SuperArray route_super_array_broadcast(ArithmeticOperator op, const SuperArray& lhs, const SuperArray& rhs,
                                              const Bitmask* null_mask_override = nullptr) {
    const size_t k = lhs.n_chunks();
    if (rhs.n_chunks() < k)
        throw KernelError(KernelError::Broadcasting, "Super Array broadcasting error - RHS has fewer chunks than LHS");
    bool one_type = k > 0;
    for (size_t i = 0; i < k; ++i) {
        if (lhs.chunks()[i].len() != rhs.chunks()[i].len())
            throw KernelError(KernelError::Broadcasting,
                              "Super Array broadcasting error - Chunk: LHS " + std::to_string(lhs.chunks()[i].len()) + " RHS " +
                                  std::to_string(rhs.chunks()[i].len()));
        one_type = one_type && lhs.chunks()[i].type() == lhs.chunks()[0].type() && rhs.chunks()[i].type() == lhs.chunks()[0].type();
    }
    SuperArray out;
    if (k == 0) return out;
    if (!one_type) {  // mixed element types: chunk by chunk through the type matrix (promotions, UnsupportedType)
        for (size_t i = 0; i < k; ++i) {
            const NumericArray &l = lhs.chunks()[i], &r = rhs.chunks()[i];
            const Bitmask* m = null_mask_override;
            Bitmask common;
            if (!m) {
                const std::optional<Bitmask>&lm = l.null_mask(), &rm = r.null_mask();
                if (lm && rm) {
                    common = or_masks(window(*lm), window(*rm));
                    m = &common;
                } else if (lm) {
                    m = &*lm;
                } else if (rm) {
                    m = &*rm;
                }
            }
            out.push(resolve_binary_arithmetic(op, l, r, m));
        }
        return out;
    }
    const NumericType t = lhs.chunks()[0].type();
    std::vector<const void*> ld(k), rd(k);
    std::vector<size_t> ll(k), rl(k);
    std::vector<const uint8_t*> lm(k), rm(k);
    std::vector<void*> od(k);
    std::vector<uint8_t*> om(k);
    std::vector<int32_t> has(k);
    for (size_t i = 0; i < k; ++i) {
        const NumericArray &l = lhs.chunks()[i], &r = rhs.chunks()[i];
        ld[i] = detail::chunk_data(l);
        rd[i] = detail::chunk_data(r);
        ll[i] = l.len();
        rl[i] = r.len();
        lm[i] = l.null_mask() ? l.null_mask()->bits.data() : nullptr;
        rm[i] = r.null_mask() ? r.null_mask()->bits.data() : nullptr;
        out.push(detail::make_chunk_of(t, ll[i], null_mask_override || lm[i] || rm[i]));
        od[i] = detail::mutable_data(out.chunks()[i]);
        om[i] = detail::mutable_mask(out.chunks()[i]);
    }
    check(ma_route_super_array_broadcast(Context::global().get(), detail::format_code(t), (int32_t)op, k, ld.data(), ll.data(),
                                         lm.data(), rd.data(), rl.data(), rm.data(),
                                         null_mask_override ? null_mask_override->bits.data() : nullptr, od.data(), om.data(),
                                         has.data()));
    return out;
}

// SuperArray (op) Scalar / Scalar (op) SuperArray — src/kernels/broadcast/super_array.rs:87-116, scalar.rs:214-243: the
// reference maps broadcast_value(chunk, scalar) over the chunks, i.e. broadcast_array_to_scalar per chunk — the scalar as a
// 1-element array of its own type through the ordinary routing with NO null mask (array.rs:183 passes None: the chunks'
// own validity is not consulted and the result chunks are dense). When the scalar has the chunks' element type ALL chunks
// run in one launch (ma_broadcast_super_array_scalar); other combinations go chunk by chunk through the type matrix
// (Int32 with a float scalar promotes, Int64 with Float64 is UnsupportedType).
namespace detail {
inline SuperArray super_array_scalar(ArithmeticOperator op, const SuperArray& sa, const Scalar& scalar, bool scalar_is_lhs) {
    SuperArray out;
    const size_t k = sa.n_chunks();
    if (k == 0) return out;
    const NumericArray one = scalar_array(scalar);
    bool one_type = true;
    for (size_t i = 0; i < k; ++i) one_type = one_type && sa.chunks()[i].type() == one.type();
    if (!one_type) {
        for (size_t i = 0; i < k; ++i)
            out.push(scalar_is_lhs ? broadcast_scalar_to_array(op, scalar, sa.chunks()[i])
                                   : broadcast_array_to_scalar(op, sa.chunks()[i], scalar));
        return out;
    }
    const NumericType t = one.type();
    std::vector<const void*> cd(k);
    std::vector<size_t> cl(k);
    std::vector<void*> od(k);
    for (size_t i = 0; i < k; ++i) {
        cd[i] = chunk_data(sa.chunks()[i]);
        cl[i] = sa.chunks()[i].len();
        out.push(make_chunk_of(t, cl[i], false));
        od[i] = mutable_data(out.chunks()[i]);
    }
    check(ma_broadcast_super_array_scalar(Context::global().get(), format_code(t), (int32_t)op, scalar_is_lhs ? 1 : 0,
                                          chunk_data(one), k, cd.data(), cl.data(), nullptr, od.data(), nullptr, nullptr));
    return out;
}
}  // namespace detail
inline SuperArray broadcast_superarray_to_scalar(ArithmeticOperator op, const SuperArray& super_array, const Scalar& scalar) {
    return detail::super_array_scalar(op, super_array, scalar, false);
}
inline SuperArray broadcast_scalar_to_superarray(ArithmeticOperator op, const Scalar& scalar, const SuperArray& super_array) {
    return detail::super_array_scalar(op, super_array, scalar, true);
}

// ArrayView (op) SuperArray / SuperArray (op) ArrayView — src/kernels/broadcast/super_array.rs:255-363 (the SuperArrayV twins
// :367-475 walk the view's slices the same way): the view must be as long as the SuperArray (ShapeError), chunk i meets the
// view's window [offset_i, offset_i + len_i), each pair goes through broadcast_value = resolve_binary_arithmetic with no
// null mask (mod.rs:191-198) — dense result chunks. With one element type everywhere the windows become a pointer table
// and ALL pairs run in one launch (ma_route_super_array_broadcast without validity); otherwise chunk by chunk.
namespace detail {
inline SuperArray arrayview_super_array(ArithmeticOperator op, const NumericArrayV& view, const SuperArray& sa, bool view_is_lhs) {
    if (view.len() != sa.len())
        throw KernelError(KernelError::Broadcasting,
                          view_is_lhs ? "ArrayView length (" + std::to_string(view.len()) + ") does not match SuperArray length (" +
                                            std::to_string(sa.len()) + ")"
                                      : "SuperArray length (" + std::to_string(sa.len()) + ") does not match ArrayView length (" +
                                            std::to_string(view.len()) + ")");
    SuperArray out;
    const size_t k = sa.n_chunks();
    if (k == 0) return out;
    bool one_type = true;
    for (size_t i = 0; i < k; ++i) one_type = one_type && sa.chunks()[i].type() == view.array.type();
    if (!one_type) {
        size_t off = 0;
        for (size_t i = 0; i < k; ++i) {
            const NumericArray& c = sa.chunks()[i];
            const NumericArrayV window(view.array, view.offset + off, c.len());
            out.push(view_is_lhs ? resolve_binary_arithmetic(op, window, NumericArrayV(c), nullptr)
                                 : resolve_binary_arithmetic(op, NumericArrayV(c), window, nullptr));
            off += c.len();
        }
        return out;
    }
    const NumericType t = view.array.type();
    const size_t elem = (t == NumericType::Int32 || t == NumericType::UInt32 || t == NumericType::Float32) ? 4 : 8;
    const char* base = static_cast<const char*>(chunk_data(view.array)) + view.offset * elem;
    std::vector<const void*> vd(k), cd(k);
    std::vector<size_t> cl(k);
    std::vector<void*> od(k);
    size_t off = 0;
    for (size_t i = 0; i < k; ++i) {
        cl[i] = sa.chunks()[i].len();
        vd[i] = base + off * elem;
        cd[i] = chunk_data(sa.chunks()[i]);
        out.push(make_chunk_of(t, cl[i], false));
        od[i] = mutable_data(out.chunks()[i]);
        off += cl[i];
    }
    check(ma_route_super_array_broadcast(Context::global().get(), format_code(t), (int32_t)op, k,
                                         view_is_lhs ? vd.data() : cd.data(), cl.data(), nullptr,
                                         view_is_lhs ? cd.data() : vd.data(), cl.data(), nullptr, nullptr, od.data(), nullptr, nullptr));
    return out;
}
}  // namespace detail
inline SuperArray broadcast_arrayview_to_superarray(ArithmeticOperator op, const NumericArrayV& array_view, const SuperArray& super_array) {
    return detail::arrayview_super_array(op, array_view, super_array, true);
}
inline SuperArray broadcast_superarray_to_arrayview(ArithmeticOperator op, const SuperArray& super_array, const NumericArrayV& array_view) {
    return detail::arrayview_super_array(op, array_view, super_array, false);
}

// Consolidate for a chunked column — src/traits/consolidate.rs:110-207: values concatenated in chunk order; the result
// has a null mask iff any chunk has one, chunks without one contribute all-valid rows (:80-105).
inline NumericArray consolidate(const SuperArray& sa) {
    if (sa.n_chunks() == 0) throw Panic("consolidate() called on empty SuperTable");
    const NumericType t = sa.chunks()[0].type();
    const size_t k = sa.n_chunks();
    std::vector<const void*> data(k);
    std::vector<size_t> lens(k);
    std::vector<const uint8_t*> masks(k);
    bool any_mask = false;
    for (size_t i = 0; i < k; ++i) {
        const NumericArray& c = sa.chunks()[i];
        if (c.type() != t) throw KernelError(KernelError::UnsupportedType, "consolidate: chunks of one column must share a type");
        data[i] = detail::chunk_data(c);
        lens[i] = c.len();
        masks[i] = c.null_mask() ? c.null_mask()->bits.data() : nullptr;
        any_mask = any_mask || masks[i];
    }
    NumericArray out = detail::make_chunk_of(t, sa.len(), any_mask);
    int32_t has = 0;
    check(ma_consolidate_column(Context::global().get(), detail::elem_size(t), k, data.data(), lens.data(), masks.data(), nullptr,
                                detail::mutable_data(out), detail::mutable_mask(out), &has));
    return out;
}

// SuperTable: batches with one schema (src/structs/chunked/super_table.rs:78-83). consolidate() — :657-743: one
// contiguous Table; column c = the consolidation of column c of every batch; the name is kept (:1646-1655).
struct SuperTable {
    std::vector<std::shared_ptr<const Table>> batches;
    std::string name;
    size_t n_rows() const {
        size_t n = 0;
        for (const auto& b : batches) n += b->n_rows();
        return n;
    }
};
inline Table consolidate(const SuperTable& st) {
    if (st.batches.empty()) throw Panic("consolidate() called on empty SuperTable");  // super_table.rs:693-696
    Table out;
    out.name = st.name;
    const size_t n_cols = st.batches[0]->n_cols();
    for (size_t c = 0; c < n_cols; ++c) {
        SuperArray col;
        for (const auto& b : st.batches) {
            if (b->n_cols() != n_cols) throw KernelError(KernelError::Broadcasting, "SuperTable batches disagree on the column count");
            col.push(b->cols[c].array);
        }
        out.cols.push_back({st.batches[0]->cols[c].name, consolidate(col)});
    }
    return out;
}

// consolidate_arena — src/structs/chunked/super_table.rs:727-743 -> consolidate_tables_arena (src/structs/arena.rs:1187-1340):
// ONE 64-byte aligned allocation holds every column's values and validity (layout: ma_arena_layout), written by one
// ma_consolidate_table_arena call; the columns of the returned Table are windows into it that keep it alive
// (ArenaRegion::to_buffer / to_bitmask, arena.rs:502-534) and copy themselves out on their first mutation.
namespace detail {
template <typename T, typename A>
inline NumericArray arena_column(const std::shared_ptr<void>& arena, size_t data_off, size_t mask_off, size_t n_rows,
                                 NumericArray (*wrap)(A)) {
    A a;
    char* base = static_cast<char*>(arena.get());
    a.data = Vec64<T>::from_shared(arena, reinterpret_cast<T*>(base + data_off), n_rows);
    if (mask_off != SIZE_MAX) {
        Bitmask m;
        m.len = n_rows;
        m.bits = Vec64<uint8_t>::from_shared(arena, reinterpret_cast<uint8_t*>(base + mask_off), ((n_rows + 63) / 64) * 8);
        a.null_mask = std::move(m);
    }
    return wrap(std::move(a));
}
}  // namespace detail
inline Table consolidate_arena(const SuperTable& st) {
    if (st.batches.empty()) throw Panic("consolidate called on empty table set");  // arena.rs:1196
    const size_t n_cols = st.batches[0]->n_cols(), n_batches = st.batches.size();
    std::vector<size_t> elem(n_cols), rows(n_batches), data_off(n_cols), mask_off(n_cols);
    std::vector<const void*> cells(n_cols * n_batches);
    std::vector<const uint8_t*> masks(n_cols * n_batches);
    std::vector<int32_t> has_nulls(n_cols, 0);
    for (size_t b = 0; b < n_batches; ++b) {
        if (st.batches[b]->n_cols() != n_cols) throw KernelError(KernelError::Broadcasting, "SuperTable batches disagree on the column count");
        rows[b] = st.batches[b]->n_rows();
    }
    for (size_t c = 0; c < n_cols; ++c) {
        const NumericType t = st.batches[0]->cols[c].array.type();
        elem[c] = detail::elem_size(t);
        for (size_t b = 0; b < n_batches; ++b) {
            const NumericArray& a = st.batches[b]->cols[c].array;
            if (a.type() != t) throw KernelError(KernelError::UnsupportedType, "consolidate: batches of one column must share a type");
            cells[c * n_batches + b] = detail::chunk_data(a);
            masks[c * n_batches + b] = a.null_mask() ? a.null_mask()->bits.data() : nullptr;
            if (masks[c * n_batches + b]) has_nulls[c] = 1;
        }
    }
    size_t capacity = 0, used = 0;
    check(ma_arena_layout(n_cols, elem.data(), has_nulls.data(), st.n_rows(), nullptr, nullptr, &capacity, &used));
    void* raw = nullptr;
    check(ma_alloc64_pinned(capacity ? capacity : 64, &raw));
    std::shared_ptr<void> arena(raw, [](void* p) { (void)ma_free_pinned(p); });
    check(ma_consolidate_table_arena(Context::global().get(), n_cols, n_batches, elem.data(), rows.data(), cells.data(), masks.data(),
                                     nullptr, raw, capacity, data_off.data(), mask_off.data(), &used));
    Table out;
    out.name = st.name;
    const size_t n = st.n_rows();
    for (size_t c = 0; c < n_cols; ++c) {
        NumericArray col;
        switch (st.batches[0]->cols[c].array.type()) {
            case NumericType::Int32: col = detail::arena_column<int32_t>(arena, data_off[c], mask_off[c], n, &NumericArray::from_int32); break;
            case NumericType::Int64: col = detail::arena_column<int64_t>(arena, data_off[c], mask_off[c], n, &NumericArray::from_int64); break;
            case NumericType::UInt32: col = detail::arena_column<uint32_t>(arena, data_off[c], mask_off[c], n, &NumericArray::from_uint32); break;
            case NumericType::UInt64: col = detail::arena_column<uint64_t>(arena, data_off[c], mask_off[c], n, &NumericArray::from_uint64); break;
            case NumericType::Float32: col = detail::arena_column<float>(arena, data_off[c], mask_off[c], n, &NumericArray::from_float32); break;
            default: col = detail::arena_column<double>(arena, data_off[c], mask_off[c], n, &NumericArray::from_float64); break;
        }
        out.cols.push_back({st.batches[0]->cols[c].name, std::move(col)});
    }
    return out;
}

// Aggregates over a view, with the hand-off shape of NumericArrayV::guarantee_f64
// (src/structs/views/collections/numeric_array_view.rs:302-317): values advanced to the window, the array's OWN
// un-windowed validity plus the view offset as bit offset. Integer variants: wrapping 64-bit sum converted to f64.
struct Aggregate {
    double sum;
    uint64_t valid_count;
    double mean() const { return valid_count ? sum / (double)valid_count : std::numeric_limits<double>::quiet_NaN(); }
};
inline Aggregate sum(const NumericArrayV& v) {
    ma_ctx* ctx = Context::global().get();
    const std::optional<Bitmask>& m = v.array.null_mask();
    const uint8_t* bits = m ? m->bits.data() : nullptr;
    Aggregate out{0.0, 0};
    switch (v.array.type()) {
        case NumericType::Int32: { int64_t s = 0; check(ma_i32_sum(ctx, v.array.try_i32_ref()->data.data() + v.offset, v.len(), bits, v.offset, -1, &s, &out.valid_count)); out.sum = (double)s; break; }
        case NumericType::Int64: { int64_t s = 0; check(ma_i64_sum(ctx, v.array.try_i64_ref()->data.data() + v.offset, v.len(), bits, v.offset, -1, &s, &out.valid_count)); out.sum = (double)s; break; }
        case NumericType::UInt32: { uint64_t s = 0; check(ma_u32_sum(ctx, v.array.try_u32_ref()->data.data() + v.offset, v.len(), bits, v.offset, -1, &s, &out.valid_count)); out.sum = (double)s; break; }
        case NumericType::UInt64: { uint64_t s = 0; check(ma_u64_sum(ctx, v.array.try_u64_ref()->data.data() + v.offset, v.len(), bits, v.offset, -1, &s, &out.valid_count)); out.sum = (double)s; break; }
        case NumericType::Float32: check(ma_f32_sum(ctx, v.array.try_f32_ref()->data.data() + v.offset, v.len(), bits, v.offset, -1, &out.sum, &out.valid_count)); break;
        default: check(ma_f64_sum(ctx, v.array.try_f64_ref()->data.data() + v.offset, v.len(), bits, v.offset, -1, &out.sum, &out.valid_count)); break;
    }
    return out;
}

}  // namespace ma
