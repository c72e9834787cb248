// ref_suite.cpp — the reference's kernel tests, replayed in C++ through the typed host mirror
// (include/minarrow_hip.hpp) on the GPU. Each function names the Rust test it restates:
//   src/kernels/arithmetic/mod.rs:117-537   (int_kernel_suite!, float_kernel_suite!, fma_*, merge_masks_correctness,
//                                            test_int_dense_power_short_vs_long_input_simd)
//   src/kernels/bitmask/simd.rs:797-955     (simd_bitmask_suite!)
//   benches/hotloop_benchmark_std.rs:49-57  (sum of 0..N)
// Exit code 0 = every assertion held. Run by tests/test_gpu_cpp_host.py.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <functional>
#include <string>
#include <vector>

#include "minarrow_hip.hpp"
#include "minarrow_hip_routing.hpp"
#include "minarrow_hip_parallel.hpp"
#include "minarrow_hip_testing.h"  // the fault hooks (live: the suite is started with MINARROW_HIP_TEST_HOOKS=1)

using namespace ma;
using Op = ArithmeticOperator;

static int g_failed = 0, g_checked = 0;
#define ASSERT(cond)                                                              \
    do {                                                                          \
        ++g_checked;                                                              \
        if (!(cond)) {                                                            \
            ++g_failed;                                                           \
            std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond);           \
        }                                                                         \
    } while (0)

template <typename F>
static bool panics(F f) {  // std::panic::catch_unwind(...).is_err()
    try {
        f();
    } catch (const Panic&) {
        return true;
    }
    return false;
}

template <typename T>
static void assert_int(const IntegerArray<T>& arr, const std::vector<T>& values, const std::vector<bool>* valid) {
    ASSERT(arr.data == values);
    if (valid) {
        ASSERT(arr.null_mask.has_value());
        if (arr.null_mask)
            for (size_t i = 0; i < valid->size(); ++i) ASSERT(arr.null_mask->get(i) == (*valid)[i]);
    } else if (arr.null_mask) {
        ASSERT(arr.null_mask->all_true());
    }
}

// int_kernel_suite! — mod.rs:117-230
template <typename T, typename Apply>
static void int_kernel_suite(const char* name, Apply apply) {
    std::printf("int_kernel_suite %s\n", name);
    {  // $fn_dense
        Vec64<T> lhs{1, 4, 9, 16}, rhs{1, 2, 3, 4};
        assert_int<T>(apply(lhs, rhs, Op::Add, nullptr), {2, 6, 12, 20}, nullptr);
        assert_int<T>(apply(lhs, rhs, Op::Subtract, nullptr), {0, 2, 6, 12}, nullptr);
        assert_int<T>(apply(lhs, rhs, Op::Multiply, nullptr), {1, 8, 27, 64}, nullptr);
        assert_int<T>(apply(lhs, rhs, Op::Divide, nullptr), {1, 2, 3, 4}, nullptr);
        assert_int<T>(apply(lhs, rhs, Op::Remainder, nullptr), {0, 0, 0, 0}, nullptr);
        std::vector<T> expected;
        for (size_t i = 0; i < 4; ++i) {
            T acc = 1;
            for (T k = 0; k < rhs[i]; ++k) acc = (T)(acc * lhs[i]);  // wrapping_mul
            expected.push_back(acc);
        }
        assert_int<T>(apply(lhs, rhs, Op::Power, nullptr), expected, nullptr);
        std::vector<T> rhs_divzero{0, 0, 0, 0};
        ASSERT(panics([&] { apply(lhs, rhs_divzero, Op::Divide, nullptr); }));     // "must panic"
        ASSERT(panics([&] { apply(lhs, rhs_divzero, Op::Remainder, nullptr); }));
    }
    {  // $fn_masked
        Vec64<T> lhs{10, 20, 30, 40}, rhs{2, 0, 3, 5};
        Bitmask mask = Bitmask::from_bools({true, false, true, false});
        std::vector<bool> expect_mask{true, false, true, false};
        assert_int<T>(apply(lhs, rhs, Op::Divide, &mask), {5, 0, 10, 0}, &expect_mask);
        assert_int<T>(apply(lhs, rhs, Op::Remainder, &mask), {0, 0, 0, 0}, &expect_mask);
        Bitmask mask_divzero = Bitmask::from_bools({true, true, true, true});
        std::vector<T> rhs_divzero{1, 0, 2, 0}, lhs2{100, 100, 100, 100};  // plain slices: pageable memory, staged
        assert_int<T>(apply(lhs2, rhs_divzero, Op::Divide, &mask_divzero), {100, 0, 50, 0}, &expect_mask);
    }
    {  // $fn_empty
        Vec64<T> lhs, rhs;
        ASSERT(apply(lhs, rhs, Op::Add, nullptr).is_empty());
    }
    {  // confirm_equal_len
        Vec64<T> lhs{1, 2, 3}, rhs{1, 2};
        bool mismatch = false;
        try {
            apply(lhs, rhs, Op::Add, nullptr);
        } catch (const KernelError& e) {
            mismatch = e.kind == KernelError::LengthMismatch;
        }
        ASSERT(mismatch);
    }
}

// float_kernel_suite! — mod.rs:293-367
template <typename T, typename Apply>
static void float_kernel_suite(const char* name, Apply apply, T eps) {
    std::printf("float_kernel_suite %s\n", name);
    Vec64<T> lhs{1.0, 4.0, 9.0, 16.0}, rhs{0.5, 2.0, 3.0, 4.0};
    ASSERT((apply(lhs, rhs, Op::Add, nullptr).data == std::vector<T>{1.5, 6.0, 12.0, 20.0}));
    ASSERT((apply(lhs, rhs, Op::Subtract, nullptr).data == std::vector<T>{0.5, 2.0, 6.0, 12.0}));
    ASSERT((apply(lhs, rhs, Op::Multiply, nullptr).data == std::vector<T>{0.5, 8.0, 27.0, 64.0}));
    ASSERT((apply(lhs, rhs, Op::Divide, nullptr).data == std::vector<T>{2.0, 2.0, 3.0, 4.0}));
    auto rem = apply(lhs, rhs, Op::Remainder, nullptr);
    for (size_t i = 0; i < 4; ++i) ASSERT(std::fabs(rem.data[i] - std::fmod(lhs[i], rhs[i])) < eps);
    auto pw = apply(lhs, rhs, Op::Power, nullptr);
    for (size_t i = 0; i < 4; ++i) {
        T expected = std::exp(rhs[i] * std::log(lhs[i]));
        ASSERT(std::fabs(pw.data[i] - expected) <= eps * std::fmax((T)1, expected) * 64);
    }
    std::vector<T> rhs_divzero{0.0, 0.0, 0.0, 0.0};
    auto dz = apply(lhs, rhs_divzero, Op::Divide, nullptr);
    for (size_t i = 0; i < 4; ++i) ASSERT(std::isinf(dz.data[i]));  // "Float division by zero should yield Inf"
    auto rz = apply(lhs, rhs_divzero, Op::Remainder, nullptr);
    for (size_t i = 0; i < 4; ++i) ASSERT(std::isnan(rz.data[i]));
    Bitmask mask = Bitmask::from_bools({true, false, true, false});
    auto m = apply(lhs, rhs, Op::Multiply, &mask);
    ASSERT((m.data == std::vector<T>{0.5, 0.0, 27.0, 0.0}));
    ASSERT(m.null_mask && m.null_mask->len == 4);
    Vec64<T> e1, e2;
    ASSERT(apply(e1, e2, Op::Add, nullptr).is_empty());
}

template <typename T, typename Fma>
static void fma_suite(const char* name, Fma fma) {  // fma_f32 / fma_f64 — mod.rs:372-399
    std::printf("fma %s\n", name);
    Vec64<T> lhs{1.0, 2.0, 3.0}, rhs{4.0, 5.0, 6.0}, acc{0.5, 0.5, 0.5};
    auto out = fma(lhs, rhs, acc, nullptr);
    ASSERT((out.data == std::vector<T>{4.5, 10.5, 18.5}));
    ASSERT(!out.null_mask);
    Bitmask mask = Bitmask::from_bools({true, false, true});
    out = fma(lhs, rhs, acc, &mask);
    ASSERT((out.data == std::vector<T>{4.5, 0.0, 18.5}));
    ASSERT(out.null_mask && out.null_mask->get(0) && !out.null_mask->get(1) && out.null_mask->get(2));
    Vec64<T> e;
    ASSERT(fma(e, e, e, nullptr).is_empty());
}

static void merge_masks_correctness() {  // mod.rs:401-409
    std::printf("merge_masks_correctness\n");
    Bitmask a = Bitmask::from_bools({true, false, true, true}), b = Bitmask::from_bools({true, true, false, true});
    auto merged = merge_bitmasks_to_new(&a, &b, 4);
    ASSERT(merged.has_value());
    std::vector<bool> expected{true, false, false, true};
    for (size_t i = 0; i < 4; ++i) ASSERT(merged->get(i) == expected[i]);
    ASSERT(!merge_bitmasks_to_new(nullptr, nullptr, 4).has_value());
}

static void int_power_short_vs_long() {  // mod.rs:507-537
    std::printf("test_int_dense_power_short_vs_long_input\n");
    for (size_t n : {(size_t)16, (size_t)128}) {
        Vec64<uint32_t> lhs(n, 2u), rhs(n, 10u);
        auto out = apply_int_u32(lhs, rhs, Op::Power);
        for (size_t i = 0; i < n; ++i) ASSERT(out.data[i] == 1024u);
    }
}

static void simd_bitmask_suite(size_t lanes) {  // bitmask/simd.rs:797-955
    std::printf("simd_bitmask_suite LANES=%zu\n", lanes);
    Bitmask a = Bitmask::from_bools({true, false, true, false, true, true, false, false});
    Bitmask b = Bitmask::from_bools({true, true, false, false, true, false, true, false});
    Bitmask c = and_masks(window(a), window(b)), o = or_masks(window(a), window(b)), x = xor_masks(window(a), window(b));
    for (size_t i = 0; i < a.len; ++i) {
        ASSERT(c.get(i) == (a.get(i) & b.get(i)));
        ASSERT(o.get(i) == (a.get(i) | b.get(i)));
        ASSERT(x.get(i) == (a.get(i) ^ b.get(i)));
    }
    Bitmask n4 = Bitmask::from_bools({true, false, true, false});
    Bitmask nn = not_mask(window(n4));
    for (size_t i = 0; i < 4; ++i) ASSERT(nn.get(i) == !n4.get(i));
    {  // test_in_mask_simd_variants
        Bitmask lhs = Bitmask::from_bools({true, false, true, false});
        Bitmask rhs_true = Bitmask::from_bools({true, true, true, true}), rhs_false = Bitmask::from_bools({false, false, false, false});
        Bitmask rhs_both = Bitmask::from_bools({true, false, true, false});
        Bitmask out = in_mask(window(lhs), window(rhs_true));
        for (size_t i = 0; i < 4; ++i) ASSERT(out.get(i) == lhs.get(i));
        out = in_mask(window(lhs), window(rhs_false));
        for (size_t i = 0; i < 4; ++i) ASSERT(out.get(i) == !lhs.get(i));
        out = in_mask(window(lhs), window(rhs_both));
        for (size_t i = 0; i < 4; ++i) ASSERT(out.get(i));
        Bitmask in = in_mask(window(lhs), window(rhs_both)), not_in = not_in_mask(window(lhs), window(rhs_both));
        for (size_t i = 0; i < 4; ++i) ASSERT(not_in.get(i) == !in.get(i));
    }
    {  // eq / ne
        Bitmask p = Bitmask::from_bools({true, false, true, false}), q = Bitmask::from_bools({true, false, false, true});
        Bitmask eq = eq_mask(window(p), window(q)), ne = ne_mask(window(p), window(q));
        for (size_t i = 0; i < 4; ++i) {
            ASSERT(eq.get(i) == (p.get(i) == q.get(i)));
            ASSERT(ne.get(i) == (p.get(i) != q.get(i)));
        }
    }
    {  // all_eq / all_ne
        Bitmask b2 = a;
        ASSERT(all_eq(window(a), window(b2)));
        b2.set(0, false);
        ASSERT(!all_eq(window(a), window(b2)));
        Bitmask p = Bitmask::from_bools({true, false, true}), q = Bitmask::from_bools({false, true, false});
        ASSERT(all_ne(window(p), window(q)));
        ASSERT(!all_ne(window(p), window(p)));
    }
    ASSERT(popcount_mask(window(Bitmask::from_bools({true, false, true, false, true, false, false, true}))) == 4);
    {  // all_true / all_false on 64 * LANES bits
        Bitmask all_true = Bitmask::new_set_all(64 * lanes, true);
        ASSERT(all_true_mask(all_true));
        ASSERT(!all_false_mask(all_true));
        Bitmask not_true = all_true;
        not_true.set(3, false);
        ASSERT(!all_true_mask(not_true));
        ASSERT(all_false_mask(Bitmask::new_set_all(64 * lanes, false)));
    }
}

static void bench_sums() {  // benches/hotloop_benchmark_std.rs:45-57, N = 1000; and N = 1_000_000 (config 1)
    std::printf("bench sums\n");
    for (size_t n : {(size_t)1000, (size_t)1000000}) {
        Vec64<int64_t> v = Vec64<int64_t>::with_capacity(n);
        Vec64<double> f = Vec64<double>::with_capacity(n);
        for (size_t i = 0; i < n; ++i) {
            v.push((int64_t)i);
            f.push((double)i);
        }
        ASSERT(sum_i64(v) == (int64_t)(n * (n - 1) / 2));
        ASSERT(sum_f64(f) == (double)(n * (n - 1) / 2));
        ASSERT(mean_f64(f) == (double)(n * (n - 1) / 2) / (double)n);
    }
}

// Routing / broadcast layer — the reference's enum-dispatch API (include/minarrow_hip_routing.hpp):
//   src/kernels/broadcast/array.rs:485-556 (array (op) array, scalar expansion), :560-626 (array to table), :685-700
//   src/kernels/broadcast/table.rs:431-566 (table (op) table / array / scalar, shape errors)
//   src/kernels/routing/binary_map.rs:76-152 (f64 pairs, Int32 promotion), routing/arithmetic.rs:403-405
static NumericArray i32s(std::initializer_list<int32_t> v) {
    IntegerArray<int32_t> a;
    a.data = Vec64<int32_t>(v);
    return NumericArray::from_int32(std::move(a));
}
static NumericArray f64s(std::initializer_list<double> v) {
    FloatArray<double> a;
    a.data = Vec64<double>(v);
    return NumericArray::from_float64(std::move(a));
}
static bool is_i32(const NumericArray& a, const std::vector<int32_t>& want) {
    return a.try_i32_ref() != nullptr && a.try_i32_ref()->data == want;
}
static bool is_f64(const NumericArray& a, const std::vector<double>& want) {
    return a.try_f64_ref() != nullptr && a.try_f64_ref()->data == want;
}
template <typename F>
static bool kernel_error(KernelError::Kind kind, const char* needle, F f) {
    try {
        f();
    } catch (const KernelError& e) {
        return e.kind == kind && std::string(e.what()).find(needle) != std::string::npos;
    }
    return false;
}
static Table test_table(const char* name, std::initializer_list<int32_t> c1, std::initializer_list<int32_t> c2) {
    Table t;  // create_test_table — table.rs:405-428
    t.name = name;
    t.cols.push_back({"col1", i32s(c1)});
    t.cols.push_back({"col2", i32s(c2)});
    return t;
}

static void routing_suite() {
    std::printf("routing suite\n");
    // test_broadcast_array_add / _scalar_expansion / _sub / _mul / _div — array.rs:485-556
    ASSERT(is_i32(i32s({1, 2, 3}) + i32s({4, 5, 6}), {5, 7, 9}));
    ASSERT(is_i32(i32s({1, 2, 3}) + i32s({10}), {11, 12, 13}));
    ASSERT(is_i32(i32s({10}) + i32s({1, 2, 3}), {11, 12, 13}));
    ASSERT(is_i32(i32s({10, 20, 30}) - i32s({1, 2, 3}), {9, 18, 27}));
    ASSERT(is_i32(i32s({2, 3, 4}) * i32s({5, 6, 7}), {10, 18, 28}));
    ASSERT(is_i32(i32s({100, 200, 300}) / i32s({10, 20, 30}), {10, 10, 10}));
    // Value-level operators — src/kernels/arithmetic/types.rs:921-1054 (test_value_addition, test_all_arithmetic_operators,
    // test_scalar_array_addition, test_reference_operations, test_broadcasting)
    ASSERT(is_i32(i32s({10, 20, 30}) + i32s({2, 4, 6}), {12, 24, 36}));
    ASSERT(is_i32(i32s({10, 20, 30}) - i32s({2, 4, 6}), {8, 16, 24}));
    ASSERT(is_i32(i32s({10, 20, 30}) * i32s({2, 4, 6}), {20, 80, 180}));
    ASSERT(is_i32(i32s({10, 20, 30}) / i32s({2, 4, 6}), {5, 5, 5}));
    ASSERT(is_i32(i32s({10, 20, 30}) % i32s({2, 4, 6}), {0, 0, 0}));
    ASSERT(is_i32(Scalar(int32_t(5)) + i32s({1, 2, 3}), {6, 7, 8}));
    ASSERT(is_i32(i32s({10, 20, 30}) / i32s({2, 4, 5}), {5, 5, 6}));
    ASSERT(is_i32(i32s({10}) * i32s({1, 2, 3, 4, 5}), {10, 20, 30, 40, 50}));
    // broadcast_scalar_to_array — src/kernels/broadcast/scalar.rs:1086-1116
    ASSERT(is_i32(broadcast_scalar_to_array(Op::Add, Scalar(int32_t(5)), i32s({10, 20, 30})), {15, 25, 35}));
    ASSERT(is_i32(broadcast_scalar_to_array(Op::Multiply, Scalar(int32_t(10)), i32s({2, 3, 4})), {20, 30, 40}));
    // test_broadcast_array_to_scalar — array.rs:685-700
    ASSERT(is_i32(broadcast_array_to_scalar(Op::Multiply, i32s({10, 20, 30}), Scalar(int32_t(2))), {20, 40, 60}));
    ASSERT(is_i32(Scalar(int32_t(100)) - i32s({1, 2, 3}), {99, 98, 97}));
    // binary_map.rs:76-152: f64 pairs, length-1 broadcast, Int32 promoted to Float64, array (op) scalar
    ASSERT(is_f64(f64s({1.0, 2.0, 3.0}) + f64s({10.0, 20.0, 30.0}), {11.0, 22.0, 33.0}));
    ASSERT(is_f64(f64s({1.0, 2.0, 3.0}) * f64s({10.0}), {10.0, 20.0, 30.0}));
    ASSERT(is_f64(i32s({1, 2, 3}) + f64s({10.0, 20.0, 30.0}), {11.0, 22.0, 33.0}));
    ASSERT(is_f64(f64s({10.0, 20.0, 30.0}) - i32s({1, 2, 3}), {9.0, 18.0, 27.0}));
    ASSERT(is_f64(f64s({1.0, 2.0, 3.0}) + Scalar(10.0), {11.0, 12.0, 13.0}));
    ASSERT(is_f64(i32s({1, 2, 3}) * Scalar(0.5), {0.5, 1.0, 1.5}));  // Int32 array (op) Float64 scalar promotes
    {   // Int32 with Float32 -> Float32
        FloatArray<float> f;
        f.data = Vec64<float>{0.5f, 0.25f};
        NumericArray r = i32s({1, 2}) + NumericArray::from_float32(std::move(f));
        ASSERT((r.type() == NumericType::Float32 && r.try_f32_ref()->data == std::vector<float>{1.5f, 2.25f}));
    }
    // routing/broadcast.rs:109-111 and routing/arithmetic.rs:403-405
    ASSERT(kernel_error(KernelError::LengthMismatch, "cannot broadcast arrays of length 3 and 2",
                        [] { (void)(i32s({1, 2, 3}) + i32s({1, 2})); }));
    {
        IntegerArray<int64_t> l;
        l.data = Vec64<int64_t>{1, 2, 3};
        NumericArray i64 = NumericArray::from_int64(std::move(l));
        ASSERT(kernel_error(KernelError::UnsupportedType, "Unsupported array type combination",
                            [&] { (void)(i64 + f64s({1.0, 2.0, 3.0})); }));
        ASSERT(kernel_error(KernelError::UnsupportedType, "Unsupported array type combination",
                            [&] { (void)(i64 + Scalar(1.0)); }));  // the scalar keeps its own type (array.rs:146-163)
        ASSERT(kernel_error(KernelError::UnsupportedType, "Unsupported array type combination",
                            [&] { (void)(i64 + i32s({1, 2, 3})); }));
    }
    {   // views are sliced [offset, offset + len) (routing/arithmetic.rs:273-285); the caller's mask gates from bit 0
        NumericArray a = i32s({1, 2, 3, 4, 5, 6}), b = i32s({10, 20, 30, 40, 50, 60});
        NumericArray r = resolve_binary_arithmetic(Op::Add, NumericArrayV(a, 1, 3), NumericArrayV(b, 2, 3));
        ASSERT(is_i32(r, {32, 43, 54}));
        Bitmask m = Bitmask::from_bools({true, false, true});
        r = resolve_binary_arithmetic(Op::Add, NumericArrayV(a, 1, 3), NumericArrayV(b, 2, 3), &m);
        ASSERT(is_i32(r, {32, 0, 54}));
        ASSERT(r.null_mask().has_value() && r.null_mask()->get(0) && !r.null_mask()->get(1) && r.null_mask()->get(2));
        // a length-1 VIEW broadcasts data[0], not data[offset] (broadcast_length_1_array: `a.data[0]`)
        r = resolve_binary_arithmetic(Op::Add, NumericArrayV(a, 0, 3), NumericArrayV(b, 4, 1));
        ASSERT(is_i32(r, {11, 12, 13}));
        // aggregates over a view use the array's own validity at the view offset (guarantee_f64 hand-off)
        IntegerArray<int64_t> wm;
        wm.data = Vec64<int64_t>{1, 2, 3, 4, 5, 6, 7, 8};
        wm.null_mask = Bitmask::from_bools({true, true, false, true, false, true, true, true});
        NumericArray w = NumericArray::from_int64(std::move(wm));
        Aggregate g = sum(NumericArrayV(w, 1, 5));  // rows 2,3,4,5,6 -> valid 2,4,6
        ASSERT(g.sum == 12.0 && g.valid_count == 3 && g.mean() == 4.0);
        g = sum(NumericArrayV(f64s({0.5, 1.5, 2.0})));
        ASSERT(g.sum == 4.0 && g.valid_count == 3);
    }
    // test_table_plus_table / multiply / to_array / to_scalar / shape errors — table.rs:431-566
    {
        Table r = broadcast_table_add(test_table("table1", {1, 2, 3}, {10, 20, 30}), test_table("table2", {4, 5, 6}, {40, 50, 60}));
        ASSERT(r.n_cols() == 2 && r.n_rows() == 3 && r.name == "table1");
        ASSERT(is_i32(r.cols[0].array, {5, 7, 9}) && is_i32(r.cols[1].array, {50, 70, 90}));
        ASSERT(r.cols[0].name == "col1" && r.cols[1].name == "col2");
        r = broadcast_table_with_operator(Op::Multiply, test_table("table1", {2, 3, 4}, {5, 6, 7}), test_table("table2", {10, 10, 10}, {2, 2, 2}));
        ASSERT(is_i32(r.cols[0].array, {20, 30, 40}) && is_i32(r.cols[1].array, {10, 12, 14}));
        r = broadcast_table_to_array(Op::Add, test_table("table1", {10, 20, 30}, {100, 200, 300}), i32s({1, 2, 3}));
        ASSERT(is_i32(r.cols[0].array, {11, 22, 33}) && is_i32(r.cols[1].array, {101, 202, 303}));
        r = broadcast_table_to_scalar(Op::Multiply, test_table("table1", {10, 20, 30}, {100, 200, 300}), Scalar(int32_t(5)));
        ASSERT(is_i32(r.cols[0].array, {50, 100, 150}) && is_i32(r.cols[1].array, {500, 1000, 1500}));
        // test_broadcast_array_to_table(_multiply) — array.rs:560-626
        r = broadcast_array_to_table(Op::Add, i32s({1, 2, 3}), test_table("test", {10, 20, 30}, {100, 200, 300}));
        ASSERT(is_i32(r.cols[0].array, {11, 22, 33}) && is_i32(r.cols[1].array, {101, 202, 303}));
        Table one;
        one.name = "table1";
        one.cols.push_back({"col1", i32s({1, 2, 3})});
        ASSERT(kernel_error(KernelError::Broadcasting, "column count mismatch",
                            [&] { (void)broadcast_table_add(one, test_table("table2", {4, 5, 6}, {40, 50, 60})); }));
        ASSERT(kernel_error(KernelError::Broadcasting, "row count mismatch", [&] {
            (void)broadcast_table_add(test_table("table1", {1, 2}, {10, 20}), test_table("table2", {4, 5, 6}, {40, 50, 60}));
        }));
        ASSERT(kernel_error(KernelError::Broadcasting, "Table column count mismatch: 1 vs 2", [&] {
            (void)broadcast_table_with_operator(Op::Add, one, test_table("table2", {4, 5, 6}, {40, 50, 60}));
        }));
    }
}

// Chunked containers: src/kernels/broadcast/super_array.rs:520-560 (test_broadcast_super_array_add),
// src/structs/chunked/super_table.rs:1305-1397, 1629-1655 (consolidate tests).
static NumericArray i32s_masked(std::initializer_list<int32_t> v, std::initializer_list<bool> valid) {
    IntegerArray<int32_t> a;
    a.data = Vec64<int32_t>(v);
    a.null_mask = Bitmask::from_bools(valid);
    return NumericArray::from_int32(std::move(a));
}
static std::shared_ptr<const Table> batch(std::vector<FieldArray> cols) {
    auto t = std::make_shared<Table>();
    t->cols = std::move(cols);
    return t;
}

static void chunked_suite() {
    std::printf("chunked suite\n");
    {   // test_broadcast_super_array_add
        SuperArray a({i32s({1, 2, 3}), i32s({4, 5, 6})}), b({i32s({10, 10, 10}), i32s({20, 20, 20})});
        SuperArray r = route_super_array_broadcast(Op::Add, a, b);
        ASSERT(r.n_chunks() == 2 && is_i32(r.chunks()[0], {11, 12, 13}) && is_i32(r.chunks()[1], {24, 25, 26}));
        ASSERT(!r.chunks()[0].null_mask().has_value());
        // common mask = union (OR) of the chunks' own masks (super_array.rs:215-229)
        SuperArray am({i32s_masked({1, 2, 3}, {true, false, false}), i32s({4, 5, 6})});
        SuperArray bm({i32s_masked({10, 10, 10}, {false, false, true}), i32s_masked({20, 20, 20}, {true, false, true})});
        r = route_super_array_broadcast(Op::Multiply, am, bm);
        ASSERT(is_i32(r.chunks()[0], {10, 0, 30}) && is_i32(r.chunks()[1], {80, 0, 120}));
        const auto& m0 = r.chunks()[0].null_mask();
        ASSERT(m0.has_value() && m0->get(0) && !m0->get(1) && m0->get(2));
        // null_mask_override replaces the common mask of every chunk (:231)
        Bitmask ov = Bitmask::from_bools({false, true, true});
        r = route_super_array_broadcast(Op::Add, a, b, &ov);
        ASSERT(is_i32(r.chunks()[0], {0, 12, 13}) && is_i32(r.chunks()[1], {0, 25, 26}));
        // mixed element types go chunk by chunk through the type matrix (Int32 with Float64 promotes)
        SuperArray f({f64s({0.5, 0.5, 0.5}), f64s({1.5, 1.5, 1.5})});
        r = route_super_array_broadcast(Op::Add, a, f);
        ASSERT(is_f64(r.chunks()[0], {1.5, 2.5, 3.5}) && is_f64(r.chunks()[1], {5.5, 6.5, 7.5}));
        // chunk shapes must agree pairwise (:202-212)
        SuperArray ragged({i32s({1, 2}), i32s({4, 5, 6})});
        ASSERT(kernel_error(KernelError::Broadcasting, "Super Array broadcasting error",
                            [&] { (void)route_super_array_broadcast(Op::Add, ragged, b); }));
        // dense integer division by zero panics in whichever chunk it happens (std.rs:53-77)
        SuperArray z({i32s({1, 1, 1}), i32s({1, 0, 1})});
        ASSERT(panics([&] { (void)route_super_array_broadcast(Op::Divide, a, z); }));
    }
    {   // test_scalar_to_superarray (scalar.rs:1119-1150), test_scalar_to_superarrayview (:1154-1193: the view's slices are
        // the chunks), and the SuperArray (op) Scalar direction (super_array.rs:87-116)
        SuperArray a({i32s({1, 2, 3}), i32s({4, 5, 6})});
        SuperArray r = broadcast_scalar_to_superarray(Op::Add, Scalar{int32_t(10)}, a);
        ASSERT(r.n_chunks() == 2 && is_i32(r.chunks()[0], {11, 12, 13}) && is_i32(r.chunks()[1], {14, 15, 16}));
        SuperArray v({i32s({10, 20, 30}), i32s({40, 50, 60})});
        r = broadcast_scalar_to_superarray(Op::Multiply, Scalar{int32_t(5)}, v);
        ASSERT(is_i32(r.chunks()[0], {50, 100, 150}) && is_i32(r.chunks()[1], {200, 250, 300}));
        r = broadcast_superarray_to_scalar(Op::Subtract, v, Scalar{int32_t(5)});
        ASSERT(is_i32(r.chunks()[0], {5, 15, 25}) && is_i32(r.chunks()[1], {35, 45, 55}));
        r = broadcast_scalar_to_superarray(Op::Subtract, Scalar{int32_t(5)}, v);  // not commutative: the side matters
        ASSERT(is_i32(r.chunks()[0], {-5, -15, -25}) && is_i32(r.chunks()[1], {-35, -45, -55}));
        // the chunks' own validity is not consulted (array.rs:183 passes None): dense result chunks
        SuperArray am({i32s_masked({1, 2, 3}, {true, false, false}), i32s({4, 5, 6})});
        r = broadcast_superarray_to_scalar(Op::Add, am, Scalar{int32_t(1)});
        ASSERT(is_i32(r.chunks()[0], {2, 3, 4}) && !r.chunks()[0].null_mask().has_value());
        // a scalar of another type goes chunk by chunk through the type matrix: Int32 with Float64 promotes
        r = broadcast_superarray_to_scalar(Op::Multiply, a, Scalar{0.5});
        ASSERT(is_f64(r.chunks()[0], {0.5, 1.0, 1.5}) && is_f64(r.chunks()[1], {2.0, 2.5, 3.0}));
        // dense integer division by a zero scalar panics (std.rs:53-77)
        ASSERT(panics([&] { (void)broadcast_superarray_to_scalar(Op::Divide, a, Scalar{int32_t(0)}); }));
    }
    {   // ArrayView (op) SuperArray and back (super_array.rs:255-363): the view's windows follow the chunk boundaries
        SuperArray sa({i32s({1, 2, 3}), i32s({4, 5}), i32s({6})});
        NumericArrayV view(i32s({0, 10, 20, 30, 40, 50, 60, 70}), 1, 6);  // [10 .. 60]
        SuperArray r = broadcast_arrayview_to_superarray(Op::Subtract, view, sa);
        ASSERT(r.n_chunks() == 3 && is_i32(r.chunks()[0], {9, 18, 27}) && is_i32(r.chunks()[1], {36, 45}) && is_i32(r.chunks()[2], {54}));
        r = broadcast_superarray_to_arrayview(Op::Subtract, sa, view);
        ASSERT(is_i32(r.chunks()[0], {-9, -18, -27}) && is_i32(r.chunks()[1], {-36, -45}) && is_i32(r.chunks()[2], {-54}));
        // a Float64 view over Int32 chunks promotes chunk by chunk
        FloatArray<double> f;
        f.data = Vec64<double>{0.5, 0.5, 0.5, 0.5, 0.5, 0.5};
        r = broadcast_superarray_to_arrayview(Op::Multiply, sa, NumericArrayV(NumericArray::from_float64(std::move(f))));
        ASSERT(is_f64(r.chunks()[0], {0.5, 1.0, 1.5}) && is_f64(r.chunks()[2], {3.0}));
        ASSERT(kernel_error(KernelError::Broadcasting, "ArrayView length (5) does not match SuperArray length (6)",
                            [&] { (void)broadcast_arrayview_to_superarray(Op::Add, NumericArrayV(i32s({0, 10, 20, 30, 40, 50}), 1, 5), sa); }));
    }
    {   // test_consolidate_arena_integer_and_float / _three_batches / _preserves_name
        SuperTable st;
        st.name = "my_table";
        FloatArray<double> f1, f2;
        f1.data = Vec64<double>{1.5, 2.5, 3.5};
        f2.data = Vec64<double>{4.5, 5.5};
        st.batches.push_back(batch({{"ints", i32s({1, 2, 3})}, {"floats", NumericArray::from_float64(std::move(f1))}}));
        st.batches.push_back(batch({{"ints", i32s({4, 5})}, {"floats", NumericArray::from_float64(std::move(f2))}}));
        Table t = consolidate(st);
        ASSERT(t.n_rows() == 5 && t.n_cols() == 2 && t.name == "my_table");
        ASSERT(is_i32(t.cols[0].array, {1, 2, 3, 4, 5}) && is_f64(t.cols[1].array, {1.5, 2.5, 3.5, 4.5, 5.5}));
        ASSERT(t.cols[0].name == "ints" && !t.cols[0].array.null_mask().has_value());
        SuperTable three;
        three.batches = {batch({{"x", i32s({1, 2})}}), batch({{"x", i32s({3})}}), batch({{"x", i32s({4, 5, 6})}})};
        ASSERT(is_i32(consolidate(three).cols[0].array, {1, 2, 3, 4, 5, 6}));
        // test_consolidate_arena_nullable_columns
        SuperTable nul;
        nul.batches = {batch({{"x", i32s_masked({10, 0, 30}, {true, false, true})}}), batch({{"x", i32s_masked({0, 50}, {false, true})}})};
        Table tn = consolidate(nul);
        const NumericArray& x = tn.cols[0].array;
        ASSERT(is_i32(x, {10, 0, 30, 0, 50}) && x.null_mask().has_value());
        const bool want_valid[5] = {true, false, true, false, true};
        for (size_t i = 0; i < 5; ++i) ASSERT(x.null_mask()->get(i) == want_valid[i]);
        // a chunk without a mask contributes all-valid rows (consolidate.rs:91-96)
        SuperTable part;
        part.batches = {batch({{"x", i32s({7, 8})}}), batch({{"x", i32s_masked({0, 50}, {false, true})}})};
        const NumericArray px = consolidate(part).cols[0].array;
        ASSERT(px.null_mask().has_value() && px.null_mask()->get(0) && px.null_mask()->get(1) && !px.null_mask()->get(2) &&
               px.null_mask()->get(3));
        // consolidate() on an empty SuperTable panics (super_table.rs:693-696)
        ASSERT(panics([] { (void)consolidate(SuperTable{}); }));
        // the same tests through the arena path (`arena` feature: consolidate() delegates to consolidate_tables_arena,
        // super_table.rs:727-743): identical columns, all of them windows of ONE 64-byte aligned allocation
        {
            Table ta = consolidate_arena(st);
            ASSERT(ta.n_rows() == 5 && ta.n_cols() == 2 && ta.name == "my_table");
            ASSERT(is_i32(ta.cols[0].array, {1, 2, 3, 4, 5}) && is_f64(ta.cols[1].array, {1.5, 2.5, 3.5, 4.5, 5.5}));
            const auto* ints = ta.cols[0].array.try_i32_ref();
            const auto* flts = ta.cols[1].array.try_f64_ref();
            ASSERT(ints->data.is_shared() && flts->data.is_shared() && !ta.cols[0].array.null_mask().has_value());
            // arena.rs:1668-1689: every region starts on a 64-byte boundary; 5 i32 = 20 bytes, so the floats start at +64
            ASSERT(((uintptr_t)ints->data.data() & 63) == 0 && (const char*)flts->data.data() - (const char*)ints->data.data() == 64);
            ASSERT(is_i32(consolidate_arena(three).cols[0].array, {1, 2, 3, 4, 5, 6}));
            Table an = consolidate_arena(nul);
            const NumericArray& ax = an.cols[0].array;
            ASSERT(is_i32(ax, {10, 0, 30, 0, 50}) && ax.null_mask().has_value() && ax.null_mask()->bits.is_shared());
            for (size_t i = 0; i < 5; ++i) ASSERT(ax.null_mask()->get(i) == want_valid[i]);
            const NumericArray apx = consolidate_arena(part).cols[0].array;
            ASSERT(apx.null_mask().has_value() && apx.null_mask()->get(0) && apx.null_mask()->get(1) && !apx.null_mask()->get(2) &&
                   apx.null_mask()->get(3));
            ASSERT(panics([] { (void)consolidate_arena(SuperTable{}); }));
            // clone is cheap, mutation copies out (arena.rs:1813-1842)
            Vec64<int32_t> view = ints->data;
            ASSERT(view.is_shared() && static_cast<const Vec64<int32_t>&>(view).data() == ints->data.data());
            Vec64<int32_t> cow = ints->data;
            cow[0] = 99;
            ASSERT(!cow.is_shared() && cow[0] == 99 && ints->data[0] == 1 && cow.size() == 5 && cow[4] == 5);
            // the arena columns feed the kernels like any other column
            Aggregate s = sum(NumericArrayV(an.cols[0].array));
            ASSERT(s.sum == 90.0 && s.valid_count == 3);
        }
        // per-column reduce of the consolidated table == fold of the per-batch reduces (config 5)
        Aggregate whole = sum(NumericArrayV(tn.cols[0].array));
        ASSERT(whole.sum == 90.0 && whole.valid_count == 3);
    }
}

// Columns that stay resident in HBM across a chain of operations (DeviceScope / Vec64::to_device): every step reads
// and writes device memory, only the final scalar or an explicit to_host() crosses PCIe — and the bits are those of the
// same chain on host-resident (pinned) columns.
static void device_residency_suite() {
    printf("device residency suite\n");
    const size_t n = 100003;
    Vec64<double> a(n), b(n), c(n);
    for (size_t i = 0; i < n; ++i) {
        a[i] = 0.5 * (double)i;
        b[i] = 1.0 / (double)(i + 1);
        c[i] = (double)(i % 7) - 3.0;
    }
    Bitmask m = Bitmask::new_set_all(n, true);
    for (size_t i = 0; i < n; i += 5) m.set(i, false);
    // host chain: (a + b) * c, masked, then the sum of the valid rows
    FloatArray<double> h1 = apply_float_f64(a, b, Op::Add, &m);
    FloatArray<double> h2 = apply_float_f64(h1.data, c, Op::Multiply, &*h1.null_mask);
    uint64_t h_valid = 0;
    const double h_sum = sum_f64(h2.data, &*h2.null_mask, &h_valid);
    // the same chain with every operand and every result in HBM
    const Vec64<double> da = a.to_device(), db = b.to_device(), dc = c.to_device();
    const Bitmask dm = m.to_device();
    ASSERT(da.is_device() && dm.bits.is_device() && da.size() == n);
    FloatArray<double> d2;
    {
        DeviceScope on_device;
        FloatArray<double> d1 = apply_float_f64(da, db, Op::Add, &dm);
        ASSERT(d1.data.is_device() && d1.null_mask->bits.is_device());
        d2 = apply_float_f64(d1.data, dc, Op::Multiply, &*d1.null_mask);
    }
    ASSERT(d2.data.is_device() && d2.data.size() == n);
    uint64_t d_valid = 0;
    const double d_sum = sum_f64(d2.data, &*d2.null_mask, &d_valid);
    ASSERT(d_valid == h_valid && d_sum == h_sum);
    const Vec64<double> back = d2.data.to_host();
    const Bitmask back_mask = d2.null_mask->to_host();
    ASSERT(!back.is_device() && back.size() == n);
    bool same = true;
    for (size_t i = 0; i < n; ++i) same = same && std::memcmp(&back[i], &h2.data[i], 8) == 0 && back_mask.get(i) == h2.null_mask->get(i);
    ASSERT(same);
    // the CPU is kept away from device memory
    ASSERT(kernel_error(KernelError::InvalidArguments, "to_host", [&] { (void)d2.data[0]; }));
    ASSERT(kernel_error(KernelError::InvalidArguments, "to_host", [&] { Vec64<double> w = d2.data; w.push(1.0); }));
    // the enum-dispatch layer on resident columns: NumericArray operators route to the same kernels
    {
        FloatArray<double> fa, fb;
        fa.data = da;
        fb.data = db;
        const NumericArray na = NumericArray::from_float64(std::move(fa)), nb = NumericArray::from_float64(std::move(fb));
        NumericArray nr;
        {
            DeviceScope on_device;
            nr = na + nb;
        }
        const FloatArray<double>* r = nr.try_f64_ref();
        ASSERT(r != nullptr && r->data.is_device() && r->data.size() == n);
        const Vec64<double> rh = r->data.to_host();
        bool ok = true;
        for (size_t i = 0; i < n; ++i) ok = ok && rh[i] == a[i] + b[i];
        ASSERT(ok);
    }
    // what residency buys: the chain (x + y) * z -> sum over 2^25 rows, columns in pinned host memory vs in HBM
    {
        const size_t big = (size_t)1 << 25;
        Vec64<double> x(big, 1.5), y(big, 0.25), z(big, 2.0);
        auto chain = [&](const Vec64<double>& p, const Vec64<double>& q, const Vec64<double>& r) {
            FloatArray<double> s1 = apply_float_f64(p, q, Op::Add);
            FloatArray<double> s2 = apply_float_f64(s1.data, r, Op::Multiply);
            return sum_f64(s2.data);
        };
        auto time_ms = [&](const std::function<double()>& f, double* out) {
            *out = f();  // warm
            double best = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                const auto t0 = std::chrono::steady_clock::now();
                *out = f();
                best = std::min(best, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
            }
            return best;
        };
        double host_sum = 0, dev_sum = 0;
        const double host_ms = time_ms([&] { return chain(x, y, z); }, &host_sum);
        const Vec64<double> dx = x.to_device(), dy = y.to_device(), dz = z.to_device();
        const double dev_ms = time_ms([&] { DeviceScope on_device; return chain(dx, dy, dz); }, &dev_sum);
        ASSERT(host_sum == dev_sum && dev_sum == 3.5 * (double)big);
        std::printf("  chain (x + y) * z -> sum, %zu rows: pinned host columns %.2f ms, HBM-resident columns %.3f ms (%.0fx)\n", big,
                    host_ms, dev_ms, host_ms / dev_ms);
    }
    // results made outside a DeviceScope land in pinned host memory again, whatever the inputs
    FloatArray<double> mixed = apply_float_f64(da, b, Op::Add);
    ASSERT(!mixed.data.is_device() && mixed.data[1] == a[1] + b[1]);
}

// The typed mirror of the Rayon path (include/minarrow_hip_parallel.hpp): rayon_simd_sum_{i64,f64} over a column
// scattered across the node's GPUs (benches/benchmark_parallel_simd.rs:81-98), dense and Bitmask-gated, against the
// closed forms of the bench's own inputs (:103,:115: v[i] = i) and a host loop for the gated case.
static void parallel_mirror_suite() {
    std::printf("parallel mirror (ma::Group, ShardedColumn)\n");
    const int n_dev = std::min<int>(ma_device_count(), 8);
    std::vector<int32_t> devs(n_dev);
    for (int i = 0; i < n_dev; ++i) devs[i] = i;
    const size_t n = 2000003;
    std::vector<int64_t> hi(n);
    std::vector<double> hf(n);
    std::vector<uint8_t> bits((n + 7) / 8 + 8, 0);
    int64_t want_masked = 0;
    uint64_t want_valid = 0;
    for (size_t i = 0; i < n; ++i) {
        hi[i] = (int64_t)i;
        hf[i] = (double)i;
        if ((i * 2654435761u >> 7) % 10 != 0) {  // ~10 % nulls
            bits[i >> 3] |= (uint8_t)(1u << (i & 7));
            want_masked += (int64_t)i;
            ++want_valid;
        }
    }
    for (uint32_t flags : {(uint32_t)(MA_GROUP_EXCHANGE_RCCL | MA_GROUP_EXCHANGE_FALLBACK_HOST), 0u}) {
        ma::Group g(devs, flags);
        ASSERT(g.size() == (size_t)n_dev && g.issue_threads());
        auto chunks = ma::row_chunks(n, g.size());
        ASSERT(chunks.front().first == 0 && chunks.back().second == n);
        for (size_t i = 1; i < chunks.size(); ++i) ASSERT(chunks[i].first == chunks[i - 1].second && chunks[i].first % 64 == 0);
        auto ci = g.scatter(hi.data(), n);
        auto cf = g.scatter(hf.data(), n);
        uint64_t valid = 0;
        const int64_t want = (int64_t)(n * (n - 1) / 2);
        ASSERT(g.rayon_simd_sum_i64(ci, &valid) == want && valid == n);
        ASSERT(g.rayon_simd_sum_f64(cf, &valid) == (double)want && valid == n);  // < 2^53: exact in any order
        auto mi = g.scatter(hi.data(), n, bits.data());
        auto mf = g.scatter(hf.data(), n, bits.data());
        ASSERT(g.rayon_simd_sum_i64(mi, &valid) == want_masked && valid == want_valid);
        ASSERT(g.rayon_simd_sum_f64(mf, &valid) == (double)want_masked && valid == want_valid);
        // the bench's step: both columns, one exchange, enqueue-only, several steps back to back
        for (int step = 0; step < 3; ++step) g.enqueue_sums(1, ci, cf);
        g.wait();
        auto both = g.sums(1);
        ASSERT(both.first == want && both.second == (double)want);
        // first contact, the C++ way: the self-test first, bounded waits, and a stalled exchange that ends in a KernelError
        // (not a blocked host) after which the same group — and the columns scattered over it — go on with a fresh exchange
        const ma_selftest_report rep = g.selftest(20000.0);
        ASSERT(std::string(rep.text).rfind("PASS", 0) == 0 && rep.n_members == n_dev && rep.forms_ok == rep.forms_tried && rep.forms_tried != 0);
        g.enqueue_sums(1, ci, cf);
        g.wait_for(20000.0);
        ASSERT(g.broken() == 0 && g.sums(1).first == want);
        ASSERT(ma_group_test_stall_next_exchange(g.get(), n_dev - 1) == MA_OK);
        g.enqueue_sums(1, ci, cf);
        bool timed_out = false;
        try {
            g.wait_for(250.0);
        } catch (const ma::KernelError& e) {
            timed_out = std::string(e.what()).find("did not finish within 250 ms") != std::string::npos;
        }
        ASSERT(timed_out && g.broken() == 1);
        g.rebuild(0u);  // the host fold: no collective left to wait for
        ASSERT(g.broken() == 0 && !g.rccl());
        ASSERT(g.rayon_simd_sum_i64(ci, &valid) == want && valid == n);
        // a column with the wrong number of chunks is refused on the host
        ma::ShardedColumn<int64_t> bad = ci;
        bad.chunks.push_back(ci.chunks[0]);
        bad.lens.push_back(1);
        bool threw = false;
        try {
            (void)g.rayon_simd_sum_i64(bad);
        } catch (const ma::KernelError&) {
            threw = true;
        }
        ASSERT(threw);
    }
}

// The reference's hot loop of sums (benches/hotloop_benchmark_avg_std.rs:48-62: ITERATIONS passes, an i64 and an f64 sum each; the pass itself: hotloop_benchmark_std.rs:109-127: one pass per call over the same arrays) as a
// pipeline on one GPU (ma::ScanLanes over ma_scan_lanes_*): consecutive fused scans on two streams, each into its own record,
// with a kernel of the host's own on the context in between (ordered in front of the scan that reads what it wrote).
static void scan_lanes_suite() {
    std::printf("scan lanes (ma::ScanLanes)\n");
    ma_ctx* ctx = nullptr;
    ASSERT(ma_ctx_create(0, &ctx) == MA_OK);
    const size_t n = 1500007;
    void *di = nullptr, *df = nullptr, *dout = nullptr, *rec = nullptr;
    ASSERT(ma_dev_alloc(ctx, n * 8 + 64, &di) == MA_OK && ma_dev_alloc(ctx, n * 8 + 64, &df) == MA_OK);
    ASSERT(ma_dev_alloc(ctx, n * 8 + 64, &dout) == MA_OK && ma_dev_alloc(ctx, 64 * 6, &rec) == MA_OK);
    ASSERT(ma_synth_iota_i64(ctx, (int64_t*)di, n, 0) == MA_OK && ma_synth_iota_f64(ctx, (double*)df, n, 0) == MA_OK);
    ASSERT(ma_ctx_set_async(ctx, 1) == MA_OK);
    const int64_t tri = (int64_t)(n * (n - 1) / 2);
    {
        ma::ScanLanes lanes(ctx);
        for (int k = 0; k < 6; ++k) {
            lanes.join();  // the add below overwrites what the scan before it read
            ASSERT(ma_apply_int_i64_scalar_rhs(ctx, (const int64_t*)di, n, (int64_t)k, MA_OP_ADD, nullptr, 0, (int64_t*)dout, nullptr) == MA_OK);
            lanes.enqueue_pair((const int64_t*)dout, (const double*)df, n, (uint64_t*)rec + 8 * k);
        }
        // ... and single columns through the typed form: the f64 column, then the i64 one re-read as 2 n i32 rows
        uint64_t* single = (uint64_t*)rec;  // records 0..5 are downloaded first, below; these two scans go to spare words behind them
        (void)single;
        lanes.synchronize();
        ASSERT(lanes.scans() == 6);
        uint64_t w[48];
        ASSERT(ma_dev_download(ctx, w, rec, sizeof(w)) == MA_OK);
        for (int k = 0; k < 6; ++k) {
            double hi, lo;
            std::memcpy(&hi, &w[8 * k + 2], 8);
            std::memcpy(&lo, &w[8 * k + 3], 8);
            ASSERT((int64_t)w[8 * k] == tri + (int64_t)k * (int64_t)n && w[8 * k + 1] == n && w[8 * k + 4] == n);
            ASSERT(hi + lo == (double)tri);  // < 2^53: exact
        }
        lanes.enqueue_sum('g', df, n, (uint64_t*)rec, (uint64_t*)rec + 2, (double*)rec + 1);
        lanes.enqueue_sum('i', di, 2 * n, (uint64_t*)rec + 8, (uint64_t*)rec + 9);  // iota < 2^31: the high halves are zero
        lanes.synchronize();
        ASSERT(lanes.scans() == 8 && ma_dev_download(ctx, w, rec, 128) == MA_OK);
        double hi, lo;
        std::memcpy(&hi, &w[0], 8);
        std::memcpy(&lo, &w[1], 8);
        ASSERT(hi + lo == (double)tri && w[2] == n && (int64_t)w[8] == tri && w[9] == 2 * n);
    }
    ASSERT(ma_ctx_set_async(ctx, 0) == MA_OK);
    for (void* p : {di, df, dout, rec}) ASSERT(ma_dev_free(ctx, p) == MA_OK);
    ma_ctx_destroy(ctx);
}

// rayon_simd_sum_{i64,f64} over the GPUs of the node, driven from ONE compiled host process straight through the C ABI
// (benches/benchmark_parallel_simd.rs:81-98: `par_chunks(1 << 20).map(simd_sum).sum()` — here one row chunk per device,
// the partials meeting in the library's exchange). Every visible device takes part (one on the test pool: the RCCL
// exchange then runs with one rank); the f64 total must equal the single-device sum bit for bit when there is one
// member and stay within 1 ULP of the exact sum otherwise.
static void multi_gpu_group_suite() {
    std::printf("multi-GPU group (C ABI)\n");
    const int n_dev = std::min<int>(ma_device_count(), 8);
    ASSERT(n_dev >= 1);
    const size_t n = 3000017;
    std::vector<int32_t> devs(n_dev);
    for (int i = 0; i < n_dev; ++i) devs[i] = i;
    for (uint32_t flags : {0u, (uint32_t)(MA_GROUP_EXCHANGE_RCCL | MA_GROUP_EXCHANGE_FALLBACK_HOST),
                           (uint32_t)(MA_GROUP_EXCHANGE_RCCL | MA_GROUP_EXCHANGE_OVERLAP),  // exchange of step k on side streams
                           (uint32_t)MA_GROUP_ISSUE_CALLER}) {                              // round-2 form: the caller issues
        ma_group* g = nullptr;
        ASSERT(ma_group_create_ex(devs.data(), n_dev, flags, &g) == MA_OK);
        if (!g) continue;
        ASSERT(ma_group_size(g) == n_dev);
        if (flags & MA_GROUP_EXCHANGE_RCCL) ASSERT(ma_group_exchange_kind(g) == 1);  // distinct devices + librccl present: no fallback taken
        ASSERT(ma_group_issue_kind(g) == ((flags & MA_GROUP_ISSUE_CALLER) ? 0 : 1));
        for (int a = 0; a < n_dev; ++a)  // probed at creation; a device always reaches itself
            ASSERT(ma_group_peer_access(g, a, a) == 1 && ma_group_peer_access(g, a, (a + 1) % n_dev) >= 0);
        ASSERT(std::string(ma_group_exchange_note(g)).find("peer access") != std::string::npos);
        // 64-row-aligned row chunks; chunk i lives on device i
        std::vector<void*> di(n_dev), df(n_dev);
        std::vector<const int64_t*> pi(n_dev);
        std::vector<const double*> pf(n_dev);
        std::vector<size_t> lens(n_dev);
        const size_t units = (n + 63) / 64;
        for (int r = 0; r < n_dev; ++r) {
            const size_t lo = std::min(n, (units * r / n_dev) * 64), hi = r + 1 == n_dev ? n : std::min(n, (units * (r + 1) / n_dev) * 64);
            lens[r] = hi - lo;
            ma_ctx* c = ma_group_ctx(g, r);
            ASSERT(ma_dev_alloc(c, lens[r] * 8 + 64, &di[r]) == MA_OK && ma_dev_alloc(c, lens[r] * 8 + 64, &df[r]) == MA_OK);
            ASSERT(ma_synth_iota_i64(c, (int64_t*)di[r], lens[r], (int64_t)lo) == MA_OK);
            ASSERT(ma_synth_iota_f64(c, (double*)df[r], lens[r], (int64_t)lo) == MA_OK);
            pi[r] = (const int64_t*)di[r];
            pf[r] = (const double*)df[r];
            // residency: the chunk lives on ITS member's device (a pointer into another GPU's HBM is refused, not faulted on)
            ASSERT(ma_pointer_device(di[r]) == ma_ctx_hip_device(c) && ma_ctx_device(c) == devs[r]);
        }
        if (n_dev > 1) {  // chunk 1 handed to member 1 but resident on member 0's device
            std::vector<const int64_t*> wrong(pi);
            wrong[1] = pi[0];
            ASSERT(ma_group_enqueue_sum_i64(g, 0, wrong.data(), lens.data(), nullptr, nullptr) == MA_ERR_INVALID_ARGUMENT);
        }
        // asynchronous form: two steps back to back, both reductions share one exchange, one synchronize
        for (int step = 0; step < 2; ++step) {
            ASSERT(ma_group_enqueue_sum_i64(g, 0, pi.data(), lens.data(), nullptr, nullptr) == MA_OK);
            ASSERT(ma_group_enqueue_sum_f64(g, 0, pf.data(), lens.data(), nullptr, nullptr) == MA_OK);
            ASSERT(ma_group_exchange(g) == MA_OK);
        }
        ASSERT(ma_group_synchronize(g) == MA_OK);
        const int64_t want = (int64_t)(n * (n - 1) / 2);
        for (int m = 0; m < n_dev; ++m) {  // every GPU holds the job's finals
            int64_t isum = 0;
            uint64_t icnt = 0, fcnt = 0;
            double fsum = 0;
            ASSERT(ma_group_member_result(g, m, 0, &isum, &icnt, &fsum, &fcnt) == MA_OK);
            ASSERT(isum == want && icnt == n && fcnt == n && fsum == (double)want);  // sum(0..n) < 2^53: exact in any order
        }
        // synchronous one-call form
        int64_t s = 0;
        uint64_t c = 0;
        ASSERT(ma_group_sum_i64(g, pi.data(), lens.data(), nullptr, nullptr, &s, &c) == MA_OK && s == want && c == n);
        // SuperArray (+) SuperArray over the group (route_super_array_broadcast, super_array.rs:180-251): chunk i on
        // device i, no exchange; checked through the linearity of the wrapping sum: sum(a + a) == 2 sum(a)
        {
            std::vector<void*> dout(n_dev);
            std::vector<const void*> lhs(n_dev);
            std::vector<const int64_t*> po(n_dev);
            std::vector<int32_t> has(n_dev, 7);
            for (int r = 0; r < n_dev; ++r) {
                ASSERT(ma_dev_alloc(ma_group_ctx(g, r), lens[r] * 8 + 64, &dout[r]) == MA_OK);
                lhs[r] = di[r];
                po[r] = (const int64_t*)dout[r];
            }
            ASSERT(ma_group_route_super_array_broadcast(g, 'l', MA_OP_ADD, (size_t)n_dev, lhs.data(), lens.data(), nullptr, lhs.data(),
                                                        lens.data(), nullptr, nullptr, dout.data(), nullptr, has.data()) == MA_OK);
            ASSERT(ma_group_synchronize(g) == MA_OK);
            for (int r = 0; r < n_dev; ++r) ASSERT(has[r] == 0);
            ASSERT(ma_group_sum_i64(g, po.data(), lens.data(), nullptr, nullptr, &s, &c) == MA_OK && s == 2 * want && c == n);
            std::vector<size_t> bad(lens);
            bad[n_dev - 1] += 1;  // "Super Array broadcasting error": reported before any member starts
            ASSERT(ma_group_route_super_array_broadcast(g, 'l', MA_OP_ADD, (size_t)n_dev, lhs.data(), lens.data(), nullptr, lhs.data(),
                                                        bad.data(), nullptr, nullptr, dout.data(), nullptr, nullptr) == MA_ERR_LENGTH_MISMATCH);
            for (int r = 0; r < n_dev; ++r) ASSERT(ma_dev_free(ma_group_ctx(g, r), dout[r]) == MA_OK);
        }
        // SuperTable::consolidate of the sharded column onto member 0 (peer copies; super_table.rs:657-743): the chunks are
        // slices of one iota column, so the consolidated column is 0, 1, 2, ... again
        {
            ma_ctx* c0 = ma_group_ctx(g, 0);
            void* whole = nullptr;
            ASSERT(ma_dev_alloc(c0, n * 8 + 64, &whole) == MA_OK);
            std::vector<const void*> chunks(n_dev);
            for (int r = 0; r < n_dev; ++r) chunks[r] = di[r];
            int32_t has = 7;
            ASSERT(ma_group_consolidate_column(g, 0, 8, (size_t)n_dev, chunks.data(), lens.data(), nullptr, nullptr, whole, nullptr,
                                               &has) == MA_OK);
            ASSERT(ma_group_synchronize(g) == MA_OK && has == 0);
            std::vector<int64_t> host(n);
            ASSERT(ma_dev_download(c0, host.data(), whole, n * 8) == MA_OK);
            bool iota = true;
            for (size_t j = 0; j < n; ++j) iota = iota && host[j] == (int64_t)j;
            ASSERT(iota);
            ASSERT(ma_dev_free(c0, whole) == MA_OK);
        }
        for (int r = 0; r < n_dev; ++r) {
            ASSERT(ma_dev_free(ma_group_ctx(g, r), di[r]) == MA_OK && ma_dev_free(ma_group_ctx(g, r), df[r]) == MA_OK);
        }
        ma_group_destroy(g);
    }
    // one process per GPU would use ma_comm_*: here a one-rank communicator on device 0
    ma_ctx* ctx = nullptr;
    ASSERT(ma_ctx_create(0, &ctx) == MA_OK);
    uint8_t id[MA_COMM_ID_BYTES];
    ma_comm* comm = nullptr;
    ASSERT(ma_comm_unique_id(id) == MA_OK && ma_comm_create(ctx, id, 0, 1, &comm) == MA_OK);
    if (comm) {
        void *col = nullptr, *rec = nullptr, *gathered = nullptr, *fin = nullptr;
        ASSERT(ma_dev_alloc(ctx, n * 8, &col) == MA_OK && ma_dev_alloc(ctx, 64, &rec) == MA_OK &&
               ma_dev_alloc(ctx, 64, &gathered) == MA_OK && ma_dev_alloc(ctx, 32, &fin) == MA_OK);
        ASSERT(ma_dev_memset(ctx, rec, 0, 64) == MA_OK && ma_synth_iota_i64(ctx, (int64_t*)col, n, 0) == MA_OK);
        uint64_t* r = (uint64_t*)rec;
        ASSERT(ma_i64_sum(ctx, (const int64_t*)col, n, nullptr, 0, 0, (int64_t*)&r[0], &r[1]) == MA_OK);
        ASSERT(ma_comm_sum_exchange(comm, r, 1, 1, (uint64_t*)gathered, (uint64_t*)fin) == MA_OK);
        uint64_t out[4] = {0, 0, 0, 0};
        ASSERT(ma_dev_download(ctx, out, fin, 32) == MA_OK);
        ASSERT((int64_t)out[0] == (int64_t)(n * (n - 1) / 2) && out[1] == n);
        // the overlapped form: the exchange runs on the communicator's own stream, slot_wait orders the next use of the set
        ASSERT(ma_ctx_set_async(ctx, 1) == MA_OK);
        for (int step = 0; step < 4; ++step) {
            ASSERT(ma_comm_slot_wait(comm, 0) == MA_OK);
            ASSERT(ma_i64_sum(ctx, (const int64_t*)col, n - (size_t)step, nullptr, 0, 0, (int64_t*)&r[0], &r[1]) == MA_OK);
            ASSERT(ma_comm_sum_exchange_overlapped(comm, 0, r, 1, 1, (uint64_t*)gathered, (uint64_t*)fin) == MA_OK);
        }
        ASSERT(ma_comm_synchronize(comm) == MA_OK && ma_ctx_set_async(ctx, 0) == MA_OK);
        ASSERT(ma_dev_download(ctx, out, fin, 32) == MA_OK);
        ASSERT((int64_t)out[0] == (int64_t)((n - 3) * (n - 4) / 2) && out[1] == n - 3);
        ma_comm_destroy(comm);
        for (void* p : {col, rec, gathered, fin}) ASSERT(ma_dev_free(ctx, p) == MA_OK);
    }
    ma_ctx_destroy(ctx);
}

int main() {
    try {
        int_kernel_suite<int32_t>("i32", [](Slice<int32_t> l, Slice<int32_t> r, Op op, const Bitmask* m) { return apply_int_i32(l, r, op, m); });
        int_kernel_suite<uint32_t>("u32", [](Slice<uint32_t> l, Slice<uint32_t> r, Op op, const Bitmask* m) { return apply_int_u32(l, r, op, m); });
        int_kernel_suite<int64_t>("i64", [](Slice<int64_t> l, Slice<int64_t> r, Op op, const Bitmask* m) { return apply_int_i64(l, r, op, m); });
        int_kernel_suite<uint64_t>("u64", [](Slice<uint64_t> l, Slice<uint64_t> r, Op op, const Bitmask* m) { return apply_int_u64(l, r, op, m); });
        float_kernel_suite<float>("f32", [](Slice<float> l, Slice<float> r, Op op, const Bitmask* m) { return apply_float_f32(l, r, op, m); }, 1e-6f);
        float_kernel_suite<double>("f64", [](Slice<double> l, Slice<double> r, Op op, const Bitmask* m) { return apply_float_f64(l, r, op, m); }, 1e-12);
        fma_suite<float>("f32", [](Slice<float> a, Slice<float> b, Slice<float> c, const Bitmask* m) { return apply_fma_f32(a, b, c, m); });
        fma_suite<double>("f64", [](Slice<double> a, Slice<double> b, Slice<double> c, const Bitmask* m) { return apply_fma_f64(a, b, c, m); });
        merge_masks_correctness();
        int_power_short_vs_long();
        for (size_t lanes : {(size_t)8, (size_t)16, (size_t)32, (size_t)64}) simd_bitmask_suite(lanes);
        bench_sums();
        routing_suite();
        chunked_suite();
        device_residency_suite();
        multi_gpu_group_suite();
        parallel_mirror_suite();
        scan_lanes_suite();
        // fused scalar broadcast: [10,20,30] * 2 = [20,40,60] (src/kernels/broadcast/array.rs:685-700)
        Vec64<int32_t> arr{10, 20, 30};
        ASSERT((apply_int_i32_scalar_rhs(arr, 2, Op::Multiply).data == std::vector<int32_t>{20, 40, 60}));
    } catch (const std::exception& e) {
        std::printf("UNEXPECTED EXCEPTION: %s\n", e.what());
        return 2;
    }
    std::printf("%d assertions, %d failed\n", g_checked, g_failed);
    return g_failed ? 1 : 0;
}
