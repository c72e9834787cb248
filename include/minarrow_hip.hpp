// minarrow_hip.hpp — typed C++17 host mirror of the reference's kernel-layer API over the C ABI
// (include/minarrow_hip.h). Header-only; link with -lminarrow_hip.
//
// The reference's host language is Rust, which this environment cannot compile; this is the host side "above
// the C ABI" in the other compiled language at hand. Names, argument order and error behaviour follow the
// reference so that tests read like its own (minarrow_amd/cpp/ref_suite.cpp replays
// src/kernels/arithmetic/mod.rs:117-537 and src/kernels/bitmask/simd.rs:797-955):
//
//   Vec64<T>            src/lib.rs:99 (vec64 crate): 64-byte aligned vector — here pinned, device-mapped memory
//                       from ma_alloc64_pinned (hipHostMalloc), so kernels use it in place
//   Bitmask             src/structs/bitmask.rs:66-71
//   IntegerArray<T>     src/structs/variants/integer.rs:106-111   { data, null_mask }
//   FloatArray<T>       src/structs/variants/float.rs:111
//   ArithmeticOperator  src/enums/operators.rs:18-48
//   KernelError         src/enums/error.rs:157-187  (thrown; Rust returns Result<_, KernelError>)
//   Panic               what the reference panics on: dense integer division by zero (std.rs:53-77)
//   apply_int_* / apply_float_* / apply_fma_*   src/kernels/arithmetic/dispatch.rs:376-418
//   and_masks ... all_false_mask                src/kernels/bitmask/dispatch.rs:47-295
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <initializer_list>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <memory>
#include <vector>

#include "minarrow_hip.h"

namespace ma {

enum class ArithmeticOperator : int32_t { Add = 0, Subtract, Multiply, Divide, Remainder, Power, FloorDiv };

struct KernelError : std::runtime_error {
    enum Kind { LengthMismatch, UnsupportedType, InvalidArguments, Device, NoDevice, Broadcasting } kind;
    KernelError(Kind k, const std::string& what) : std::runtime_error(what), kind(k) {}
};
// The reference panics (it does not return Err) on dense integer ÷0; catch_unwind in its tests == catch here.
struct Panic : std::runtime_error {
    explicit Panic(const std::string& what) : std::runtime_error(what) {}
};

inline void check(ma_status st) {
    if (st == MA_OK) return;
    std::string msg = ma_last_error_string();
    switch (st) {
        case MA_ERR_LENGTH_MISMATCH: throw KernelError(KernelError::LengthMismatch, msg);
        case MA_ERR_DIVIDE_BY_ZERO: throw Panic(msg);
        case MA_ERR_UNSUPPORTED: throw KernelError(KernelError::UnsupportedType, msg);
        case MA_ERR_INVALID_ARGUMENT: throw KernelError(KernelError::InvalidArguments, msg);
        case MA_ERR_NO_DEVICE: throw KernelError(KernelError::NoDevice, msg);
        default: throw KernelError(KernelError::Device, msg);
    }
}

// One device + one stream. `Context::global()` is what the free functions below use (device 0).
class Context {
  public:
    explicit Context(int device = 0) { check(ma_ctx_create(device, &ctx_)); }
    ~Context() { ma_ctx_destroy(ctx_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    ma_ctx* get() const { return ctx_; }
    static Context& global() {
        static Context c(0);
        return c;
    }

  private:
    ma_ctx* ctx_ = nullptr;
};

// Where the typed API allocates its RESULTS. Host (default): pinned host memory, the Vec64 stand-in, readable and
// writable by the CPU. Device: HBM (ma_dev_alloc) — a column that stays resident across several operations runs every
// one of them at HBM rate instead of crossing PCIe twice per call; the CPU must not dereference such a
// vector: download it with to_host(). Inputs may live anywhere regardless (the library classifies every pointer).
enum class Placement { Host, Device };
namespace detail {
inline Placement& result_placement() {
    static thread_local Placement p = Placement::Host;
    return p;
}
}  // namespace detail
// RAII: results of apply_* / consolidate / ... made on this thread while the scope lives are allocated in HBM.
struct DeviceScope {
    Placement saved;
    DeviceScope() : saved(detail::result_placement()) { detail::result_placement() = Placement::Device; }
    ~DeviceScope() { detail::result_placement() = saved; }
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
};

// 64-byte aligned, pinned, device-mapped vector: the Vec64 stand-in.
template <typename T>
class Vec64 {
  public:
    Vec64() = default;
    explicit Vec64(size_t n, T fill = T()) { resize(n, fill); }
    Vec64(std::initializer_list<T> init) {
        reserve(init.size());
        for (const T& v : init) data_[len_++] = v;
    }
    // Cloning a view of shared memory shares it (cheap, src/structs/arena.rs:1813-1826); cloning owned memory copies.
    Vec64(const Vec64& o) {
        if (o.owner_) {
            owner_ = o.owner_;
            data_ = o.data_;
            len_ = cap_ = o.len_;
            device_ = o.device_;
            return;
        }
        reserve(o.len_);
        if (o.len_) std::memcpy(data_, o.data_, o.len_ * sizeof(T));
        len_ = o.len_;
    }
    Vec64(Vec64&& o) noexcept : data_(o.data_), len_(o.len_), cap_(o.cap_), owner_(std::move(o.owner_)), device_(o.device_) {
        o.data_ = nullptr;
        o.len_ = o.cap_ = 0;
        o.device_ = false;
    }
    Vec64& operator=(Vec64 o) noexcept {
        std::swap(data_, o.data_);
        std::swap(len_, o.len_);
        std::swap(cap_, o.cap_);
        std::swap(owner_, o.owner_);
        std::swap(device_, o.device_);
        return *this;
    }
    ~Vec64() {
        if (data_ && !owner_) ma_free_pinned(data_);
    }
    static Vec64 with_capacity(size_t n) {
        if (detail::result_placement() == Placement::Device) return on_device(n, 0, /*written_by_kernels=*/true);
        Vec64 v;
        v.reserve(n);
        return v;
    }
    // `cap` elements of HBM owned by the returned vector (a window whose owner frees the allocation), length `len`.
    // Padded like the host form so that whole-u64-word bitmap writes stay inside it.
    // written_by_kernels: a result column (the `out` the reference allocates with Vec64::with_capacity,
    // src/kernels/arithmetic/dispatch.rs:88-89) — large ones are picked for their write rate (ma_dev_alloc_output).
    static Vec64 on_device(size_t cap, size_t len, bool written_by_kernels = false) {
        void* p = nullptr;
        ma_ctx* ctx = Context::global().get();
        const size_t bytes = ((cap * sizeof(T) + 63) / 64) * 64 + 64;
        if (written_by_kernels) check(ma_dev_alloc_output(ctx, bytes, &p, nullptr));
        else check(ma_dev_alloc(ctx, bytes, &p));
        Vec64 v = from_shared(std::shared_ptr<void>(p, [ctx](void* q) { (void)ma_dev_free(ctx, q); }), static_cast<T*>(p), len);
        v.cap_ = cap;
        v.device_ = true;
        return v;
    }
    bool is_device() const { return device_; }
    // Upload / download (ma_dev_upload / ma_dev_download); a vector already on the requested side is shared, not copied.
    Vec64 to_device() const {
        if (device_) return *this;
        Vec64 d = on_device(len_, len_);
        if (len_) check(ma_dev_upload(Context::global().get(), d.data_, data_, len_ * sizeof(T)));
        return d;
    }
    Vec64 to_host() const {
        if (!device_) return *this;
        Vec64 h;
        h.reserve(len_);
        if (len_) check(ma_dev_download(Context::global().get(), h.data_, data_, len_ * sizeof(T)));
        h.len_ = len_;
        return h;
    }
    // A window of memory kept alive by `owner` — Buffer::from_shared over a SharedBuffer (src/structs/buffer.rs:168-217,
    // src/structs/shared_buffer/mod.rs:82-87); what ArenaRegion::to_buffer hands out (src/structs/arena.rs:502-517).
    // Read access uses it in place; the first mutation copies it out (make_owned_mut, buffer.rs:474-507).
    static Vec64 from_shared(std::shared_ptr<void> owner, T* ptr, size_t n) {
        Vec64 v;
        v.owner_ = std::move(owner);
        v.data_ = ptr;
        v.len_ = v.cap_ = n;
        return v;
    }
    bool is_shared() const { return (bool)owner_; }
    void reserve(size_t n) {
        if (owner_) make_owned(n);
        if (n <= cap_) return;
        void* p = nullptr;
        // round up so that whole-u64-word accesses of a bitmap stay inside the allocation
        size_t bytes = ((n * sizeof(T) + 63) / 64) * 64 + 64;
        check(ma_alloc64_pinned(bytes, &p));
        if (len_) std::memcpy(p, data_, len_ * sizeof(T));
        if (data_) ma_free_pinned(data_);
        data_ = static_cast<T*>(p);
        cap_ = bytes / sizeof(T);
    }
    void resize(size_t n, T fill = T()) {
        reserve(n);
        for (size_t i = len_; i < n; ++i) data_[i] = fill;
        len_ = n;
    }
    void set_len(size_t n) {  // dispatch.rs:88-89 `unsafe { out.set_len(len) }`
        if (device_ && n > cap_) throw KernelError(KernelError::InvalidArguments, "set_len beyond a device vector's capacity");
        len_ = n;
    }
    void push(T v) {
        if (owner_) make_owned(len_ + 1);
        if (len_ == cap_) reserve(cap_ ? cap_ * 2 : 16);
        data_[len_++] = v;
    }
    T* data() {  // a device vector is handed out as is: its only writers are kernels
        if (owner_ && !device_) make_owned(len_);
        return data_;
    }
    const T* data() const { return data_; }
    size_t size() const { return len_; }
    bool empty() const { return len_ == 0; }
    T& operator[](size_t i) {
        if (owner_) make_owned(len_);
        return data_[i];
    }
    const T& operator[](size_t i) const {
        if (device_) throw KernelError(KernelError::InvalidArguments, "a device-resident Vec64 cannot be read by the CPU: to_host() first");
        return data_[i];
    }
    const T* begin() const { return data_; }
    const T* end() const { return data_ + len_; }
    bool operator==(const std::vector<T>& o) const {
        if (device_) return to_host() == o;
        if (o.size() != len_) return false;
        for (size_t i = 0; i < len_; ++i)
            if (!(data_[i] == o[i])) return false;
        return true;
    }

  private:
    // copy-on-write: leave the shared region, keep at least `n` elements of room
    void make_owned(size_t n) {
        if (device_) throw KernelError(KernelError::InvalidArguments, "a device-resident Vec64 cannot be modified by the CPU: to_host() first");
        const T* src = data_;
        const size_t keep = len_;
        std::shared_ptr<void> hold = std::move(owner_);
        owner_.reset();
        data_ = nullptr;
        len_ = cap_ = 0;
        reserve(n > keep ? n : keep);
        if (keep) std::memcpy(data_, src, keep * sizeof(T));
        len_ = keep;
    }
    T* data_ = nullptr;
    size_t len_ = 0, cap_ = 0;
    std::shared_ptr<void> owner_;
    bool device_ = false;
};

// Arrow validity bitmap — src/structs/bitmask.rs:66-71. LSB first, 1 = valid, bits >= len zero.
class Bitmask {
  public:
    Vec64<uint8_t> bits;
    size_t len = 0;

    Bitmask() = default;
    // Bitmask::new_set_all — bitmask.rs:94-105
    static Bitmask new_set_all(size_t len, bool set) {
        Bitmask m;
        m.len = len;
        if (detail::result_placement() == Placement::Device) {  // a result bitmap in HBM: the kernel writes every word
            m.bits = Vec64<uint8_t>::on_device(((len + 63) / 64) * 8, ((len + 63) / 64) * 8);
            return m;
        }
        m.bits.resize(((len + 63) / 64) * 8, 0);  // word-padded capacity (the kernels write whole u64 words)
        size_t n_bytes = (len + 7) / 8;
        for (size_t i = 0; i < n_bytes; ++i) m.bits[i] = set ? 0xFF : 0;
        m.mask_trailing_bits();
        return m;
    }
    static Bitmask from_bools(std::initializer_list<bool> v) {
        Bitmask m = new_set_all(v.size(), false);
        size_t i = 0;
        for (bool b : v) m.set(i++, b);
        return m;
    }
    void mask_trailing_bits() {  // bitmask.rs:83-90
        if (len == 0 || (len & 7) == 0) return;
        bits[(len + 7) / 8 - 1] &= (uint8_t)((1u << (len & 7)) - 1);
    }
    bool get(size_t i) const {  // bitmask.rs:745-748
        if (bits.is_device()) throw KernelError(KernelError::InvalidArguments, "Bitmask::get on a device-resident bitmap: to_host() first");
        return (bits[i >> 3] >> (i & 7)) & 1;
    }
    void set(size_t i, bool v) {                                        // bitmask.rs:248-258
        if (v) bits[i >> 3] |= (uint8_t)(1u << (i & 7));
        else bits[i >> 3] &= (uint8_t)~(1u << (i & 7));
    }
    size_t count_ones() const;
    bool all_true() const;
    Bitmask to_device() const {
        Bitmask m;
        m.len = len;
        m.bits = bits.to_device();
        return m;
    }
    Bitmask to_host() const {
        Bitmask m;
        m.len = len;
        m.bits = bits.to_host();
        return m;
    }
};

// BitmaskVT = (&Bitmask, offset, len) — src/aliases.rs:172
struct BitmaskVT {
    const Bitmask* mask;
    size_t offset;
    size_t len;
};
inline BitmaskVT window(const Bitmask& m) { return {&m, 0, m.len}; }

template <typename T>
struct IntegerArray {
    Vec64<T> data;
    std::optional<Bitmask> null_mask;
    bool is_empty() const { return data.empty(); }
};
template <typename T>
struct FloatArray {
    Vec64<T> data;
    std::optional<Bitmask> null_mask;
    bool is_empty() const { return data.empty(); }
};

template <typename T>
struct Slice {
    const T* ptr;
    size_t len;
    Slice(const T* p, size_t n) : ptr(p), len(n) {}
    Slice(const Vec64<T>& v) : ptr(v.data()), len(v.size()) {}
    Slice(const std::vector<T>& v) : ptr(v.data()), len(v.size()) {}
};

// ---- elementwise arithmetic: apply_int_<t>, apply_float_<t>, apply_fma_<t> (dispatch.rs:376-418) -------------
#define MA_HPP_APPLY(FAMILY, TAG, T, ARRAY)                                                                          \
    inline ARRAY<T> apply_##FAMILY##_##TAG(Slice<T> lhs, Slice<T> rhs, ArithmeticOperator op,                        \
                                           const Bitmask* mask = nullptr) {                                         \
        ARRAY<T> out;                                                                                                \
        out.data = Vec64<T>::with_capacity(lhs.len);                                                                 \
        out.data.set_len(lhs.len);                                                                                   \
        if (mask) out.null_mask = Bitmask::new_set_all(lhs.len, true);                                               \
        check(ma_apply_##FAMILY##_##TAG(Context::global().get(), lhs.ptr, lhs.len, rhs.ptr, rhs.len, (int32_t)op,    \
                                        mask ? mask->bits.data() : nullptr, 0, out.data.data(),                      \
                                        mask ? out.null_mask->bits.data() : nullptr));                               \
        return out;                                                                                                  \
    }                                                                                                                \
    /* fused Array (op) Scalar / Scalar (op) Array: routing/broadcast.rs:25-112 without the materialised copy */     \
    inline ARRAY<T> apply_##FAMILY##_##TAG##_scalar_rhs(Slice<T> lhs, T scalar, ArithmeticOperator op,               \
                                                        const Bitmask* mask = nullptr) {                            \
        ARRAY<T> out;                                                                                                \
        out.data = Vec64<T>::with_capacity(lhs.len);                                                                 \
        out.data.set_len(lhs.len);                                                                                   \
        if (mask) out.null_mask = Bitmask::new_set_all(lhs.len, true);                                               \
        check(ma_apply_##FAMILY##_##TAG##_scalar_rhs(Context::global().get(), lhs.ptr, lhs.len, scalar, (int32_t)op, \
                                                     mask ? mask->bits.data() : nullptr, 0, out.data.data(),         \
                                                     mask ? out.null_mask->bits.data() : nullptr));                  \
        return out;                                                                                                  \
    }                                                                                                                \
    inline ARRAY<T> apply_##FAMILY##_##TAG##_scalar_lhs(T scalar, Slice<T> rhs, ArithmeticOperator op,               \
                                                        const Bitmask* mask = nullptr) {                            \
        ARRAY<T> out;                                                                                                \
        out.data = Vec64<T>::with_capacity(rhs.len);                                                                 \
        out.data.set_len(rhs.len);                                                                                   \
        if (mask) out.null_mask = Bitmask::new_set_all(rhs.len, true);                                               \
        check(ma_apply_##FAMILY##_##TAG##_scalar_lhs(Context::global().get(), scalar, rhs.ptr, rhs.len, (int32_t)op, \
                                                     mask ? mask->bits.data() : nullptr, 0, out.data.data(),         \
                                                     mask ? out.null_mask->bits.data() : nullptr));                  \
        return out;                                                                                                  \
    }

MA_HPP_APPLY(int, i32, int32_t, IntegerArray)
MA_HPP_APPLY(int, u32, uint32_t, IntegerArray)
MA_HPP_APPLY(int, i64, int64_t, IntegerArray)
MA_HPP_APPLY(int, u64, uint64_t, IntegerArray)
MA_HPP_APPLY(float, f32, float, FloatArray)
MA_HPP_APPLY(float, f64, double, FloatArray)
#undef MA_HPP_APPLY

#define MA_HPP_FMA(TAG, T)                                                                                           \
    inline FloatArray<T> apply_fma_##TAG(Slice<T> lhs, Slice<T> rhs, Slice<T> acc, const Bitmask* mask = nullptr) {  \
        FloatArray<T> out;                                                                                           \
        out.data = Vec64<T>::with_capacity(lhs.len);                                                                 \
        out.data.set_len(lhs.len);                                                                                   \
        if (mask) out.null_mask = Bitmask::new_set_all(lhs.len, true);                                               \
        check(ma_apply_fma_##TAG(Context::global().get(), lhs.ptr, lhs.len, rhs.ptr, rhs.len, acc.ptr, acc.len,      \
                                 mask ? mask->bits.data() : nullptr, 0, out.data.data(),                             \
                                 mask ? out.null_mask->bits.data() : nullptr));                                      \
        return out;                                                                                                  \
    }
MA_HPP_FMA(f32, float)
MA_HPP_FMA(f64, double)
#undef MA_HPP_FMA

// ---- bitmask kernels (src/kernels/bitmask/dispatch.rs:47-295) ------------------------------------------------
#define MA_HPP_MASK_BINOP(NAME)                                                                                      \
    inline Bitmask NAME(BitmaskVT lhs, BitmaskVT rhs) {                                                              \
        Bitmask out = Bitmask::new_set_all(lhs.len, false);                                                          \
        check(ma_##NAME(Context::global().get(), lhs.mask->bits.data(), lhs.offset, rhs.mask->bits.data(),           \
                        rhs.offset, lhs.len, out.bits.data()));                                                      \
        return out;                                                                                                  \
    }
MA_HPP_MASK_BINOP(and_masks)
MA_HPP_MASK_BINOP(or_masks)
MA_HPP_MASK_BINOP(xor_masks)
MA_HPP_MASK_BINOP(in_mask)
MA_HPP_MASK_BINOP(not_in_mask)
MA_HPP_MASK_BINOP(eq_mask)
MA_HPP_MASK_BINOP(ne_mask)
#undef MA_HPP_MASK_BINOP

inline Bitmask not_mask(BitmaskVT src) {
    Bitmask out = Bitmask::new_set_all(src.len, false);
    check(ma_not_mask(Context::global().get(), src.mask->bits.data(), src.offset, src.len, out.bits.data()));
    return out;
}
inline bool all_eq(BitmaskVT a, BitmaskVT b) {
    int32_t r = 0;
    check(ma_all_eq(Context::global().get(), a.mask->bits.data(), a.offset, b.mask->bits.data(), b.offset, a.len, &r));
    return r != 0;
}
inline bool all_ne(BitmaskVT a, BitmaskVT b) {
    int32_t r = 0;
    check(ma_all_ne(Context::global().get(), a.mask->bits.data(), a.offset, b.mask->bits.data(), b.offset, a.len, &r));
    return r != 0;
}
inline size_t popcount_mask(BitmaskVT m) {
    uint64_t r = 0;
    check(ma_popcount_mask(Context::global().get(), m.mask->bits.data(), m.offset, m.len, &r));
    return (size_t)r;
}
inline bool all_true_mask(const Bitmask& m) {
    int32_t r = 0;
    check(ma_all_true_mask(Context::global().get(), m.bits.data(), m.len, &r));
    return r != 0;
}
inline bool all_false_mask(const Bitmask& m) {
    int32_t r = 0;
    check(ma_all_false_mask(Context::global().get(), m.bits.data(), m.len, &r));
    return r != 0;
}
// merge_bitmasks_to_new — src/kernels/bitmask/mod.rs:171-196
inline std::optional<Bitmask> merge_bitmasks_to_new(const Bitmask* lhs, const Bitmask* rhs, size_t len) {
    if (!lhs && !rhs) return std::nullopt;
    Bitmask out = Bitmask::new_set_all(len, true);
    int32_t some = 0;
    check(ma_merge_bitmasks_to_new(Context::global().get(), lhs ? lhs->bits.data() : nullptr,
                                   rhs ? rhs->bits.data() : nullptr, len, out.bits.data(), &some));
    return out;
}
inline size_t Bitmask::count_ones() const { return popcount_mask({this, 0, len}); }
inline bool Bitmask::all_true() const { return all_true_mask(*this); }

// ---- reductions (benches/benchmark_parallel_simd.rs:44-98) ---------------------------------------------------
inline int64_t sum_i64(Slice<int64_t> s, const Bitmask* mask = nullptr, uint64_t* valid = nullptr) {
    int64_t out = 0;
    check(ma_i64_sum(Context::global().get(), s.ptr, s.len, mask ? mask->bits.data() : nullptr, 0, -1, &out, valid));
    return out;
}
inline double sum_f64(Slice<double> s, const Bitmask* mask = nullptr, uint64_t* valid = nullptr) {
    double out = 0;
    check(ma_f64_sum(Context::global().get(), s.ptr, s.len, mask ? mask->bits.data() : nullptr, 0, -1, &out, valid));
    return out;
}
inline double mean_f64(Slice<double> s, const Bitmask* mask = nullptr) {
    double out = 0;
    check(ma_f64_mean(Context::global().get(), s.ptr, s.len, mask ? mask->bits.data() : nullptr, 0, -1, &out, nullptr));
    return out;
}

}  // namespace ma
