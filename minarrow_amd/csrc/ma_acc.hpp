// Reduction accumulators shared by the sum kernels (ma_reduce.hip, ma_reduce_batch.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "ma_common.hpp"

namespace ma {

// ------------------------------------------------------------------------------------------------
// Accumulators
// ------------------------------------------------------------------------------------------------

// Wrapping 64-bit integer accumulator. i32 sign-extends, u32 zero-extends; i64/u64 pass through.
struct IntAcc {
    uint64_t s;
    __device__ __forceinline__ void init() { s = 0; }
    template <typename T>
    __device__ __forceinline__ void add(T v) {
        s += (uint64_t)(int64_t)v;  // (int64_t) of an unsigned 32-bit value is its zero extension
    }
    __device__ __forceinline__ void merge(const IntAcc& o) { s += o.s; }
    __device__ __forceinline__ void shfl_down_merge(int off) {
        s += (uint64_t)__shfl_down((unsigned long long)s, off, 64);
    }
    __device__ __forceinline__ void to_partial(Partial& p) const {
        p.a = s;
        p.b = 0;
    }
    __device__ __forceinline__ void from_words(uint64_t a, uint64_t) { s = a; }
};

// 32-bit integers, for a FEW THOUSAND adds at most (one chunk of a chunked column): the halves of a value go to two 32-bit words
// — `lo` takes bits 0..15 (zero-extended), `hi` bits 16..31 (sign- or zero-extended as T is) — two
// full-rate adds with a sub-dword operand (v_add_u32_sdwa ... src1_sel:WORD_0 / sext(WORD_1)) where IntAcc::add is a shift, a
// register copy and a 64-bit add per value. 65 535 adds cannot overflow either word (65 535 x 65 535 < 2^32, 65 535 x 32 768 < 2^31);
// the owner widens (and starts over) long before: ma_reduce_batch.hip's column_waves_kernel, per chunk.
template <typename T>
struct SplitAcc {
    static_assert(sizeof(T) == 4 && std::is_integral<T>::value, "32-bit integers");
    typedef typename std::conditional<std::is_signed<T>::value, int32_t, uint32_t>::type Hi;
    uint32_t lo;
    Hi hi;
    __device__ __forceinline__ void init() {
        lo = 0;
        hi = 0;
    }
    __device__ __forceinline__ void add(T v) {
        // lo += v & 0xffff; hi += v >> 16 (arithmetic for a signed T) — said in the instruction's own words: left to itself the
        // compiler extracts both halves first (v_and, v_ashrrev) and pairs the adds (v_add3), three instructions per value again
        asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(lo) : "v"(lo), "v"(v));
        if constexpr (std::is_signed<T>::value)
            asm("v_add_u32_sdwa %0, %1, sext(%2) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(hi) : "v"(hi), "v"(v));
        else
            asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(hi) : "v"(hi), "v"(v));
    }
    // add(valid ? v : 0) for a validity bit `valid` (0 or 1) without the select: the bit is the other factor of two 16-bit
    // multiply-adds (lo += v.lo16 * valid; hi += v.hi16 * valid) — a bit extract and two instructions per value where a mask
    // and two adds are four
    __device__ __forceinline__ void add_if(T v, unsigned valid) {
        asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[0,0,0,0]" : "=v"(lo) : "v"(v), "v"(valid), "v"(lo));
        if constexpr (std::is_signed<T>::value) asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(hi) : "v"(v), "v"(valid), "v"(hi));
        else asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(hi) : "v"(v), "v"(valid), "v"(hi));
    }
    __device__ __forceinline__ void merge(const SplitAcc& o) {
        lo += o.lo;
        hi += o.hi;
    }
    __device__ __forceinline__ IntAcc widen() const {
        IntAcc a;
        a.s = (uint64_t)(int64_t)hi * 65536u + lo;  // wrapping; (int64_t) of an unsigned Hi is its zero extension
        return a;
    }
};

// Double-double accumulator: hi + lo carries the running sum to ~106 bits.
struct DDAcc {
    double hi, lo;
    __device__ __forceinline__ void init() {
        hi = 0.0;
        lo = 0.0;
    }
    // Knuth two-sum: t + e == hi + v exactly.
    __device__ __forceinline__ void add_d(double v) {
        double t = hi + v;
        double bp = t - hi;
        double e = (hi - (t - bp)) + (v - bp);
        hi = t;
        lo += e;
    }
    template <typename T>
    __device__ __forceinline__ void add(T v) {
        add_d((double)v);
    }
    __device__ __forceinline__ void merge(const DDAcc& o) {
        double t = hi + o.hi;
        double bp = t - hi;
        double e = (hi - (t - bp)) + (o.hi - bp);
        hi = t;
        lo += e + o.lo;
    }
    __device__ __forceinline__ void shfl_down_merge(int off) {
        DDAcc o;
        o.hi = __shfl_down(hi, off, 64);
        o.lo = __shfl_down(lo, off, 64);
        merge(o);
    }
    // Renormalise so that hi is the correctly rounded value of hi + lo (fast two-sum).
    __device__ __forceinline__ void normalise() {
        // Inf/NaN in the data (or an overflow) poisons `lo` with NaN while `hi` already holds the IEEE
        // answer a plain sum would give; only a finite pair is renormalised.
        if (isfinite(hi) && isfinite(lo)) {
            double t = hi + lo;
            lo = lo - (t - hi);
            hi = t;
        } else {
            lo = 0.0;
        }
    }
    __device__ __forceinline__ void to_partial(Partial& p) const {
        p.a = (uint64_t)__double_as_longlong(hi);
        p.b = (uint64_t)__double_as_longlong(lo);
    }
    __device__ __forceinline__ void from_words(uint64_t a, uint64_t b) {
        hi = __longlong_as_double((long long)a);
        lo = __longlong_as_double((long long)b);
    }
};

template <typename T>
struct AccOf {
    typedef IntAcc type;
};
template <>
struct AccOf<double> {
    typedef DDAcc type;
};
template <>
struct AccOf<float> {
    typedef DDAcc type;
};

}  // namespace ma
