// Reduction accumulators shared by the sum kernels (ma_reduce.hip, ma_reduce_batch.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

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
