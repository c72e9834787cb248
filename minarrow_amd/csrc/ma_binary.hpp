// Elementwise binary arithmetic and FMA for gfx950: kernel templates + host dispatch, instantiated once per
// element type by ma_binary_<t>.hip.
//
// Replaces (reference paths relative to the minarrow repository root):
//   apply_int_{i32,u32,i64,u64,...}   src/kernels/arithmetic/dispatch.rs:65-133, :376-387
//   apply_float_{f32,f64}             src/kernels/arithmetic/dispatch.rs:138-206, :389-402
//   apply_fma_{f32,f64}               src/kernels/arithmetic/dispatch.rs:211-290, :404-418
//   int_dense_body_{std,simd}         src/kernels/arithmetic/std.rs:41-80,   simd.rs:52-113
//   int_masked_body_{std,simd}        src/kernels/arithmetic/std.rs:86-138,  simd.rs:118-370
//   float_{dense,masked}_body_*       src/kernels/arithmetic/std.rs:144-194, simd.rs:376-589
//   fma_{dense,masked}_body_*         src/kernels/arithmetic/std.rs:196-230, simd.rs:591-751
//   maybe_broadcast_scalar_array      src/kernels/routing/broadcast.rs:25-112 (fused here: the scalar is a
//                                     kernel argument instead of a materialised vec64![x; n])
//
// Two kernels per operation:
//   vec kernel  — the bandwidth path. A wave owns a contiguous run; each lane moves 16 bytes per access
//                 (global_load_dwordx4 / global_store_dwordx4, non-temporal), UNROLL accesses per operand in
//                 flight before the first use. `head` rows are peeled so that the stores are 16-byte aligned; the
//                 inputs may sit on any element-aligned phase (load16u).
//   row kernel  — one row per lane, for the unaligned head / ragged tail of the vec kernel.
// Output validity of a masked op equals the input validity: a word-wise funnel-shift copy (mask_copy_kernel) writes
// it, so the vec kernel only READS validity (to zero the null slots). The exception is masked integer
// Div/Rem/FloorDiv, whose output validity is (valid && divisor != 0): the vec kernel packs each lane's result bits
// into validity words (pack_lane_bits), the row kernel uses one wave64 ballot per 64 rows.
#pragma once

#include "ma_device.hpp"

namespace ma {

enum : int { kAA = 0, kAS = 1, kSA = 2 };  // array⊕array, array⊕scalar, scalar⊕array

// ------------------------------------------------------------------------------------------------
// Element functions. `dz` is set when an integer divisor is zero (value 0 is returned, as the masked
// reference bodies store; the dense reference bodies panic — the host turns the latch into a status).
// ------------------------------------------------------------------------------------------------
template <typename T, bool IS_FLOAT = std::is_floating_point<T>::value>
struct Elem;

template <typename T>
struct Elem<T, false> {
    typedef typename std::make_unsigned<T>::type U;
    static constexpr bool kSigned = std::is_signed<T>::value;

    // rhs.to_u32().unwrap_or(0) — std.rs:67
    static __device__ __forceinline__ uint32_t exponent(T e) {
        if (kSigned && e < (T)0) return 0u;
        if ((uint64_t)e > 0xFFFFFFFFull) return 0u;
        return (uint32_t)e;
    }
    // x.pow(e) with wrapping multiplies (release-mode Rust; equals simd.rs:94-101's repeated wrapping_mul)
    static __device__ __forceinline__ T pow(T base, uint32_t e) {
        U acc = 1, b = (U)base;
        while (e) {
            if (e & 1u) acc = (U)(acc * b);
            b = (U)(b * b);
            e >>= 1;
        }
        return (T)acc;
    }
    static __device__ __forceinline__ T div(T a, T b) {
        if constexpr (sizeof(T) == 1) {
            // 8-bit quotients through one reciprocal instead of the ~30-instruction 32-bit division sequence: a * rcp(b)
            // carries <= 2 ulp (2.4e-7 relative); a non-integer quotient sits >= 1/255 away from the next integer, so
            // it cannot cross it, and an exact integer quotient is kept on the right side of the truncation by scaling
            // with 1 + 2^-20. Exhaustively checked over all 65 536 operand pairs of both types
            // (tests/test_gpu_arith.py::test_8bit_division_exhaustive). MIN / -1 gives 128, which wraps to MIN.
            const float q = (float)(int)a * __builtin_amdgcn_rcpf((float)(int)b) * 1.00000095367431640625f;
            return (T)(int)q;  // truncation toward zero
        } else {
            if (kSigned && b == (T)-1) return (T)((U)0 - (U)a);  // MIN / -1 wraps to MIN like the SIMD lanes
            return (T)(a / b);
        }
    }
    static __device__ __forceinline__ T rem(T a, T b) {
        if constexpr (sizeof(T) == 1) {
            return (T)((int)a - (int)div(a, b) * (int)b);  // MIN % -1: -128 - (-128)(-1) wraps to 0
        } else {
            if (kSigned && b == (T)-1) return (T)0;
            return (T)(a % b);
        }
    }
    template <int OP>
    static __device__ __forceinline__ T apply(T a, T b, bool& dz) {
        if constexpr (OP == MA_OP_ADD) return (T)((U)a + (U)b);
        if constexpr (OP == MA_OP_SUBTRACT) return (T)((U)a - (U)b);
        if constexpr (OP == MA_OP_MULTIPLY) return (T)((U)a * (U)b);
        if constexpr (OP == MA_OP_POWER) return pow(a, exponent(b));
        if constexpr (OP == MA_OP_DIVIDE || OP == MA_OP_REMAINDER || OP == MA_OP_FLOORDIV) {
            if (b == (T)0) {
                dz = true;
                return (T)0;
            }
            if constexpr (OP == MA_OP_DIVIDE) return div(a, b);
            if constexpr (OP == MA_OP_REMAINDER) return rem(a, b);
            // FloorDiv — std.rs:68-77
            T d = div(a, b), r = rem(a, b);
            if (kSigned && r != (T)0 && ((T)(a ^ b)) < (T)0) return (T)((U)d - (U)1);
            return d;
        }
        return (T)0;
    }
    static __device__ __forceinline__ T apply_rt(int op, T a, T b, bool& dz) {
        switch (op) {
            case MA_OP_ADD: return apply<MA_OP_ADD>(a, b, dz);
            case MA_OP_SUBTRACT: return apply<MA_OP_SUBTRACT>(a, b, dz);
            case MA_OP_MULTIPLY: return apply<MA_OP_MULTIPLY>(a, b, dz);
            case MA_OP_DIVIDE: return apply<MA_OP_DIVIDE>(a, b, dz);
            case MA_OP_REMAINDER: return apply<MA_OP_REMAINDER>(a, b, dz);
            case MA_OP_POWER: return apply<MA_OP_POWER>(a, b, dz);
            default: return apply<MA_OP_FLOORDIV>(a, b, dz);
        }
    }
};

// ln for the f64 Power path: max 0.518 ULP, mean 0.247 ULP, correctly rounded on 99.86 % of 106 931 inputs (subnormals, a
// cloud around 1, every power of two and its neighbours) against 265-bit decimal arithmetic — glibc's log on the same
// inputs: 0.608 / 0.247 / 99.67 % — in a third of the instructions of the library routine. Reproduce:
// tools/check_pow_series.py -> tests/golden/pow_series_kat.npz -> tests/test_gpu_pow_series.py (bounds), tools/
// pow_series_report.py -> profiles/r03_pow_series_accuracy.json (figures). ln x = e ln2 + 2 atanh(s), s = (m - 1) / (m + 1) on m in [1/sqrt2, sqrt2):
// the quotient is carried as s_hi + s_lo (the residual f - s_hi (d + d_lo) comes out of one fma), the series runs to
// s^23, and the one rounding that matters — e ln2_hi + 2 s_hi — is compensated (Fast2Sum).
__device__ __forceinline__ double pow_f64_ln(double x) {
    double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1), subnormals included
    int e = __builtin_amdgcn_frexp_exp(x);
    if (m < 0.70710678118654752) {
        m += m;
        e -= 1;
    }
    const double f = m - 1.0;           // exact
    const double d = 2.0 + f;           // rounded; d_lo is what the rounding dropped (exact: 2 >= |f|)
    const double d_lo = (2.0 - d) + f;
    double r = __builtin_amdgcn_rcp(d);  // 2^-24.4
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    const double s_hi = f * r;
    const double s_lo = (fma(-s_hi, d, f) - s_hi * d_lo) * r;
    const double z = s_hi * s_hi;
    double p = 1.0 / 23;
    p = fma(p, z, 1.0 / 21);
    p = fma(p, z, 1.0 / 19);
    p = fma(p, z, 1.0 / 17);
    p = fma(p, z, 1.0 / 15);
    p = fma(p, z, 1.0 / 13);
    p = fma(p, z, 1.0 / 11);
    p = fma(p, z, 1.0 / 9);
    p = fma(p, z, 1.0 / 7);
    p = fma(p, z, 1.0 / 5);
    p = fma(p, z, 1.0 / 3);
    const double t = s_hi * z * p;  // atanh(s) - s
    const double ed = (double)e;
    const double small = fma(ed, 1.90821492927058770002e-10, 2.0 * (s_lo + t));
    const double big = ed * 6.93147180369123816490e-01;  // ln2_hi has 21 trailing zero bits: exact for |e| < 2^11
    const double two_s = s_hi + s_hi;
    const double a = big + two_s;
    const double a_err = (big - a) + two_s;
    double res = a + (a_err + small);
    if (x == __builtin_inf()) res = x;
    if (x == 0.0) res = -__builtin_inf();
    if (!(x >= 0.0)) res = __builtin_nan("");  // negative or NaN
    return res;
}

template <>
struct Elem<double, true> {
    template <int OP>
    static __device__ __forceinline__ double apply(double a, double b, bool&) {
        if constexpr (OP == MA_OP_ADD) return a + b;
        if constexpr (OP == MA_OP_SUBTRACT) return a - b;
        if constexpr (OP == MA_OP_MULTIPLY) return a * b;
        if constexpr (OP == MA_OP_DIVIDE) return a / b;
        if constexpr (OP == MA_OP_REMAINDER) return fmod(a, b);       // Rust `%` on floats = C fmod
        if constexpr (OP == MA_OP_POWER) return exp(b * pow_f64_ln(a));  // std.rs:153: (rhs * lhs.ln()).exp()
        if constexpr (OP == MA_OP_FLOORDIV) return floor(a / b);
        return 0.0;
    }
    static __device__ __forceinline__ double apply_rt(int op, double a, double b, bool& dz) {
        switch (op) {
            case MA_OP_ADD: return apply<MA_OP_ADD>(a, b, dz);
            case MA_OP_SUBTRACT: return apply<MA_OP_SUBTRACT>(a, b, dz);
            case MA_OP_MULTIPLY: return apply<MA_OP_MULTIPLY>(a, b, dz);
            case MA_OP_DIVIDE: return apply<MA_OP_DIVIDE>(a, b, dz);
            case MA_OP_REMAINDER: return apply<MA_OP_REMAINDER>(a, b, dz);
            case MA_OP_POWER: return apply<MA_OP_POWER>(a, b, dz);
            default: return apply<MA_OP_FLOORDIV>(a, b, dz);
        }
    }
    static __device__ __forceinline__ double fma3(double a, double b, double c) { return fma(a, b, c); }
};

// ln and exp for the f32 Power path, evaluated in f64 to ~2^-45 relative — far inside the half-ULP of the f32 value they
// are rounded to, at a third of the instructions of the 1-ULP f64 routines (atanh series to s^15 on [1/sqrt2, sqrt2),
// Taylor to r^11 on |r| <= ln2/2). Rounded to f32, both ARE the correctly rounded logf / expf values on every one of
// 60 000 + 67 500 test inputs (subnormal to overflow), and f32 Power end to end equals the exact three-rounding formulation
// RN32(exp(RN32(b * RN32(ln a)))) on all 60 000 test pairs (same fixture, same tests as above).
__device__ __forceinline__ double pow_f32_ln(double x) {
    double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
    if (m < 0.70710678118654752) {
        m += m;
        e -= 1;
    }
    const double f = m - 1.0;
    const double d = 2.0 + f;
    double r = __builtin_amdgcn_rcp(d);  // 2^-24.4 on gfx950 (tools/probe_rcp_f64.hip); one Newton step: 2^-48.7
    r = fma(fma(-d, r, 1.0), r, r);
    const double s = f * r;  // ln m = 2 atanh(s)
    const double z = s * s;
    double p = 1.0 / 15;
    p = fma(p, z, 1.0 / 13);
    p = fma(p, z, 1.0 / 11);
    p = fma(p, z, 1.0 / 9);
    p = fma(p, z, 1.0 / 7);
    p = fma(p, z, 1.0 / 5);
    p = fma(p, z, 1.0 / 3);
    p = p * z;
    const double two_s = s + s;
    return fma((double)e, 0.6931471805599453, fma(two_s, p, two_s));  // x = 0, inf, negative, NaN: the caller's business
}
__device__ __forceinline__ double pow_f32_exp(double yc) {  // |yc| <= 150 (clamped by the caller), not NaN
    const double k = __builtin_rint(yc * 1.4426950408889634);
    double r = fma(-k, 0.6931471803691238, yc);  // ln2 split: the high part's product with |k| < 2^10 is exact
    r = fma(-k, 1.9082149292705877e-10, r);
    double p = 1.0 / 39916800.0;
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return __builtin_amdgcn_ldexp(p, (int)k);
}

template <>
struct Elem<float, true> {
    template <int OP>
    static __device__ __forceinline__ float apply(float a, float b, bool&) {
        if constexpr (OP == MA_OP_ADD) return a + b;
        if constexpr (OP == MA_OP_SUBTRACT) return a - b;
        if constexpr (OP == MA_OP_MULTIPLY) return a * b;
        if constexpr (OP == MA_OP_DIVIDE) return a / b;
        if constexpr (OP == MA_OP_REMAINDER) return fmodf(a, b);
        // (rhs * lhs.ln()).exp() — std.rs:153. The reference's ln/exp are the host libm's logf/expf, which are
        // correctly rounded in all but a handful of cases; evaluating both through f64 and rounding at the same
        // three points (ln, product, exp) reproduces those bits instead of adding a second set of libm errors
        // that the exp() would amplify by |b ln a|. The f64 evaluations only need to be good to the f32 rounding:
        // pow_f32_ln / pow_f32_exp above.
        if constexpr (OP == MA_OP_POWER) {
            // the special operands are sorted out in f32 (one-register compares and selects)
            float l = (float)pow_f32_ln((double)a);
            l = a == __builtin_inff() ? a : l;
            l = a == 0.0f ? -__builtin_inff() : l;
            l = a >= 0.0f ? l : __builtin_nanf("");  // negative or NaN
            const float y = b * l;
            const float yc = __builtin_fminf(__builtin_fmaxf(y, -150.0f), 150.0f);  // far beyond the f32 range; NaN -> -150
            const float r = (float)pow_f32_exp((double)yc);
            return y != y ? y : r;
        }
        if constexpr (OP == MA_OP_FLOORDIV) return floorf(a / b);
        return 0.0f;
    }
    static __device__ __forceinline__ float apply_rt(int op, float a, float b, bool& dz) {
        switch (op) {
            case MA_OP_ADD: return apply<MA_OP_ADD>(a, b, dz);
            case MA_OP_SUBTRACT: return apply<MA_OP_SUBTRACT>(a, b, dz);
            case MA_OP_MULTIPLY: return apply<MA_OP_MULTIPLY>(a, b, dz);
            case MA_OP_DIVIDE: return apply<MA_OP_DIVIDE>(a, b, dz);
            case MA_OP_REMAINDER: return apply<MA_OP_REMAINDER>(a, b, dz);
            case MA_OP_POWER: return apply<MA_OP_POWER>(a, b, dz);
            default: return apply<MA_OP_FLOORDIV>(a, b, dz);
        }
    }
    static __device__ __forceinline__ float fma3(float a, float b, float c) { return fmaf(a, b, c); }
};

// ------------------------------------------------------------------------------------------------
// Kernel arguments
// ------------------------------------------------------------------------------------------------
template <typename T>
struct BinArgs {
    const T* lhs;           // element pointers of row 0 (nullptr for a scalar side)
    const T* rhs;
    const T* acc;           // FMA addend (nullptr otherwise)
    T* out;
    T scalar;               // the scalar side's value
    size_t n;               // rows
    size_t head;            // rows before the first 16-byte boundary (vec kernel starts here)
    size_t n_tiles;         // full workgroup tiles the vec kernel covers
    const uint64_t* words;  // input validity (8-byte aligned base) or nullptr
    size_t bit_off;         // bit index of row 0 relative to `words`
    size_t last_word;       // last word index holding a window bit
    const uint64_t* words2; // optional SECOND input validity, combined with the first per row: apply_datetime's AND
    size_t bit_off2;        //   (merge_bitmasks_to_new, dispatch.rs:336-341) or the chunk pair's OR (Bitmask::union,
    size_t last_word2;      //   broadcast/super_array.rs:224) — fused here instead of a merged temporary
    int combine_and;        // 1: words & words2, 0: words | words2
    uint64_t* out_words;    // output validity words (data-dependent validity only) or nullptr
    uint32_t* flags;        // device latch: bit 0 = integer divide by zero seen in a dense kernel
    int op;                 // row kernel: runtime ArithmeticOperator
    int kind;               // row kernel: kAA / kAS / kSA
    int ballot_mask;        // row kernel: 1 = also produce the out_words of its rows, one ballot per 64 rows
};

// ------------------------------------------------------------------------------------------------
// vec kernel
// ------------------------------------------------------------------------------------------------
// Element k of a loaded 16-byte vector. 1-byte elements are taken out of the dwords by hand: with a <16 x i8> value between
// the load and its use the optimiser dropped the loads' non-temporal hint (every i8 / u8 kernel read with plain loads).
template <typename T, typename VL>
__device__ __forceinline__ T loaded_elem(const VL& v, int k) {
    if constexpr (sizeof(T) == 1) return (T)(uint8_t)(v[k >> 2] >> (8 * (k & 3)));
    else return (T)v[k];
}

template <typename T, int OP, int KIND, bool MASKED, int UNROLL, bool NTS = true>
__global__ __launch_bounds__(kBlock) void binary_vec_kernel(BinArgs<T> a) {
    typedef typename Vec16<T>::type V;
    typedef typename std::conditional<sizeof(T) == 1, MaU4, V>::type VL;  // what a load yields
    constexpr int R = 16 / (int)sizeof(T);
    constexpr int WPT = R * UNROLL;
    constexpr size_t WAVE_ROWS = (size_t)64 * R * UNROLL;
    constexpr size_t TILE_ROWS = WAVE_ROWS * kWaves;
    // Masked integer Div/Rem/FloorDiv: the output validity depends on the data (a zero divisor nulls the row), so
    // this kernel also produces the validity words instead of the mask-copy kernel.
    constexpr bool DATA_VALIDITY = MASKED && std::is_integral<T>::value &&
                                   (OP == MA_OP_DIVIDE || OP == MA_OP_REMAINDER || OP == MA_OP_FLOORDIV);
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bool dz = false;

    for (size_t t = blockIdx.x; t < a.n_tiles; t += gridDim.x) {
        const size_t row0 = a.head + t * TILE_ROWS + (size_t)wave * WAVE_ROWS;
        VL va[UNROLL], vb[UNROLL];
        if constexpr (KIND != kSA) {
            const VL* __restrict__ p = (const VL*)(a.lhs + row0) + lane;
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) va[u] = load16u<VL, true>(p + (size_t)u * 64);
        }
        if constexpr (KIND != kAS) {
            const VL* __restrict__ q = (const VL*)(a.rhs + row0) + lane;
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) vb[u] = load16u<VL, true>(q + (size_t)u * 64);
        }
        RunWords<MASKED ? WPT : 1> aw;
        if constexpr (MASKED) {
            aw.load(a.words, a.bit_off + row0, a.last_word, lane);
            if (a.words2) {  // wave-uniform
                RunWords<WPT> bw;
                bw.load(a.words2, a.bit_off2 + row0, a.last_word2, lane);
                aw.combine(bw, a.combine_and != 0);
            }
        }
        V* __restrict__ o = (V*)(a.out + row0) + lane;
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            unsigned bits = ~0u;
            if constexpr (MASKED) bits = aw.template bits<R>(u, lane);
            unsigned out_bits = bits;
            V r;
            if constexpr (sizeof(T) == 1 && (OP == MA_OP_ADD || OP == MA_OP_SUBTRACT)) {
                // 1-byte wrapping add / subtract, four elements per 32-bit operation (carries cut at the byte borders:
                // the low 7 bits add / subtract freely, the top bit is patched in by XOR) instead of sixteen 16-bit
                // operations with sub-dword selects per vector.
                const unsigned splat = (unsigned)(uint8_t)a.scalar * 0x01010101u;
                const U32x4 ss = {splat, splat, splat, splat};
                U32x4 x = ss, y = ss;
                if constexpr (KIND != kSA) x = __builtin_bit_cast(U32x4, va[u]);
                if constexpr (KIND != kAS) y = __builtin_bit_cast(U32x4, vb[u]);
                constexpr unsigned H = 0x80808080u, L = 0x7f7f7f7fu;
                U32x4 z;
                if constexpr (OP == MA_OP_ADD) z = ((x & L) + (y & L)) ^ ((x ^ y) & H);
                else z = ((x | H) - (y & L)) ^ ((x ^ ~y) & H);
                r = __builtin_bit_cast(V, z);
            } else {
#pragma unroll
            for (int k = 0; k < R; ++k) {
                T x = KIND == kSA ? a.scalar : loaded_elem<T>(va[u], k);
                T y = KIND == kAS ? a.scalar : loaded_elem<T>(vb[u], k);
                bool dzk = false;
                T v = Elem<T>::template apply<OP>(x, y, dzk);
                dz |= dzk;
                // null slots hold 0 (simd.rs:315); 1- and 2-byte vectors are masked as a whole below
                if constexpr (MASKED && sizeof(T) > 2) v = ((bits >> k) & 1u) ? v : (T)0;
                if constexpr (DATA_VALIDITY) out_bits &= ~((dzk ? 1u : 0u) << k);  // m & !div_zero (simd.rs:319-326)
                r[k] = v;
            }
            }
            if constexpr (MASKED && sizeof(T) <= 2) r = zero_null_slots<V, (int)sizeof(T)>(r, bits);
            store16<V, NTS>(o + (size_t)u * 64, r);
            if constexpr (DATA_VALIDITY) {
                // head == 0 here (host dispatch), so a step's 64*R rows are exactly R output validity words.
                constexpr int LPW = 64 / R;
                const uint64_t word = pack_lane_bits<R>(out_bits & ((1u << R) - 1u), lane);
                if (lane % LPW == 0) a.out_words[(row0 >> 6) + (size_t)u * R + lane / LPW] = word;
            }
        }
    }
    if constexpr (!MASKED && std::is_integral<T>::value &&
                  (OP == MA_OP_DIVIDE || OP == MA_OP_REMAINDER || OP == MA_OP_FLOORDIV)) {
        if (__any(dz) && lane == 0) atomicOr(a.flags, 1u);
    }
}

// FMA vec kernel: out = fma(lhs, rhs, acc)
template <typename T, bool MASKED, int UNROLL>
__global__ __launch_bounds__(kBlock) void fma_vec_kernel(BinArgs<T> a) {
    typedef typename Vec16<T>::type V;
    constexpr int R = 16 / (int)sizeof(T);
    constexpr int WPT = R * UNROLL;
    constexpr size_t WAVE_ROWS = (size_t)64 * R * UNROLL;
    constexpr size_t TILE_ROWS = WAVE_ROWS * kWaves;
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (size_t t = blockIdx.x; t < a.n_tiles; t += gridDim.x) {
        const size_t row0 = a.head + t * TILE_ROWS + (size_t)wave * WAVE_ROWS;
        V va[UNROLL], vb[UNROLL], vc[UNROLL];
        const V* __restrict__ p = (const V*)(a.lhs + row0) + lane;
        const V* __restrict__ q = (const V*)(a.rhs + row0) + lane;
        const V* __restrict__ c = (const V*)(a.acc + row0) + lane;
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) va[u] = load16u<V, true>(p + (size_t)u * 64);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) vb[u] = load16u<V, true>(q + (size_t)u * 64);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) vc[u] = load16u<V, true>(c + (size_t)u * 64);
        RunWords<MASKED ? WPT : 1> aw;
        if constexpr (MASKED) aw.load(a.words, a.bit_off + row0, a.last_word, lane);
        V* __restrict__ o = (V*)(a.out + row0) + lane;
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            unsigned bits = ~0u;
            if constexpr (MASKED) bits = aw.template bits<R>(u, lane);
            V r;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                T v = Elem<T>::fma3((T)va[u][k], (T)vb[u][k], (T)vc[u][k]);
                if constexpr (MASKED) v = ((bits >> k) & 1u) ? v : (T)0;
                r[k] = v;
            }
            store16<V, true>(o + (size_t)u * 64, r);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// row kernel: one row per lane; a wave owns 64 consecutive rows = one validity word of the output.
// Processes rows [0, head) and [tail_start, n). With ballot_mask == 1 (masked integer Div/Rem/FloorDiv) head is 0 and
// it also produces the output validity words of its rows by ballot (the vec kernel wrote the ones in front).
// ------------------------------------------------------------------------------------------------
template <typename T, bool MASKED, bool FMA>
__global__ __launch_bounds__(kBlock) void binary_row_kernel(BinArgs<T> a, size_t tail_start) {
    const unsigned lane = threadIdx.x & 63;
    const size_t wave_id = ((size_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * kBlock) >> 6;
    const size_t n_words = (a.n + 63) >> 6;
    // Only the words that hold ragged rows are visited: [0, head_words) and [tail_word0, n_words).
    const size_t head_words = a.ballot_mask ? 0 : (a.head + 63) >> 6;
    size_t tail_word0 = tail_start >> 6;  // ballot mode: the vec kernel wrote the validity words in front of it
    if (tail_word0 < head_words) tail_word0 = head_words;
    const size_t n_visit = head_words + (n_words - tail_word0);
    bool dz_any = false;
    for (size_t k = wave_id; k < n_visit; k += n_waves) {
        const size_t w = k < head_words ? k : tail_word0 + (k - head_words);
        const size_t row = w * 64 + lane;
        bool in_range = row < a.n && (row < a.head || row >= tail_start);
        bool valid = in_range;
        bool dz = false;
        if (in_range) {
            if constexpr (MASKED) {
                valid = row_bit(a.words, a.bit_off + row) != 0;
                if (a.words2) {
                    const bool v2 = row_bit(a.words2, a.bit_off2 + row) != 0;
                    valid = a.combine_and ? (valid && v2) : (valid || v2);
                }
            }
            T v;
            if constexpr (FMA) {
                v = Elem<T>::fma3(a.lhs[row], a.rhs[row], a.acc[row]);
            } else {
                T x = a.kind == kSA ? a.scalar : a.lhs[row];
                T y = a.kind == kAS ? a.scalar : a.rhs[row];
                v = Elem<T>::apply_rt(a.op, x, y, dz);
            }
            if constexpr (MASKED) v = valid ? v : (T)0;
            a.out[row] = v;
        }
        if constexpr (MASKED) {
            if (a.ballot_mask) {
                // out-mask = source validity && !div_zero (simd.rs:319-326); rows >= n contribute 0 bits.
                unsigned long long word = __ballot(valid && !dz);
                if (lane == 0) a.out_words[w] = word;
            }
        } else {
            dz_any |= dz;
        }
    }
    if constexpr (!MASKED && !FMA && std::is_integral<T>::value) {
        if (__any(dz_any) && lane == 0) atomicOr(a.flags, 1u);
    }
}

// Output validity = input validity window re-based to bit 0 (bits >= n zero). Defined in ma_bitmask.hip.
ma_status launch_mask_copy(ma_ctx* ctx, const uint64_t* words, size_t bit_off, size_t n, uint64_t* out_words);
// Output validity = (window 1) AND / OR (window 2), re-based to bit 0 (bits >= n zero). Defined in ma_bitmask.hip.
ma_status launch_mask_combine(ma_ctx* ctx, const uint64_t* w1, size_t off1, const uint64_t* w2, size_t off2, size_t n,
                              bool is_and, uint64_t* out_words);

// ------------------------------------------------------------------------------------------------
// Host dispatch
// ------------------------------------------------------------------------------------------------
template <typename T>
struct BinaryCall {
    int op = 0;
    int kind = kAA;
    const T* lhs = nullptr;
    size_t lhs_len = 0;
    const T* rhs = nullptr;
    size_t rhs_len = 0;
    const T* acc = nullptr;  // FMA only
    size_t acc_len = 0;
    bool fma = false;
    T scalar = T();
    const uint8_t* mask_bits = nullptr;
    size_t mask_bit_offset = 0;
    const uint8_t* mask2_bits = nullptr;  // optional second validity, combined per row with the first (AND / OR)
    size_t mask2_bit_offset = 0;
    bool combine_and = true;
    T* out = nullptr;
    uint8_t* out_mask_bits = nullptr;
};

// Unroll of the vec kernel: 8 or 4 sixteen-byte accesses per operand per lane. Masked kernels read R * UNROLL validity
// words per wave run (RunWords: up to 128, i.e. 1-byte types at 8 — two wave instructions).
template <typename T>
constexpr int clamp_unroll(int unroll, bool masked) {
    (void)masked;
    return unroll == 8 ? 8 : 4;
}

template <typename T, int OP, int KIND, bool MASKED>
static void launch_vec(ma_ctx* ctx, const BinArgs<T>& a, int grid, int unroll) {
    if constexpr (MA_TUNING) {  // the 4-deep tile is a tuning form (ctx variant unroll = 4): not in the shipped library
        if (unroll != 8) {
            hipLaunchKernelGGL((binary_vec_kernel<T, OP, KIND, MASKED, 4>), dim3(grid), dim3(kBlock), 0, ctx->stream, a);
            return;
        }
    }
    hipLaunchKernelGGL((binary_vec_kernel<T, OP, KIND, MASKED, 8>), dim3(grid), dim3(kBlock), 0, ctx->stream, a);
}

template <typename T, int OP, bool MASKED>
static void launch_vec_kind(ma_ctx* ctx, const BinArgs<T>& a, int grid, int unroll, int kind) {
    switch (kind) {
        case kAS: launch_vec<T, OP, kAS, MASKED>(ctx, a, grid, unroll); break;
        case kSA: launch_vec<T, OP, kSA, MASKED>(ctx, a, grid, unroll); break;
        default: launch_vec<T, OP, kAA, MASKED>(ctx, a, grid, unroll); break;
    }
}

template <typename T, bool MASKED>
static void launch_vec_op(ma_ctx* ctx, const BinArgs<T>& a, int grid, int unroll, int kind, int op) {
    switch (op) {
        case MA_OP_ADD: launch_vec_kind<T, MA_OP_ADD, MASKED>(ctx, a, grid, unroll, kind); break;
        case MA_OP_SUBTRACT: launch_vec_kind<T, MA_OP_SUBTRACT, MASKED>(ctx, a, grid, unroll, kind); break;
        case MA_OP_MULTIPLY: launch_vec_kind<T, MA_OP_MULTIPLY, MASKED>(ctx, a, grid, unroll, kind); break;
        case MA_OP_POWER: launch_vec_kind<T, MA_OP_POWER, MASKED>(ctx, a, grid, unroll, kind); break;
        case MA_OP_DIVIDE: launch_vec_kind<T, MA_OP_DIVIDE, MASKED>(ctx, a, grid, unroll, kind); break;
        case MA_OP_REMAINDER: launch_vec_kind<T, MA_OP_REMAINDER, MASKED>(ctx, a, grid, unroll, kind); break;
        default: launch_vec_kind<T, MA_OP_FLOORDIV, MASKED>(ctx, a, grid, unroll, kind); break;
    }
}

// Enqueues the kernels of one call (or of one tile of a host-resident call) on ctx->stream. Every pointer of `a` is
// device-reachable; a.words / a.bit_off / a.out_words describe the validity of row 0 onwards. copy_mask: also write the
// output validity of a masked op whose validity is not data-dependent (the tiled path does that once for all tiles).
template <typename T>
ma_status enqueue_binary(ma_ctx* ctx, BinArgs<T> a, bool fma, bool masked, bool copy_mask) {
    constexpr bool kInt = std::is_integral<T>::value;
    const size_t n = a.n;
    if (masked) a.last_word = (a.bit_off + n - 1) >> 6;
    if (masked && a.words2) a.last_word2 = (a.bit_off2 + n - 1) >> 6;
    const bool int_div = kInt && !fma &&
                         (a.op == MA_OP_DIVIDE || a.op == MA_OP_REMAINDER || a.op == MA_OP_FLOORDIV);
    const bool ballot = masked && int_div;  // output validity depends on the data: the kernels write out_words themselves
    // The vec kernel peels `head` rows so that its STORES are 16-byte aligned; the inputs may sit on any element-aligned
    // phase (views sliced at different offsets, routing/arithmetic.rs:273-285): load16u.
    uintptr_t phase = (uintptr_t)a.out & 15;
    const bool same_phase = true;
    constexpr int R = 16 / (int)sizeof(T);
    // Launch shape (profiles/r01_sweep_grid.json, r01_sweep_binary.txt, r01_ubench_stream.txt). With a store stream in
    // the mix — unlike the read-only sums — MORE resident workgroups help (the memory system batches writes better
    // with more of them queued), 8 accesses per operand in flight, and the 16-KiB tiles of UNROLL = 4 resonate with
    // power-of-two grids (512/1024/2048: -5..-10 %). UNROLL = 8 with 6 workgroups per CU sits on the plateau
    // (a(+)b 3.92 ms, a(+)scalar 2.59 ms, fma 5.36 ms at 10^9 f64 rows; device-to-device spread is ~10 %).
    int unroll = 8;
    int bpc = ctx->blocks_per_cu > 0 ? ctx->blocks_per_cu : 6;
    switch ((tuning_variant(ctx) >> 1) & 7) {
        case 2: unroll = 4; break;
        case 3: unroll = 8; break;
        default: break;
    }
    if (fma) {
        unroll = 8;
        switch ((tuning_variant(ctx) >> 1) & 7) {
            case 1: unroll = 2; break;
            case 2: unroll = 4; break;
            default: break;
        }
    } else {
        unroll = clamp_unroll<T>(unroll, masked);
    }
    const size_t tile_rows = (size_t)64 * R * unroll * kWaves;
    size_t head = 0, n_tiles = 0;
    // Data-dependent validity: the vec kernel writes whole validity words, which needs row 0 on a 16-byte boundary
    // (head == 0); otherwise the row kernel ballots over everything.
    if (same_phase && !(ballot && phase != 0)) {
        head = phase ? (16 - phase) / sizeof(T) : 0;
        if (head > n) head = n;
        n_tiles = (n - head) / tile_rows;
    }
    a.head = head;
    a.n_tiles = n_tiles;
    a.ballot_mask = ballot ? 1 : 0;
    const size_t tail_start = head + n_tiles * tile_rows;  // ballot mode: head == 0 and this is a multiple of 64

    if (masked && !ballot && copy_mask) {
        if (a.words2) MA_TRY(launch_mask_combine(ctx, a.words, a.bit_off, a.words2, a.bit_off2, n, a.combine_and != 0, a.out_words));
        else MA_TRY(launch_mask_copy(ctx, a.words, a.bit_off, n, a.out_words));
    }
    if (n_tiles) {
        int grid = grid_for(ctx, n_tiles, bpc);
        if (fma) {
            if constexpr (!kInt) {
#define MA_FMA_LAUNCH(M, U) hipLaunchKernelGGL((fma_vec_kernel<T, M, U>), dim3(grid), dim3(kBlock), 0, ctx->stream, a)
                if constexpr (MA_TUNING) {
                    if (masked) {
                        if (unroll == 2) MA_FMA_LAUNCH(true, 2); else if (unroll == 4) MA_FMA_LAUNCH(true, 4);
                    } else {
                        if (unroll == 2) MA_FMA_LAUNCH(false, 2); else if (unroll == 4) MA_FMA_LAUNCH(false, 4);
                    }
                }
                if (unroll == 8) {
                    if (masked) MA_FMA_LAUNCH(true, 8); else MA_FMA_LAUNCH(false, 8);
                }
#undef MA_FMA_LAUNCH
            }
        } else if (masked) {
            launch_vec_op<T, true>(ctx, a, grid, unroll, a.kind, a.op);
        } else {
            launch_vec_op<T, false>(ctx, a, grid, unroll, a.kind, a.op);
        }
        MA_HIP(hipGetLastError());
    }
    if (head > 0 || tail_start < n) {
        size_t words_touched = (head + 63) / 64 + (n - tail_start + 63) / 64 + 1;
        int grid = grid_for(ctx, (words_touched + kWaves - 1) / kWaves, 8);
        if (fma) {
            if constexpr (!kInt) {
                if (masked) hipLaunchKernelGGL((binary_row_kernel<T, true, true>), dim3(grid), dim3(kBlock), 0, ctx->stream, a, tail_start);
                else hipLaunchKernelGGL((binary_row_kernel<T, false, true>), dim3(grid), dim3(kBlock), 0, ctx->stream, a, tail_start);
            }
        } else {
            if (masked) hipLaunchKernelGGL((binary_row_kernel<T, true, false>), dim3(grid), dim3(kBlock), 0, ctx->stream, a, tail_start);
            else hipLaunchKernelGGL((binary_row_kernel<T, false, false>), dim3(grid), dim3(kBlock), 0, ctx->stream, a, tail_start);
        }
        MA_HIP(hipGetLastError());
    }

    return MA_OK;
}

// After the kernels of a dense integer Div/Rem/FloorDiv have run: turns the device latch into the reference's panic.
static ma_status check_divide_latch(ma_ctx* ctx, int op) {
    uint32_t flags = 0;
    MA_HIP(hipMemcpyAsync(&flags, ctx->dev_flags, sizeof(flags), hipMemcpyDeviceToHost, ctx->stream));
    MA_HIP(hipStreamSynchronize(ctx->stream));
    if (flags & 1u) {
        MA_HIP(hipMemsetAsync(ctx->dev_flags, 0, sizeof(flags), ctx->stream));
        MA_HIP(hipStreamSynchronize(ctx->stream));
        // The reference panics here: "Division by zero" / "Remainder by zero" / "Floor division by zero"
        // (src/kernels/arithmetic/std.rs:53-77); asserted by src/kernels/arithmetic/mod.rs:161-177.
        set_error("%s by zero in a dense integer kernel",
                  op == MA_OP_DIVIDE ? "Division" : op == MA_OP_REMAINDER ? "Remainder" : "Floor division");
        return MA_ERR_DIVIDE_BY_ZERO;
    }
    return MA_OK;
}

// One tile of a host-resident call (run_tiled, ma_pipeline.hip): operand order lhs, rhs, acc, out.
template <typename T>
struct TileCall {
    ma_ctx* ctx;
    BinArgs<T> base;  // scalar / op / kind / flags and the whole call's validity (words, bit_off, out_words)
    bool fma, masked;
    static ma_status run(void* user, size_t row0, size_t rows, void* const* ptrs) {
        const TileCall& t = *(const TileCall*)user;
        BinArgs<T> a = t.base;
        a.lhs = (const T*)ptrs[0];
        a.rhs = (const T*)ptrs[1];
        a.acc = (const T*)ptrs[2];
        a.out = (T*)ptrs[3];
        a.n = rows;
        if (t.masked) {  // row0 is a multiple of 64: the tile's output validity starts on a word
            const size_t bit = t.base.bit_off + row0;
            a.words = t.base.words + (bit >> 6);
            a.bit_off = bit & 63;
            a.out_words = t.base.out_words + (row0 >> 6);
            if (t.base.words2) {
                const size_t bit2 = t.base.bit_off2 + row0;
                a.words2 = t.base.words2 + (bit2 >> 6);
                a.bit_off2 = bit2 & 63;
            }
        }
        return enqueue_binary<T>(t.ctx, a, t.fma, t.masked, false);
    }
};

template <typename T>
ma_status binary_impl(ma_ctx* ctx, const BinaryCall<T>& c) {
    constexpr bool kInt = std::is_integral<T>::value;
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    MA_REQUIRE(!(c.fma && kInt), MA_ERR_UNSUPPORTED, "FMA is defined for f32/f64 only");
    MA_REQUIRE(c.fma || (c.op >= MA_OP_ADD && c.op <= MA_OP_FLOORDIV), MA_ERR_INVALID_ARGUMENT,
               "unknown ArithmeticOperator code %d", c.op);
    // confirm_equal_len — src/utils.rs:163-171 via dispatch.rs:81,228-229
    size_t n = c.kind == kSA ? c.rhs_len : c.lhs_len;
    if (c.kind == kAA && c.lhs_len != c.rhs_len) {
        set_error("apply numeric: length mismatch (lhs: %zu, rhs: %zu)", c.lhs_len, c.rhs_len);
        return MA_ERR_LENGTH_MISMATCH;
    }
    if (c.fma && c.acc_len != n) {
        set_error("acc length mismatch (lhs: %zu, rhs: %zu)", n, c.acc_len);
        return MA_ERR_LENGTH_MISMATCH;
    }
    if (n == 0) return MA_OK;
    const bool masked = c.mask_bits != nullptr;
    MA_REQUIRE(c.out != nullptr, MA_ERR_INVALID_ARGUMENT, "out is NULL");
    MA_REQUIRE(c.kind == kSA || c.lhs != nullptr, MA_ERR_INVALID_ARGUMENT, "lhs is NULL");
    MA_REQUIRE(c.kind == kAS || c.rhs != nullptr, MA_ERR_INVALID_ARGUMENT, "rhs is NULL");
    MA_REQUIRE(!c.fma || c.acc != nullptr, MA_ERR_INVALID_ARGUMENT, "acc is NULL");
    MA_REQUIRE(!masked || c.out_mask_bits != nullptr, MA_ERR_INVALID_ARGUMENT,
               "a masked call needs an output bitmap (reference: Bitmask::new_set_all(len, true), dispatch.rs:92)");
    auto aligned_elem = [](const void* p) { return ((uintptr_t)p % sizeof(T)) == 0; };
    MA_REQUIRE(aligned_elem(c.lhs) && aligned_elem(c.rhs) && aligned_elem(c.acc) && aligned_elem(c.out),
               MA_ERR_INVALID_ARGUMENT, "a data pointer is not aligned to its element size");

    MA_ENTER(ctx);
    MA_HIP(hipSetDevice(ctx->device));
    const bool int_div = kInt && !c.fma &&
                         (c.op == MA_OP_DIVIDE || c.op == MA_OP_REMAINDER || c.op == MA_OP_FLOORDIV);
    const bool may_latch = int_div && !masked;
    CallScope scope(ctx);
    BinArgs<T> a{};
    a.scalar = c.scalar;
    a.n = n;
    a.op = c.op;
    a.kind = c.kind;
    a.flags = ctx->dev_flags;

    // Host-resident columns (a Rust &[T] that is not ma_alloc64_pinned memory) cross PCIe in tiles, both directions
    // at once (ma_pipeline.hip); small calls and device-reachable operands take the direct path below.
    const size_t tile_rows = (ctx->staging_tile_bytes / sizeof(T)) & ~(size_t)32767;  // whole vec tiles, whole words
    if (tile_rows && n >= 2 * tile_rows && !ctx->capturing) {
        PipeOperand ops[4] = {{c.kind != kSA ? c.lhs : nullptr, nullptr, sizeof(T), false},
                              {c.kind != kAS ? c.rhs : nullptr, nullptr, sizeof(T), false},
                              {c.fma ? c.acc : nullptr, nullptr, sizeof(T), false},
                              {nullptr, c.out, sizeof(T), false}};
        bool any = false;
        for (auto& o : ops) {
            const void* q = o.out ? o.out : o.in;
            o.staged = q != nullptr && crosses_in_tiles(ctx, pointer_kind(q));
            any = any || o.staged;
        }
        if (any) {
            if (masked) {
                MA_TRY(scope.in_mask(c.mask_bits, c.mask_bit_offset, n, &a.words, &a.bit_off));
                if (c.mask2_bits) MA_TRY(scope.in_mask(c.mask2_bits, c.mask2_bit_offset, n, &a.words2, &a.bit_off2));
                a.combine_and = c.combine_and ? 1 : 0;
                MA_TRY(scope.out_mask(c.out_mask_bits, n, &a.out_words));
                if (!int_div) {
                    if (a.words2) MA_TRY(launch_mask_combine(ctx, a.words, a.bit_off, a.words2, a.bit_off2, n, c.combine_and, a.out_words));
                    else MA_TRY(launch_mask_copy(ctx, a.words, a.bit_off, n, a.out_words));
                }
            }
            TileCall<T> call{ctx, a, c.fma, masked};
            MA_TRY(run_tiled(ctx, n, tile_rows, ops, 4, &TileCall<T>::run, &call));
            MA_TRY(scope.finish());  // validity temporaries back to the host
            return may_latch ? check_divide_latch(ctx, c.op) : MA_OK;
        }
    }

    const void* p = nullptr;
    if (c.kind != kSA) {
        MA_TRY(scope.in(c.lhs, n * sizeof(T), &p));
        a.lhs = (const T*)p;
    }
    if (c.kind != kAS) {
        MA_TRY(scope.in(c.rhs, n * sizeof(T), &p));
        a.rhs = (const T*)p;
    }
    if (c.fma) {
        MA_TRY(scope.in(c.acc, n * sizeof(T), &p));
        a.acc = (const T*)p;
    }
    void* po = nullptr;
    MA_TRY(scope.out(c.out, n * sizeof(T), &po));
    a.out = (T*)po;
    if (masked) {
        MA_TRY(scope.in_mask(c.mask_bits, c.mask_bit_offset, n, &a.words, &a.bit_off));
        if (c.mask2_bits) MA_TRY(scope.in_mask(c.mask2_bits, c.mask2_bit_offset, n, &a.words2, &a.bit_off2));
        a.combine_and = c.combine_and ? 1 : 0;
        MA_TRY(scope.out_mask(c.out_mask_bits, n, &a.out_words));
    }
    MA_TRY(enqueue_binary<T>(ctx, a, c.fma, masked, true));

    if (may_latch && (is_async(ctx) && !scope.staged())) ctx->pending_flags = true;
    MA_TRY(end_call(ctx, scope));
    if (may_latch && (!is_async(ctx) || scope.staged())) return check_divide_latch(ctx, c.op);
    return MA_OK;
}

}  // namespace ma

// ------------------------------------------------------------------------------------------------
// C ABI stamping
// ------------------------------------------------------------------------------------------------
#define MA_DEFINE_APPLY(FAMILY, TAG, T)                                                                               \
    extern "C" ma_status ma_apply_##FAMILY##_##TAG(ma_ctx* ctx, const T* lhs, size_t lhs_len, const T* rhs,           \
                                                   size_t rhs_len, int32_t op, const uint8_t* mask_bits,              \
                                                   size_t mask_bit_offset, T* out, uint8_t* out_mask_bits) {          \
        ::ma::BinaryCall<T> c;                                                                                        \
        c.op = op; c.kind = ::ma::kAA; c.lhs = lhs; c.lhs_len = lhs_len; c.rhs = rhs; c.rhs_len = rhs_len;            \
        c.mask_bits = mask_bits; c.mask_bit_offset = mask_bit_offset; c.out = out; c.out_mask_bits = out_mask_bits;   \
        return ::ma::binary_impl<T>(ctx, c);                                                                          \
    }                                                                                                                 \
    extern "C" ma_status ma_apply_##FAMILY##_##TAG##_scalar_rhs(ma_ctx* ctx, const T* lhs, size_t lhs_len, T scalar,  \
                                                                int32_t op, const uint8_t* mask_bits,                 \
                                                                size_t mask_bit_offset, T* out,                       \
                                                                uint8_t* out_mask_bits) {                             \
        ::ma::BinaryCall<T> c;                                                                                        \
        c.op = op; c.kind = ::ma::kAS; c.lhs = lhs; c.lhs_len = lhs_len; c.scalar = scalar;                           \
        c.mask_bits = mask_bits; c.mask_bit_offset = mask_bit_offset; c.out = out; c.out_mask_bits = out_mask_bits;   \
        return ::ma::binary_impl<T>(ctx, c);                                                                          \
    }                                                                                                                 \
    extern "C" ma_status ma_apply_##FAMILY##_##TAG##_scalar_lhs(ma_ctx* ctx, T scalar, const T* rhs, size_t rhs_len,  \
                                                                int32_t op, const uint8_t* mask_bits,                 \
                                                                size_t mask_bit_offset, T* out,                       \
                                                                uint8_t* out_mask_bits) {                             \
        ::ma::BinaryCall<T> c;                                                                                        \
        c.op = op; c.kind = ::ma::kSA; c.rhs = rhs; c.rhs_len = rhs_len; c.scalar = scalar;                           \
        c.mask_bits = mask_bits; c.mask_bit_offset = mask_bit_offset; c.out = out; c.out_mask_bits = out_mask_bits;   \
        return ::ma::binary_impl<T>(ctx, c);                                                                          \
    }

// apply_datetime's shape (dispatch.rs:309-372): both operands carry their own validity, the kernel combines them per
// row in registers instead of materialising merge_bitmasks_to_new's result. Internal (ma_datetime.hip), not exported.
#define MA_DEFINE_APPLY_TWO_MASKS(TAG, T)                                                                             \
    namespace ma {                                                                                                    \
    ma_status apply_int_two_masks_##TAG(ma_ctx* ctx, const T* lhs, size_t lhs_len, const T* rhs, size_t rhs_len,      \
                                        int32_t op, const uint8_t* mask1, size_t off1, const uint8_t* mask2,          \
                                        size_t off2, bool combine_and, T* out, uint8_t* out_mask_bits) {             \
        BinaryCall<T> c;                                                                                              \
        c.op = op; c.kind = kAA; c.lhs = lhs; c.lhs_len = lhs_len; c.rhs = rhs; c.rhs_len = rhs_len;                  \
        c.mask_bits = mask1; c.mask_bit_offset = off1; c.mask2_bits = mask2; c.mask2_bit_offset = off2;               \
        c.combine_and = combine_and; c.out = out; c.out_mask_bits = out_mask_bits;                                    \
        return binary_impl<T>(ctx, c);                                                                                \
    }                                                                                                                 \
    }

#define MA_DEFINE_APPLY_FMA(TAG, T)                                                                                   \
    extern "C" ma_status ma_apply_fma_##TAG(ma_ctx* ctx, const T* lhs, size_t lhs_len, const T* rhs, size_t rhs_len,  \
                                            const T* acc, size_t acc_len, const uint8_t* mask_bits,                   \
                                            size_t mask_bit_offset, T* out, uint8_t* out_mask_bits) {                 \
        ::ma::BinaryCall<T> c;                                                                                        \
        c.fma = true; c.kind = ::ma::kAA; c.lhs = lhs; c.lhs_len = lhs_len; c.rhs = rhs; c.rhs_len = rhs_len;         \
        c.acc = acc; c.acc_len = acc_len;                                                                             \
        c.mask_bits = mask_bits; c.mask_bit_offset = mask_bit_offset; c.out = out; c.out_mask_bits = out_mask_bits;   \
        return ::ma::binary_impl<T>(ctx, c);                                                                          \
    }
