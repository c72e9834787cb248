// Device-side building blocks shared by the streaming kernels (gfx950 / wave64).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "ma_common.hpp"

namespace ma {

// ------------------------------------------------------------------------------------------------
// 16-byte vectors: one global_load_dwordx4 / global_store_dwordx4 per lane, 1 KiB per wave instruction.
// ------------------------------------------------------------------------------------------------
template <typename T>
struct Vec16;
template <>
struct Vec16<int64_t> {
    typedef long long type __attribute__((ext_vector_type(2)));
};
template <>
struct Vec16<uint64_t> {
    typedef unsigned long long type __attribute__((ext_vector_type(2)));
};
template <>
struct Vec16<double> {
    typedef double type __attribute__((ext_vector_type(2)));
};
template <>
struct Vec16<int32_t> {
    typedef int type __attribute__((ext_vector_type(4)));
};
template <>
struct Vec16<uint32_t> {
    typedef unsigned int type __attribute__((ext_vector_type(4)));
};
template <>
struct Vec16<float> {
    typedef float type __attribute__((ext_vector_type(4)));
};
template <>
struct Vec16<int16_t> {
    typedef short type __attribute__((ext_vector_type(8)));
};
template <>
struct Vec16<uint16_t> {
    typedef unsigned short type __attribute__((ext_vector_type(8)));
};
template <>
struct Vec16<int8_t> {
    typedef signed char type __attribute__((ext_vector_type(16)));
};
template <>
struct Vec16<uint8_t> {
    typedef unsigned char type __attribute__((ext_vector_type(16)));
};

// Idle cycles between two consecutive load instructions of a wave. A wave that fires its UNROLL loads back to back
// hands the memory pipeline a burst and then nothing; a few cycles between them read 1-2 % faster on MI355X
// (tools/ubench_pace.hip, profiles/r01_ubench_pace.txt: 91.9 vs 90.2 % of 8 TB/s for 8 loads per wave, one wave per
// SIMD). Compile-time cycle count (s_nop n idles n + 1 cycles); the memory clobber keeps the loads in program order.
template <int CYCLES>
__device__ __forceinline__ void pace_loads() {
    static_assert(CYCLES >= 0 && CYCLES <= 32, "pace_loads: 0..32 cycles");
    if constexpr (CYCLES > 16) asm volatile("s_nop 15" ::: "memory");
    if constexpr (CYCLES % 16 != 0) asm volatile("s_nop %0" ::"n"(CYCLES % 16 - 1) : "memory");
    if constexpr (CYCLES == 16 || CYCLES == 32) asm volatile("s_nop 15" ::: "memory");
}

// Every pointer these helpers see is GLOBAL memory (device or pinned host) — never LDS or scratch. Saying so matters when
// the pointer was itself loaded from memory (a chunk descriptor table): the compiler then knows nothing about its address
// space and emits FLAT instructions, which count on lgkmcnt as well as vmcnt — every wait for a descriptor prefetch or an
// LDS access then also waits for the data stream. With the cast the batched kernels' streams are global_load / global_store
// like those of the kernels whose pointers are kernel arguments.
template <typename T>
__device__ __forceinline__ T __attribute__((address_space(1)))* as_global(T* p) {
    return (T __attribute__((address_space(1)))*)p;
}

// The 16 bytes move as four dwords whatever V's element type: a non-temporal load or store of a vector of 1-byte elements
// loses its `nt` bit when the compiler legalises <16 x i8> (the 1-byte sums read at 6.2 TB/s with plain loads against 6.9
// for the 2-byte types, whose vectors keep it).
typedef unsigned int MaU4 __attribute__((ext_vector_type(4), may_alias));
typedef unsigned int MaU2 __attribute__((ext_vector_type(2), may_alias));
template <size_t BYTES>
struct MaDwords;  // the dword vector a load / store of BYTES bytes moves as (16: global_*_dwordx4, 8: dwordx2)
template <>
struct MaDwords<16> {
    typedef MaU4 type;
};
template <>
struct MaDwords<8> {
    typedef MaU2 type;
};

template <typename V, bool NT>
__device__ __forceinline__ V load16(const V* p) {
    typedef typename MaDwords<sizeof(V)>::type D;
    typedef const D __attribute__((address_space(1)))* GP;
    if constexpr (NT) {
        return __builtin_bit_cast(V, __builtin_nontemporal_load((GP)p));
    } else {
        return __builtin_bit_cast(V, *(GP)p);
    }
}

// The same 16-byte load from an address that is only ELEMENT aligned (a view that starts mid-vector, an operand on a
// different 16-byte phase than the output). gfx950 under HSA runs with unaligned access mode on: one
// global_load_dwordx4 either way; a misaligned wave access touches one extra cache line per KiB.
template <typename V, bool NT>
__device__ __forceinline__ V load16u(const V* p) {
    typedef typename MaDwords<sizeof(V)>::type VU __attribute__((aligned(1)));
    typedef const VU __attribute__((address_space(1)))* GP;  // spelled out: a template would drop the typedef's alignment
    if constexpr (NT) {
        return __builtin_bit_cast(V, __builtin_nontemporal_load((GP)p));
    } else {
        return __builtin_bit_cast(V, *(GP)p);
    }
}

template <typename V, bool NT>
__device__ __forceinline__ void store16(V* p, V v) {
    typedef typename MaDwords<sizeof(V)>::type D;
    typedef D __attribute__((address_space(1)))* GP;
    if constexpr (NT) {
        __builtin_nontemporal_store(__builtin_bit_cast(D, v), (GP)p);
    } else {
        *(GP)p = __builtin_bit_cast(D, v);
    }
}

// Agent-scope relaxed load: bypasses this CU's L1 (global_load ... sc1).
__device__ __forceinline__ uint64_t load_agent(const uint64_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Agent-scope relaxed store: write-through to device scope (global_store ... sc1), dropped from this XCD's L2.
__device__ __forceinline__ void store_agent(uint64_t* p, uint64_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------------------------------------
// Validity words for one wave's contiguous run of rows.
//
// A run covers WPT <= 64 consecutive u64 words once it is shifted onto its first bit. Lane k (k <= WPT) loads
// word w0+k; the pair (k, k+1) is funnel-shifted by the run's sub-word bit offset, so lane k ends up with
// run-word k: bit j of it is the validity of run row 64*k + j. Works for ANY bit offset.
// ------------------------------------------------------------------------------------------------
template <int WPT>
__device__ __forceinline__ uint64_t load_run_words(const uint64_t* __restrict__ words, size_t bit0, size_t last_word,
                                                   unsigned lane) {
    static_assert(WPT <= 64, "a wave loads its run's validity words with one lane each");
    const size_t w0 = bit0 >> 6;
    const unsigned sh = (unsigned)(bit0 & 63);
    const auto gw = as_global(words);
    uint64_t mw = 0;
    if (lane <= (unsigned)WPT && w0 + lane <= last_word) mw = gw[w0 + lane];
    uint64_t nx = (uint64_t)__shfl_down((unsigned long long)mw, 1, 64);
    if constexpr (WPT == 64) {
        // all 64 lanes hold a run word; the last one needs word 64 for its funnel shift and fetches it itself
        if (lane == 63) nx = (sh && w0 + 64 <= last_word) ? gw[w0 + 64] : 0;
    }
    return sh ? ((mw >> sh) | (nx << (64 - sh))) : mw;
}

// In two steps, for kernels that fetch the words of the NEXT run while they work on this one: the raw word of this lane
// (a load and nothing that waits for it), and the funnel shift across lanes once the value is wanted.
template <int WPT>
__device__ __forceinline__ uint64_t load_run_raw(const uint64_t* __restrict__ words, size_t bit0, size_t last_word, unsigned lane) {
    static_assert(WPT < 64, "lane WPT holds the word the last run-word funnels from");
    const size_t w0 = bit0 >> 6;
    const auto gw = as_global(words);
    uint64_t mw = 0;
    if (lane <= (unsigned)WPT && w0 + lane <= last_word) mw = gw[w0 + lane];
    return mw;
}
__device__ __forceinline__ uint64_t finish_run_words(uint64_t mw, size_t bit0) {
    const unsigned sh = (unsigned)(bit0 & 63);
    const uint64_t nx = (uint64_t)__shfl_down((unsigned long long)mw, 1, 64);
    return sh ? ((mw >> sh) | (nx << (64 - sh))) : mw;
}

// The same for runs of up to 128 words (1-byte types at 8 loads per lane: 16 rows x 8 x 64 lanes = 128 validity
// words): lane k holds run-words k and 64 + k, fetched by two wave instructions. word(i) hands every lane run-word i
// when i is wave-uniform, or each lane its own word when it is not.
template <int WPT>
struct RunWords {
    static_assert(WPT >= 1 && WPT <= 128, "a wave covers at most 128 validity words per run");
    static constexpr int N = (WPT + 63) / 64;
    uint64_t w[N];

    __device__ __forceinline__ void load(const uint64_t* __restrict__ words, size_t bit0, size_t last_word, unsigned lane) {
        const size_t w0 = bit0 >> 6;
        const unsigned sh = (unsigned)(bit0 & 63);
        uint64_t raw[N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const size_t idx = w0 + (size_t)(64 * j) + lane;
            raw[j] = ((unsigned)(64 * j) + lane <= (unsigned)WPT && idx <= last_word) ? words[idx] : 0;
        }
#pragma unroll
        for (int j = 0; j < N; ++j) {
            uint64_t nx = (uint64_t)__shfl_down((unsigned long long)raw[j], 1, 64);
            if (j + 1 < N) {
                const uint64_t first_of_next = (uint64_t)__shfl((unsigned long long)raw[j + 1 < N ? j + 1 : j], 0, 64);
                if (lane == 63) nx = first_of_next;
            } else if (WPT == 64 * N) {
                // every lane of the last register holds a run word; the very last one fetches its funnel partner itself
                if (lane == 63) nx = (sh && w0 + (size_t)(64 * N) <= last_word) ? words[w0 + (size_t)(64 * N)] : 0;
            }
            w[j] = sh ? ((raw[j] >> sh) | (nx << (64 - sh))) : raw[j];
        }
    }
    // this lane's R validity bits for load step u (u is a compile-time constant after unrolling)
    template <int R>
    __device__ __forceinline__ unsigned bits(int u, unsigned lane) const {
        constexpr int LPW = 64 / R;
        const int idx = u * R + (int)(lane / LPW);
        const uint64_t word = (uint64_t)__shfl((unsigned long long)w[(u * R) >> 6], idx & 63, 64);
        return (unsigned)(word >> ((lane % LPW) * R)) & ((1u << R) - 1u);
    }
    __device__ __forceinline__ void combine(const RunWords& o, bool is_and) {
#pragma unroll
        for (int j = 0; j < N; ++j) w[j] = is_and ? (w[j] & o.w[j]) : (w[j] | o.w[j]);
    }
};

// The R validity bits of this lane's rows for load step `u` of the run (R rows per lane per 16-byte load).
template <int R>
__device__ __forceinline__ unsigned lane_bits(uint64_t run_word_of_lane, int u, unsigned lane) {
    constexpr int LPW = 64 / R;  // lanes per validity word
    uint64_t w = (uint64_t)__shfl((unsigned long long)run_word_of_lane, u * R + (int)(lane / LPW), 64);
    return (unsigned)(w >> ((lane % LPW) * R)) & ((1u << R) - 1u);
}

// The inverse of lane_bits: lane l holds R result bits for rows l*R .. l*R+R-1 of a 64*R-row step. Butterfly OR over
// the 64/R lanes that share an output word; afterwards EVERY lane of such a group holds the finished word, and the
// group's first lane (lane % (64/R) == 0) stores it as word lane / (64/R) of the step.
template <int R>
__device__ __forceinline__ uint64_t pack_lane_bits(unsigned bits, unsigned lane) {
    constexpr int LPW = 64 / R;
    uint64_t v = (uint64_t)bits << ((lane % LPW) * R);
#pragma unroll
    for (int s = 1; s < LPW; s <<= 1) v |= (uint64_t)__shfl_xor((unsigned long long)v, s, 64);
    return v;
}

// Zeroes the null slots of a 16-byte vector of 1- or 2-byte elements: the lane's R validity bits are spread into byte /
// halfword masks with one multiply per 32-bit word (bit i of a nibble -> byte i: n * 0x00204081 puts b0..b3 at bits
// 0, 8, 16, 24 with no carries; x 0xFF fills the bytes) instead of a test + select per element — 16 VALU operations per
// vector where the per-element form needs 48.
typedef unsigned int U32x4 __attribute__((ext_vector_type(4)));
template <typename V, int ELEM_BYTES>
__device__ __forceinline__ V zero_null_slots(V v, unsigned bits) {
    static_assert(ELEM_BYTES == 1 || ELEM_BYTES == 2, "wider elements select per element");
    U32x4 w = __builtin_bit_cast(U32x4, v);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned m;
        if constexpr (ELEM_BYTES == 1) m = ((((bits >> (4 * i)) & 0xFu) * 0x00204081u) & 0x01010101u) * 0xFFu;
        else m = ((((bits >> (2 * i)) & 0x3u) * 0x8001u) & 0x00010001u) * 0xFFFFu;
        w[i] &= m;
    }
    return __builtin_bit_cast(V, w);
}

__device__ __forceinline__ unsigned row_bit(const uint64_t* words, size_t bit) {
    return (unsigned)(as_global((const uint8_t*)words)[bit >> 3] >> (bit & 7)) & 1u;
}

// Narrow integer columns (i8 / u8 / i16 / u16 — the reference's extended_numeric_types, src/enums/collections/
// numeric_array.rs:81-99): 16 or 8 rows per 16-byte load. Widening every element to the u64 accumulator would cost ~48
// VALU instructions per load; instead the valid elements of one load are summed inside 32-bit registers — bytes four at a
// time with v_sad_u8 (sum of absolute differences against 0), halves with two masked adds — and only the per-load total
// (< 2^20) goes to the 64-bit accumulator. Signed types are biased into unsigned ones (x ^ 0x80.. = x + 128 | 32768 as an
// unsigned value) and the bias of the VALID elements is taken off again: exact, wrapping like every integer sum here.
// bits: validity of the load's R rows, bit r = row r (all ones for a dense scan).
template <typename T, typename V16>
__device__ __forceinline__ int64_t narrow_vec_sum(const V16& v, unsigned bits) {
    static_assert(sizeof(V16) == 16, "one 16-byte load");
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const u4 d = __builtin_bit_cast(u4, v);
    constexpr bool kSigned = std::is_signed<T>::value;
    unsigned s = 0;
    if constexpr (sizeof(T) == 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned x = kSigned ? (d[k] ^ 0x80808080u) : d[k];
            const unsigned b = (bits >> (4 * k)) & 15u;
            // 4 validity bits -> 4 bytes of 0 / 1: bit i lands on bit 8 i (no two partial products share a position); the
            // valid bytes are then summed by ONE dot product with those bytes (v_dot4_u32_u8) — round 4; a second multiply
            // to widen the bits into byte masks, an AND and v_sad_u8 before
            const unsigned m01 = (b * 0x00204081u) & 0x01010101u;
            s = __builtin_amdgcn_udot4(x, m01, s, false);
        }
        const int n_valid = __popc(bits & 0xFFFFu);
        return kSigned ? (int64_t)s - 128 * (int64_t)n_valid : (int64_t)s;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned x = kSigned ? (d[k] ^ 0x80008000u) : d[k];
            const unsigned b = (bits >> (2 * k)) & 3u;
            const unsigned m = (b & 1u) * 0xFFFFu + (b >> 1) * 0xFFFF0000u;
            x &= m;
            s += (x & 0xFFFFu) + (x >> 16);
        }
        const int n_valid = __popc(bits & 0xFFu);
        return kSigned ? (int64_t)s - 32768 * (int64_t)n_valid : (int64_t)s;
    }
}

}  // namespace ma
