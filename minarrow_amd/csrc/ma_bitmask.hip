// Bitmask kernels for gfx950 — src/kernels/bitmask/{mod,std,simd,dispatch}.rs.
// A bitmap window is (bits, bit offset, bit length) = BitmaskVT (src/aliases.rs:172). One u64 word = 64 rows
// = one wave64, so these are plain word-wise kernels: one thread per output word, coalesced 8-byte accesses.
#include <atomic>
#include <chrono>

#include "ma_device.hpp"

namespace ma {

// Funnel-shifted read of the window word `j` (bits [bit_off + 64 j, +64)), zero beyond `last_word`.
__device__ __forceinline__ uint64_t window_word(const uint64_t* __restrict__ words, size_t bit_off, size_t last_word,
                                                size_t j) {
    const size_t b = bit_off + (j << 6);
    const size_t w = b >> 6;
    const unsigned sh = (unsigned)(b & 63);
    uint64_t lo = w <= last_word ? words[w] : 0;
    if (sh == 0) return lo;
    uint64_t hi = (w + 1) <= last_word ? words[w + 1] : 0;
    return (lo >> sh) | (hi << (64 - sh));
}

__global__ __launch_bounds__(kBlock) void mask_copy_kernel(const uint64_t* __restrict__ words, size_t bit_off,
                                                           size_t last_word, size_t n,
                                                           uint64_t* __restrict__ out_words) {
    const size_t n_words = (n + 63) >> 6;
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t j = (size_t)blockIdx.x * kBlock + threadIdx.x; j < n_words; j += stride) {
        uint64_t w = window_word(words, bit_off, last_word, j);
        if (j == n_words - 1 && (n & 63)) w &= (((uint64_t)1) << (n & 63)) - 1;  // bits >= n are zero
        out_words[j] = w;
    }
}

ma_status launch_mask_copy(ma_ctx* ctx, const uint64_t* words, size_t bit_off, size_t n, uint64_t* out_words) {
    if (n == 0) return MA_OK;
    const size_t n_words = (n + 63) >> 6;
    const size_t last_word = (bit_off + n - 1) >> 6;
    int grid = grid_for(ctx, (n_words + kBlock - 1) / kBlock, 8);
    hipLaunchKernelGGL(mask_copy_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, words, bit_off, last_word, n,
                       out_words);
    MA_HIP(hipGetLastError());
    return MA_OK;
}

}  // namespace ma

// ================================================================================================
// Word-wise kernels
// ================================================================================================
namespace ma {

enum : int { kBitAnd = 0, kBitOr = 1, kBitXor = 2, kBitNot = 3, kBitXnor = 4, kBitCopy = 5 };

struct BitArgs {
    const uint64_t* lw;  // lhs words (8-byte aligned base)
    size_t lo;           // bit offset of the lhs window
    size_t l_last;       // last lhs word index holding a window bit
    const uint64_t* rw;  // rhs words or nullptr
    size_t ro;
    size_t r_last;
    size_t n;            // window length in bits
    uint64_t* out;       // output words, window re-based to bit 0
    int op;
};

__device__ __forceinline__ uint64_t bit_op(int op, uint64_t x, uint64_t y) {
    switch (op) {
        case kBitAnd: return x & y;
        case kBitOr: return x | y;
        case kBitXor: return x ^ y;
        case kBitNot: return ~x;
        case kBitXnor: return ~(x ^ y);
        default: return x;
    }
}

typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
typedef u64x2 u64x2_u __attribute__((aligned(1)));  // read at a byte-aligned address (window starting mid-word)
constexpr int kVecUnroll = 8;  // 16-byte accesses per operand a lane keeps in flight in the word-pair paths
// Tile -> lane mapping of the word-pair paths, the one the elementwise kernels use: a wave owns kVecUnroll KiB of
// consecutive pairs, lane l takes pair l of each KiB.
__device__ __forceinline__ size_t pair_index(size_t tile, int u) {
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    return (tile * kWaves + wave) * ((size_t)kVecUnroll * 64) + (size_t)u * 64 + lane;
}

// Word pair of a window that starts at ANY bit: the pair is read with a byte-aligned 16-byte load at the window's first
// byte; the remaining 1..7 bits are funnelled in registers — x0 from (p.x, p.y), x1 from (p.y, the next pair's first
// byte), which is the neighbouring lane's p.x (lane 63 of a wave instruction reads that one byte itself).
__device__ __forceinline__ u64x2 funnel_pair(u64x2 p, unsigned sub, uint64_t next_of_lane63, unsigned lane) {
    if (sub == 0) return p;  // wave-uniform
    uint64_t nx = (uint64_t)__shfl_down((unsigned long long)p.x, 1, 64);
    if (lane == 63) nx = next_of_lane63;
    u64x2 r;
    r.x = (p.x >> sub) | (p.y << (64 - sub));
    r.y = (p.y >> sub) | (nx << (64 - sub));
    return r;
}

// `vec`: two words move per lane per access (global_load/store_dwordx4, non-temporal). The windows may start at any
// bit (round 2; byte-aligned starts only in round 1); the output must sit on a 16-byte boundary.
template <bool VEC>  // separate instantiations: the word-pair path's registers must not cost the scalar path its occupancy
__global__ __launch_bounds__(kBlock) void bit_words_kernel(BitArgs a) {
    const size_t n_words = (a.n + 63) >> 6;
    const size_t stride = (size_t)gridDim.x * kBlock;
    size_t first_scalar = 0;
    if constexpr (VEC) {
        const unsigned lane = threadIdx.x & 63;
        const unsigned sl = (unsigned)(a.lo & 7), sr = (unsigned)(a.ro & 7);
        const uint8_t* __restrict__ lb = (const uint8_t*)a.lw + (a.lo >> 3);
        const uint8_t* __restrict__ rb = a.rw ? (const uint8_t*)a.rw + (a.ro >> 3) : nullptr;
        const u64x2_u* __restrict__ lp = (const u64x2_u*)lb;
        const u64x2_u* __restrict__ rp = (const u64x2_u*)rb;
        u64x2* __restrict__ op = (u64x2*)a.out;
        // Pairs the vector body may touch: the last word (trailing-bit mask) always goes through the scalar path, and a
        // pair reads 17 bytes (16 + the funnel byte) which must stay inside the words that hold window bits.
        size_t n_pairs = (n_words - 1) >> 1;
        auto cap = [&](size_t off_bits, size_t last_word) {
            const size_t avail = (last_word + 1) * 8, b0 = off_bits >> 3;  // readable bytes; the window's first byte
            const size_t safe = avail >= b0 + 17 ? (avail - b0 - 17) / 16 + 1 : 0;
            if (safe < n_pairs) n_pairs = safe;
        };
        cap(a.lo, a.l_last);
        if (rb) cap(a.ro, a.r_last);
        // Tiles of kVecUnroll * kBlock pairs: every lane issues its kVecUnroll loads per operand before the first use.
        const size_t n_tiles = n_pairs / ((size_t)kVecUnroll * kBlock);
        for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
            u64x2 x[kVecUnroll], y[kVecUnroll];
#pragma unroll
            for (int u = 0; u < kVecUnroll; ++u) x[u] = __builtin_nontemporal_load(lp + pair_index(t, u));
#pragma unroll
            for (int u = 0; u < kVecUnroll; ++u) {
                y[u] = u64x2{0, 0};
                if (rp) y[u] = __builtin_nontemporal_load(rp + pair_index(t, u));
            }
            // Lane 63's funnel partner is lane 0's pair of the NEXT access (the wave's run is contiguous); only the last
            // access needs a byte from beyond the run, fetched along with the rest.
            uint64_t xt = 0, yt = 0;
            if (lane == 63) {
                const size_t after = 16 * (pair_index(t, kVecUnroll - 1) + 1);
                if (sl) xt = lb[after];
                if (rp && sr) yt = rb[after];
            }
#pragma unroll
            for (int u = 0; u < kVecUnroll; ++u) {
                const size_t j = pair_index(t, u);
                const uint64_t xn = u + 1 < kVecUnroll ? (uint64_t)__shfl((unsigned long long)x[u + 1 < kVecUnroll ? u + 1 : u].x, 0, 64) : xt;
                const uint64_t yn = u + 1 < kVecUnroll ? (uint64_t)__shfl((unsigned long long)y[u + 1 < kVecUnroll ? u + 1 : u].x, 0, 64) : yt;
                const u64x2 xx = funnel_pair(x[u], sl, xn, lane);
                u64x2 yy = y[u];
                if (rp) yy = funnel_pair(y[u], sr, yn, lane);
                u64x2 r;
                r.x = bit_op(a.op, xx.x, yy.x);
                r.y = bit_op(a.op, xx.y, yy.y);
                __builtin_nontemporal_store(r, op + j);
            }
        }
        // whole waves only: funnel_pair shuffles across the lanes of a wave instruction
        const size_t done = n_tiles * ((size_t)kVecUnroll * kBlock);
        const size_t n_vec = done + ((n_pairs - done) / 64) * 64;
        for (size_t j = done + (size_t)blockIdx.x * kBlock + threadIdx.x; j < n_vec; j += stride) {
            uint64_t xt = 0, yt = 0;
            if (lane == 63) {
                if (sl) xt = lb[16 * (j + 1)];
                if (rp && sr) yt = rb[16 * (j + 1)];
            }
            u64x2 x = funnel_pair(__builtin_nontemporal_load(lp + j), sl, xt, lane);
            u64x2 y = {0, 0};
            if (rp) y = funnel_pair(__builtin_nontemporal_load(rp + j), sr, yt, lane);
            u64x2 r;
            r.x = bit_op(a.op, x.x, y.x);
            r.y = bit_op(a.op, x.y, y.y);
            __builtin_nontemporal_store(r, op + j);
        }
        first_scalar = n_vec << 1;
    }
    for (size_t j = first_scalar + (size_t)blockIdx.x * kBlock + threadIdx.x; j < n_words; j += stride) {
        uint64_t x = window_word(a.lw, a.lo, a.l_last, j);
        uint64_t y = a.rw ? window_word(a.rw, a.ro, a.r_last, j) : 0;
        uint64_t r = bit_op(a.op, x, y);
        // clear_trailing_bits / mask_trailing_bits — bitmask/mod.rs:141-150, structs/bitmask.rs:83-90
        if (j == n_words - 1 && (a.n & 63)) r &= (((uint64_t)1) << (a.n & 63)) - 1;
        a.out[j] = r;
    }
}

// Scalar facts about one or two windows, four words per workgroup:
//   [0] = popcount(x)                [1] = any bit set in x
//   [2] = any bit clear in x         [3] = any bit where x != y
// Epilogue: the workgroup publishes its four words as one partial and takes a ticket (below); the workgroup whose arrival is
// the last folds all partials, writes the totals to the PINNED host slot, re-arms the ticket and stamps the completion word
// the host polls — no memset before, no copy after, no hipStreamSynchronize wake-up on the small-bitmap path (null counts).
// the context's arrival counters, laid out as ma_reduce.hip uses them (kernels of one context are stream-ordered)
constexpr unsigned kScanTicketShards = 8, kScanShardWord0 = 64, kScanShardStride = 16, kScanShardFrom = 96;
struct ScanOut {
    Partial* partials;          // device: one 32-byte partial per workgroup (the context's)
    unsigned int* ticket;       // device: the context's ticket word (+ shard counters), zero between launches
    unsigned long long* host;   // pinned: where the four totals go
    uint64_t* done_word;        // pinned: stamped with done_seq after the totals
    uint64_t done_seq;
};
template <bool VEC>
__global__ __launch_bounds__(kBlock) void bit_scan_kernel(BitArgs a, ScanOut so) {
    const size_t n_words = (a.n + 63) >> 6;
    const size_t stride = (size_t)gridDim.x * kBlock;
    unsigned long long pop = 0, any_set = 0, any_clear = 0, any_diff = 0;
    size_t first_scalar = 0;
    if constexpr (VEC) {
        const size_t n_pairs = (n_words - 1) >> 1;
        // windows start on BYTES here (vec_ok): a byte-shifted pointer is the window, no funnel shift needed
        const u64x2_u* __restrict__ lp = (const u64x2_u*)((const uint8_t*)a.lw + (a.lo >> 3));
        const u64x2_u* __restrict__ rp = a.rw ? (const u64x2_u*)((const uint8_t*)a.rw + (a.ro >> 3)) : nullptr;
        const size_t n_tiles = n_pairs / ((size_t)kVecUnroll * kBlock);
        for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
            u64x2 x[kVecUnroll], y[kVecUnroll];
#pragma unroll
            for (int u = 0; u < kVecUnroll; ++u) x[u] = __builtin_nontemporal_load(lp + pair_index(t, u));
            if (rp) {
#pragma unroll
                for (int u = 0; u < kVecUnroll; ++u) y[u] = __builtin_nontemporal_load(rp + pair_index(t, u));
            }
#pragma unroll
            for (int u = 0; u < kVecUnroll; ++u) {
                pop += (unsigned long long)(__popcll(x[u].x) + __popcll(x[u].y));
                any_set |= x[u].x | x[u].y;
                any_clear |= ~(x[u].x & x[u].y);
                if (rp) any_diff |= (x[u].x ^ y[u].x) | (x[u].y ^ y[u].y);
            }
        }
        for (size_t j = n_tiles * ((size_t)kVecUnroll * kBlock) + (size_t)blockIdx.x * kBlock + threadIdx.x; j < n_pairs;
             j += stride) {
            u64x2 x = __builtin_nontemporal_load(lp + j);
            pop += (unsigned long long)(__popcll(x.x) + __popcll(x.y));
            any_set |= x.x | x.y;
            any_clear |= ~(x.x & x.y);
            if (rp) {
                u64x2 y = __builtin_nontemporal_load(rp + j);
                any_diff |= (x.x ^ y.x) | (x.y ^ y.y);
            }
        }
        first_scalar = n_pairs << 1;
    }
    for (size_t j = first_scalar + (size_t)blockIdx.x * kBlock + threadIdx.x; j < n_words; j += stride) {
        uint64_t live = ~(uint64_t)0;
        if (j == n_words - 1 && (a.n & 63)) live = (((uint64_t)1) << (a.n & 63)) - 1;
        uint64_t x = window_word(a.lw, a.lo, a.l_last, j) & live;
        pop += (unsigned long long)__popcll(x);
        any_set |= x;
        any_clear |= (~x) & live;
        if (a.rw) any_diff |= (x ^ window_word(a.rw, a.ro, a.r_last, j)) & live;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        pop += __shfl_down(pop, off, 64);
        any_set |= __shfl_down(any_set, off, 64);
        any_clear |= __shfl_down(any_clear, off, 64);
        any_diff |= __shfl_down(any_diff, off, 64);
    }
    // ---- the sum kernels' hand-off (ma_reduce.hip): one 32-byte partial per workgroup, published with write-through stores
    // and drained, arrival on sharded tickets, the last workgroup folds every partial. Rounds 1-3 had every workgroup add its
    // four words to four global words with atomics: 512 workgroups x 4 atomics on one cache line serialise at ~21 ns each — a
    // 40-us tail on a 0.6-ms scan (popcount of a 4-GiB bitmap 6.78 -> 7.2 TB/s).
    __shared__ unsigned long long part[kWaves][4];
    __shared__ int is_last;
    const unsigned tid = threadIdx.x, G = gridDim.x, b = blockIdx.x;
    if ((tid & 63) == 0) {
        unsigned long long* q = part[tid >> 6];
        q[0] = pop;
        q[1] = any_set;
        q[2] = any_clear;
        q[3] = any_diff;
    }
    __syncthreads();
    if (tid == 0) {
        pop = any_set = any_clear = any_diff = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            pop += part[w][0];
            any_set |= part[w][1];
            any_clear |= part[w][2];
            any_diff |= part[w][3];
        }
        uint64_t* q = (uint64_t*)&so.partials[b];
        store_agent(q, (uint64_t)pop);
        store_agent(q + 1, (uint64_t)any_set);
        store_agent(q + 2, (uint64_t)any_clear);
        store_agent(q + 3, (uint64_t)any_diff);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // performed, not merely issued, before this workgroup arrives
        int last;
        if (G <= kScanShardFrom) {
            last = __hip_atomic_fetch_add(so.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == G - 1;
        } else {  // workgroups b and b + 8 share an XCD: a shard's arrivals stay on one L2; its last arrival goes to the top
            const unsigned sh = b & (kScanTicketShards - 1);
            const unsigned members = (G - sh + kScanTicketShards - 1) / kScanTicketShards;
            unsigned int* shard = so.ticket + kScanShardWord0 + sh * kScanShardStride;
            last = 0;
            if (__hip_atomic_fetch_add(shard, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1) {
                __hip_atomic_store(shard, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = __hip_atomic_fetch_add(so.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kScanTicketShards - 1;
            }
        }
        is_last = last;
    }
    __syncthreads();
    if (!is_last) return;
    pop = any_set = any_clear = any_diff = 0;
    for (unsigned i = tid; i < G; i += kBlock) {  // independent loads: one round trip
        const uint64_t* q = (const uint64_t*)&so.partials[i];
        pop += load_agent(q);
        any_set |= load_agent(q + 1);
        any_clear |= load_agent(q + 2);
        any_diff |= load_agent(q + 3);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        pop += __shfl_down(pop, off, 64);
        any_set |= __shfl_down(any_set, off, 64);
        any_clear |= __shfl_down(any_clear, off, 64);
        any_diff |= __shfl_down(any_diff, off, 64);
    }
    __syncthreads();  // part[][] is reused
    if ((tid & 63) == 0) {
        unsigned long long* q = part[tid >> 6];
        q[0] = pop;
        q[1] = any_set;
        q[2] = any_clear;
        q[3] = any_diff;
    }
    __syncthreads();
    if (tid != 0) return;
    pop = any_set = any_clear = any_diff = 0;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
        pop += part[w][0];
        any_set |= part[w][1];
        any_clear |= part[w][2];
        any_diff |= part[w][3];
    }
    so.host[0] = pop;
    so.host[1] = any_set ? 1 : 0;
    so.host[2] = any_clear ? 1 : 0;
    so.host[3] = any_diff ? 1 : 0;
    __hip_atomic_store(so.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch on this stream
    __hip_atomic_store(so.done_word, so.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// simd_eq_mask_u{8,16,32,64} — bit j = ((data[j] & field_mask) == target).
// Vector path (16-byte aligned data): a lane loads 16 bytes = R rows, compares them in registers and the R result
// bits of the 64/R lanes that share an output word are OR-ed together with a butterfly (pack_lane_bits); a wave
// step covers 1 KiB of data and stores R words. Rows past the last full wave tile (and unaligned data) take the
// one-ballot-per-64-rows path.
// The R result bits of one 16-byte load. 4- and 8-byte elements compare one by one; 1- and 2-byte elements are compared
// four / two at a time inside their 32-bit words: y = (word & mask) ^ target has a zero byte exactly where the row
// matches, ~(((y & 0x7F..) + 0x7F..) | y) & 0x80.. marks the zero bytes (exact per byte: the add cannot carry out of a
// byte; the ~ & is ONE v_bfi), and the marks — bytes of 0x80 or 0 — are gathered by a dot product with the bytes' bit
// weights (v_dot4_u32_u8: dword 0 with 1, 2, 4, 8, dword 1 with 16 .. 128, accumulating; dwords 2 and 3 the same into the
// upper byte), one shift by 7 at the end: 7 operations per 4 rows (round 5; 10 with the multiply-and-shift gather of round 4,
// 14 comparing row by row — counters: profiles/r05_subfamily_counters.md).
template <typename T, typename V16>
__device__ __forceinline__ unsigned eq_bits(const V16& x, T field_mask, T target) {
    constexpr int R = 16 / (int)sizeof(T);
    unsigned bits = 0;
    if constexpr (sizeof(T) == 1) {
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
        const u4 d = __builtin_bit_cast(u4, x);
        const unsigned m4 = (unsigned)(uint8_t)field_mask * 0x01010101u, t4 = (unsigned)(uint8_t)target * 0x01010101u;
        unsigned lo = 0, hi = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned y = (d[i] & m4) ^ t4;
            const unsigned t = ((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y;
            const unsigned z = ~t & 0x80808080u;  // 0x80 in every byte that matched
            const unsigned w = (i & 1) ? 0x80402010u : 0x08040201u;
            if (i < 2) lo = __builtin_amdgcn_udot4(z, w, lo, false);
            else hi = __builtin_amdgcn_udot4(z, w, hi, false);
        }
        bits = (lo >> 7) | ((hi >> 7) << 8);  // each sum is 0x80 x (the byte of its eight marks)
    } else if constexpr (sizeof(T) == 2) {
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
        const u4 d = __builtin_bit_cast(u4, x);
        const unsigned m2 = (unsigned)(uint16_t)field_mask * 0x00010001u, t2 = (unsigned)(uint16_t)target * 0x00010001u;
        unsigned acc = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned y = (d[i] & m2) ^ t2;
            const unsigned t = ((y & 0x7FFF7FFFu) + 0x7FFF7FFFu) | y;
            const unsigned z = ~t & 0x80008000u;  // 0x8000 in every halfword that matched
            // halves x (2^(2i), 2^(2i+1)): 0x8000 x the pair's two result bits, accumulated over the four dwords (v_dot2_u32_u16)
            acc = __builtin_amdgcn_udot2(__builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned short, z),
                                         __builtin_bit_cast(__attribute__((ext_vector_type(2))) unsigned short,
                                                            (unsigned)((1u << (2 * i)) | (2u << (2 * i + 16)))),
                                         acc, false);
        }
        bits = acc >> 15;
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) bits |= ((T)((T)x[r] & field_mask) == target ? 1u : 0u) << r;
    }
    return bits;
}

// The scan is a read stream with 1/64 .. 1/8 of its bytes written back: it takes the shape of the sums (ma_reduce.hip:
// eight 16-byte loads in flight per lane, one or two workgroups per CU) rather than that of the read + write kernels —
// 2^33 bytes of u64 at 4 loads x 8 workgroups per CU: 5.75 TB/s; see launch_eq_mask for the swept figures.
// The scan is a read stream with 1/64 .. 1/8 of its bytes written back (tools/sweep_eq_mask.py, 2^33 bytes of data,
// profiles/r03_sweep_eq_mask.jsonl). With the stores compiled out it reads at 6.8-7.5 TB/s like the sums; ANY form of the
// stores costs 12-20 % of that (the few writes interleave with the read stream in HBM), the round-2 form — eight-byte
// stores from the 2 .. 16 lanes that hold a finished word, one store instruction per 1-KiB step, 4 loads per lane, 8
// workgroups per CU — cost 25-35 %: u8 4.88, u16 5.17, u32 5.62, u64 5.75 TB/s; round 3: u8 5.65, u16 5.78, u32 5.88, u64 6.16.
// Now (u8 6.31, u16 6.62, u32 6.83, u64 6.95; profiles/r04_sweep_eq_mask.jsonl):
//   * eight 16-byte loads in flight per lane and one or two workgroups per CU, the shape of the sums;
//   * a wave collects the bits of its eight steps in its own 64 R bytes of LDS — lane l's R bits of step u ARE bits
//     [(64 u + l) R, +R) of the wave's output, so no butterfly over the lanes is needed — and writes them back as ONE
//     non-temporal store of 16 bytes per lane, contiguous;
//   * that store is issued late, behind a later tile's loads (see the loop);
//   * (round 4) the next tile's rows are requested before this tile's are compared, on ONE workgroup per CU.
template <typename T, int UNROLL>
__global__ __launch_bounds__(kBlock) void eq_mask_vec_kernel(const T* __restrict__ data, size_t n_tiles, T field_mask,
                                                             T target, uint64_t* __restrict__ out) {
    using V = typename Vec16<T>::type;
    // 1- and 2-byte elements stay four dwords from the load to the packed compare (a vector of 1-byte elements loses the
    // loads' non-temporal hint on its way through the optimiser: ma_device.hpp)
    typedef typename std::conditional<(sizeof(T) <= 2), MaU4, V>::type VL;
    typedef unsigned long long u2a8 __attribute__((ext_vector_type(2), aligned(8)));
    typedef unsigned long long u2 __attribute__((ext_vector_type(2)));
    constexpr int R = 16 / (int)sizeof(T);
    constexpr int STEP_BYTES = 8 * R;                // result bytes of one 1-KiB step (64 R rows)
    constexpr int WAVE_BYTES = UNROLL * STEP_BYTES;  // 128 (8-byte elements) .. 1024 (1-byte) for UNROLL = 8
    static_assert(WAVE_BYTES % 16 == 0 && WAVE_BYTES / 16 <= 64, "one 16-byte store per lane");
    __shared__ __attribute__((aligned(16))) uint8_t staged[2][kWaves][WAVE_BYTES];
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const V* __restrict__ vp = (const V*)data;
    // The words of tile k are stored AFTER the loads of a later tile have been issued: vmcnt counts loads and stores in issue
    // order, so a store issued in front of a tile's loads has to be acknowledged before the first of them can be consumed;
    // issued behind them it is never waited for. And the NEXT tile's rows are requested before this tile's are compared
    // (round 4; the masked sums' scheme, ma_reduce.hip): a request is always UNROLL loads — this workgroup's first tile once
    // more when nothing is left, so that the compiler's in-order wait counts see one queue length on every path — into two
    // register sets that swap roles.
    auto flush = [&](int which, size_t at_step0) {
        if (lane < (unsigned)(WAVE_BYTES / 16)) {
            const u2 v = *(const u2*)(staged[which][wave] + lane * 16);
            __builtin_nontemporal_store(v, (u2a8*)((uint8_t*)out + at_step0 * STEP_BYTES + (size_t)lane * 16));  // `out` is 8-byte aligned
        }
    };
    const size_t G = gridDim.x, first = blockIdx.x;
    const size_t n_mine = first < n_tiles ? (n_tiles - first + G - 1) / G : 0;  // this workgroup's tiles
    if (n_mine == 0) return;
    auto issue = [&](size_t k, VL (&x)[UNROLL], size_t& step0) {
        const bool real = k < n_mine;
        step0 = ((first + (real ? k * G : 0)) * kWaves + wave) * UNROLL;  // index of this wave's first 1-KiB step
        const size_t stride = real ? 64 : 0;
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) x[u] = load16<VL, true>((const VL*)vp + step0 * 64 + (size_t)u * stride + lane);
    };
    auto use = [&](const VL (&x)[UNROLL], int par) {
        uint8_t* mine = staged[par][wave];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const unsigned bits = eq_bits<T>(x[u], field_mask, target);
            if constexpr (R == 16) {
                ((uint16_t*)mine)[u * 64 + lane] = (uint16_t)bits;
            } else if constexpr (R == 8) {
                mine[u * 64 + lane] = (uint8_t)bits;
            } else {  // 2 or 4 bits per lane: the 8 / R lanes of a byte combine first
                constexpr int LPB = 8 / R;
                unsigned b = bits << ((lane % LPB) * R);
                // neighbours inside a quad: DPP quad_perm [1,0,3,2] (lane ^ 1) and [2,3,0,1] (lane ^ 2) — no LDS round trip
                b |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)b, 0xB1, 0xF, 0xF, false);
                if constexpr (LPB == 4) b |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)b, 0x4E, 0xF, 0xF, false);
                if (lane % LPB == 0) mine[u * STEP_BYTES + lane / LPB] = (uint8_t)b;
            }
        }
        // the wave reads back what ITS lanes wrote: LDS operations of one wave complete in order, the fences keep the
        // compiler from moving the LDS loads of the flush above these stores (two buffers: the next tile writes the other)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    VL xa[UNROLL], xb[UNROLL];
    size_t step_a = 0, step_b = 0;
    issue(0, xa, step_a);
    for (size_t k = 0; k < n_mine; k += 2) {
        const size_t done_b = step_b;   // tile k - 1 sits in staged[1]
        issue(k + 1, xb, step_b);
        if (k > 0) flush(1, done_b);
        use(xa, 0);                     // tile k
        const size_t done_a = step_a;
        issue(k + 2, xa, step_a);
        flush(0, done_a);
        if (k + 1 < n_mine) use(xb, 1);  // tile k + 1
    }
    if ((n_mine & 1) == 0) flush(1, step_b);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void eq_mask_kernel(const T* __restrict__ data, size_t first_row, size_t n,
                                                         T field_mask, T target, uint64_t* __restrict__ out) {
    const unsigned lane = threadIdx.x & 63;
    const size_t wave_id = ((size_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * kBlock) >> 6;
    const size_t n_words = (n + 63) >> 6;
    for (size_t w = (first_row >> 6) + wave_id; w < n_words; w += n_waves) {
        size_t i = w * 64 + lane;
        bool hit = i < n && (T)(data[i] & field_mask) == target;
        unsigned long long word = __ballot(hit);
        if (lane == 0) out[w] = word;
    }
}

template <typename T, int UNROLL>
static size_t launch_eq_mask_vec(ma_ctx* ctx, const T* d, size_t n, T field_mask, T target, uint64_t* ow, int bpc) {
    constexpr size_t kTileRows = (size_t)64 * (16 / sizeof(T)) * UNROLL * kWaves;
    if (((uintptr_t)d & 15) != 0 || n < kTileRows) return 0;
    const size_t n_tiles = n / kTileRows;
    int grid = grid_for(ctx, n_tiles, bpc);
    hipLaunchKernelGGL((eq_mask_vec_kernel<T, UNROLL>), dim3(grid), dim3(kBlock), 0, ctx->stream, d, n_tiles, field_mask, target,
                       ow);
    return n_tiles * kTileRows;  // a multiple of 64: the tail starts on a word
}

template <typename T>
static void launch_eq_mask(ma_ctx* ctx, const T* d, size_t n, T field_mask, T target, uint64_t* ow) {
    // variant bit 2048: 4 loads per lane on 8 workgroups per CU (round 2's shape) for A/B; blocks_per_cu overrides the grid
    const bool old_shape = (tuning_variant(ctx) & 2048) != 0;
    const int bpc = ctx->blocks_per_cu > 0 ? ctx->blocks_per_cu : (old_shape ? 8 : 1);  // with a tile requested ahead: one, for every width
    size_t done = 0;
    if constexpr (MA_TUNING) {
        done = old_shape ? launch_eq_mask_vec<T, 4>(ctx, d, n, field_mask, target, ow, bpc)
                         : launch_eq_mask_vec<T, 8>(ctx, d, n, field_mask, target, ow, bpc);
    } else {
        done = launch_eq_mask_vec<T, 8>(ctx, d, n, field_mask, target, ow, bpc);
    }
    if (done < n) {
        const size_t n_words = ((n - done) + 63) >> 6;
        int grid = grid_for(ctx, (n_words + kWaves - 1) / kWaves, 8);
        hipLaunchKernelGGL((eq_mask_kernel<T>), dim3(grid), dim3(kBlock), 0, ctx->stream, d, done, n, field_mask, target,
                           ow);
    }
}

struct BitScan {
    unsigned long long pop, any_set, any_clear, any_diff;
};

static void fill_window(BitArgs& a, const uint64_t* lw, size_t lo, const uint64_t* rw, size_t ro, size_t n) {
    a.lw = lw;
    a.lo = lo;
    a.l_last = n ? (lo + n - 1) >> 6 : 0;
    a.rw = rw;
    a.ro = ro;
    a.r_last = n ? (ro + n - 1) >> 6 : 0;
    a.n = n;
}

// Runs bit_scan_kernel and brings the four words back to the host (synchronises the stream).
// Two-words-per-lane path: windows start on a byte (the granularity at which and/or/xor/not address their windows in
// the reference anyway, bitmask/mod.rs:124-128); inputs are read with byte-aligned 16-byte loads, the output (always
// re-based to bit 0) must sit on a 16-byte boundary.
static int vec_ok(const BitArgs& a, bool with_out) {
    if (!with_out && ((a.lo & 7) || (a.rw && (a.ro & 7)))) return 0;  // the scan kernel reads byte-aligned windows only
    if (with_out && ((uintptr_t)a.out & 15)) return 0;                 // the word kernel funnels any bit offset
    return a.n >= 256 ? 1 : 0;
}

static ma_status scan_windows(ma_ctx* ctx, const BitArgs& a, BitScan* out, bool staged) {
    MA_NO_CAPTURE(ctx, "a bitmap scan that returns its result to the host");
    ScanOut so;
    so.partials = ctx->partials;
    so.ticket = ctx->ticket;
    ResultSlot* slot = ctx->result;
    so.host = (unsigned long long*)&slot[3];  // one 32-byte slot
    volatile uint64_t* done = (volatile uint64_t*)&slot[2].b;
    so.done_word = (uint64_t*)done;
    so.done_seq = ++ctx->result_seq;
    const size_t n_words = (a.n + 63) >> 6;
    const int vec = vec_ok(a, false);
    // vec: read-only stream like the sums (8 x 16 bytes per lane in flight)
    // one window of 256 MiB or more: ONE workgroup per CU, the dense sums' shape (a 4-GiB popcount 601 us call to result against
    // 626-636 at two; below that, and for two windows — sixteen loads per lane — two: tools/ab_bit_scan.py)
    const int scan_bpc = ctx->blocks_per_cu > 0 ? ctx->blocks_per_cu : ((a.rw == nullptr && n_words >= ((size_t)1 << 25)) ? 1 : 2);
    int grid = vec ? grid_for(ctx, n_words / 2 / ((size_t)kVecUnroll * kBlock) + 1, scan_bpc)
                   : grid_for(ctx, (n_words + kBlock - 1) / kBlock, 8);
    if (vec) hipLaunchKernelGGL(bit_scan_kernel<true>, dim3(grid), dim3(kBlock), 0, ctx->stream, a, so);
    else hipLaunchKernelGGL(bit_scan_kernel<false>, dim3(grid), dim3(kBlock), 0, ctx->stream, a, so);
    MA_HIP(hipGetLastError());
    // The result always comes back to the host: poll the completion word for a while (a small bitmap — a null count —
    // is done in a few microseconds; hipStreamSynchronize's wake-up alone costs ~10), then block.
    bool landed = false;
    if (!staged && ctx->poll_us > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0;; ++spins) {
            if (*done == so.done_seq) {
                landed = true;
                break;
            }
            if ((spins & 63) == 63 &&
                std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() >= ctx->poll_us)
                break;
            __builtin_ia32_pause();
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    }
    if (!landed) MA_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(out, (const void*)so.host, sizeof(BitScan));
    return MA_OK;
}

static ma_status launch_words(ma_ctx* ctx, const BitArgs& a) {
    const size_t n_words = (a.n + 63) >> 6;
    const int vec = vec_ok(a, true);
    // vec: a store stream in the mix, the launch shape of the elementwise kernels (6 workgroups per CU)
    int grid = vec ? grid_for(ctx, n_words / 2 / ((size_t)kVecUnroll * kBlock) + 1, 6)
                   : grid_for(ctx, (n_words + kBlock - 1) / kBlock, 8);
    if (vec) hipLaunchKernelGGL(bit_words_kernel<true>, dim3(grid), dim3(kBlock), 0, ctx->stream, a);
    else hipLaunchKernelGGL(bit_words_kernel<false>, dim3(grid), dim3(kBlock), 0, ctx->stream, a);
    MA_HIP(hipGetLastError());
    return MA_OK;
}

// Output validity of an op with two input validities (ma_binary.hpp): (window 1) AND / OR (window 2), re-based to bit 0.
ma_status launch_mask_combine(ma_ctx* ctx, const uint64_t* w1, size_t off1, const uint64_t* w2, size_t off2, size_t n,
                              bool is_and, uint64_t* out_words) {
    if (n == 0) return MA_OK;
    BitArgs a{};
    fill_window(a, w1, off1, w2, off2, n);
    a.out = out_words;
    a.op = is_and ? kBitAnd : kBitOr;
    return launch_words(ctx, a);
}

// One entry for every word-producing op. `lhs_round` / `rhs_round`: the reference addresses these windows at
// byte (8) or word (64) granularity; the offset is rounded DOWN accordingly so results match it bit for bit.
static ma_status words_op(ma_ctx* ctx, int op, const uint8_t* lhs, size_t lo, const uint8_t* rhs, size_t ro, size_t len,
                          uint8_t* out_bits, size_t round) {
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    if (len == 0) return MA_OK;
    MA_REQUIRE(lhs != nullptr && out_bits != nullptr, MA_ERR_INVALID_ARGUMENT, "NULL bitmap");
    const bool binary = op == kBitAnd || op == kBitOr || op == kBitXor || op == kBitXnor;
    MA_REQUIRE(!binary || rhs != nullptr, MA_ERR_INVALID_ARGUMENT, "NULL rhs bitmap");
    lo -= lo % round;
    ro -= ro % round;
    MA_ENTER(ctx);
    MA_HIP(hipSetDevice(ctx->device));
    CallScope scope(ctx);
    BitArgs a{};
    const uint64_t *lw = nullptr, *rw = nullptr;
    size_t lo2 = 0, ro2 = 0;
    MA_TRY(scope.in_mask(lhs, lo, len, &lw, &lo2));
    if (binary) MA_TRY(scope.in_mask(rhs, ro, len, &rw, &ro2));
    fill_window(a, lw, lo2, rw, ro2, len);
    MA_TRY(scope.out_mask(out_bits, len, &a.out));
    a.op = op;
    MA_TRY(launch_words(ctx, a));
    return end_call(ctx, scope);
}

static ma_status fill_bits(ma_ctx* ctx, uint8_t* out_bits, size_t len, bool value) {
    // Bitmask::new_set_all(len, value) — src/structs/bitmask.rs:94-105
    MA_ENTER(ctx);
    MA_NO_CAPTURE(ctx, "a constant bitmap fill");
    MA_HIP(hipSetDevice(ctx->device));
    CallScope scope(ctx);
    uint64_t* ow = nullptr;
    MA_TRY(scope.out_mask(out_bits, len, &ow));
    const size_t n_words = (len + 63) >> 6;
    MA_HIP(hipMemsetAsync(ow, value ? 0xFF : 0, n_words * 8, ctx->stream));
    if (value && (len & 63)) {  // trailing bits of the last word are zero: whole bytes, the partial byte, the zero bytes
        uint8_t* last = (uint8_t*)(ow + n_words - 1);
        const size_t tail = len & 63, whole = tail >> 3;
        if (tail & 7) MA_HIP(hipMemsetAsync(last + whole, (1 << (tail & 7)) - 1, 1, ctx->stream));
        const size_t first_zero = whole + ((tail & 7) ? 1 : 0);
        if (first_zero < 8) MA_HIP(hipMemsetAsync(last + first_zero, 0, 8 - first_zero, ctx->stream));
    }
    return end_call(ctx, scope);
}

static ma_status scan_op(ma_ctx* ctx, const uint8_t* lhs, size_t lo, const uint8_t* rhs, size_t ro, size_t len,
                         BitScan* out) {
    MA_ENTER(ctx);
    MA_HIP(hipSetDevice(ctx->device));
    CallScope scope(ctx);
    BitArgs a{};
    const uint64_t *lw = nullptr, *rw = nullptr;
    size_t lo2 = 0, ro2 = 0;
    MA_TRY(scope.in_mask(lhs, lo, len, &lw, &lo2));
    if (rhs) MA_TRY(scope.in_mask(rhs, ro, len, &rw, &ro2));
    fill_window(a, lw, lo2, rw, ro2, len);
    MA_TRY(scan_windows(ctx, a, out, scope.staged()));
    return MA_OK;
}

}  // namespace ma

using namespace ma;

extern "C" {

// ---- and / or / xor / not — dispatch.rs:47-144 -------------------------------------------------------
ma_status ma_bitmask_binop(ma_ctx* ctx, int32_t logical_op, const uint8_t* lhs_bits, size_t lhs_offset,
                           const uint8_t* rhs_bits, size_t rhs_offset, size_t len, uint8_t* out_bits) {
    MA_REQUIRE(logical_op >= MA_LOGICAL_AND && logical_op <= MA_LOGICAL_XOR, MA_ERR_INVALID_ARGUMENT,
               "unknown LogicalOperator code %d", logical_op);
    // bitmask_window_bytes starts the window at byte offset/8 (bitmask/mod.rs:124-128)
    return words_op(ctx, logical_op, lhs_bits, lhs_offset, rhs_bits, rhs_offset, len, out_bits, 8);
}
ma_status ma_and_masks(ma_ctx* ctx, const uint8_t* lhs_bits, size_t lhs_offset, const uint8_t* rhs_bits,
                       size_t rhs_offset, size_t len, uint8_t* out_bits) {
    return words_op(ctx, kBitAnd, lhs_bits, lhs_offset, rhs_bits, rhs_offset, len, out_bits, 8);
}
ma_status ma_or_masks(ma_ctx* ctx, const uint8_t* lhs_bits, size_t lhs_offset, const uint8_t* rhs_bits,
                      size_t rhs_offset, size_t len, uint8_t* out_bits) {
    return words_op(ctx, kBitOr, lhs_bits, lhs_offset, rhs_bits, rhs_offset, len, out_bits, 8);
}
ma_status ma_xor_masks(ma_ctx* ctx, const uint8_t* lhs_bits, size_t lhs_offset, const uint8_t* rhs_bits,
                       size_t rhs_offset, size_t len, uint8_t* out_bits) {
    return words_op(ctx, kBitXor, lhs_bits, lhs_offset, rhs_bits, rhs_offset, len, out_bits, 8);
}
ma_status ma_not_mask(ma_ctx* ctx, const uint8_t* src_bits, size_t offset, size_t len, uint8_t* out_bits) {
    return words_op(ctx, kBitNot, src_bits, offset, nullptr, 0, len, out_bits, 8);
}

// Bitmask::slice_clone(offset, len): bit-accurate copy of a window to bit 0.
ma_status ma_bitmask_slice(ma_ctx* ctx, const uint8_t* src_bits, size_t offset, size_t len, uint8_t* out_bits) {
    return words_op(ctx, kBitCopy, src_bits, offset, nullptr, 0, len, out_bits, 1);
}

// ---- eq / ne — simd.rs:402-472: offsets must be multiples of 64 (the reference panics otherwise) -----------
ma_status ma_eq_mask(ma_ctx* ctx, const uint8_t* a_bits, size_t a_offset, const uint8_t* b_bits, size_t b_offset,
                     size_t len, uint8_t* out_bits) {
    if (len == 0) return MA_OK;
    MA_REQUIRE(a_offset % 64 == 0 && b_offset % 64 == 0, MA_ERR_INVALID_ARGUMENT,
               "eq_bits_mask: offsets must be 64-bit aligned (got a: %zu, b: %zu)", a_offset, b_offset);
    return words_op(ctx, kBitXnor, a_bits, a_offset, b_bits, b_offset, len, out_bits, 64);
}
ma_status ma_ne_mask(ma_ctx* ctx, const uint8_t* a_bits, size_t a_offset, const uint8_t* b_bits, size_t b_offset,
                     size_t len, uint8_t* out_bits) {
    if (len == 0) return MA_OK;
    MA_REQUIRE(a_offset % 64 == 0 && b_offset % 64 == 0, MA_ERR_INVALID_ARGUMENT,
               "eq_bits_mask: offsets must be 64-bit aligned (got a: %zu, b: %zu)", a_offset, b_offset);
    return words_op(ctx, kBitXor, a_bits, a_offset, b_bits, b_offset, len, out_bits, 64);
}

// ---- all_eq / all_ne — simd.rs:490-581 -------------------------------------------------------------------
ma_status ma_all_eq(ma_ctx* ctx, const uint8_t* a_bits, size_t a_offset, const uint8_t* b_bits, size_t b_offset,
                    size_t len, int32_t* out_bool) {
    MA_REQUIRE(ctx != nullptr && out_bool != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx or out is NULL");
    *out_bool = 1;
    if (len == 0) return MA_OK;
    MA_REQUIRE(a_bits && b_bits, MA_ERR_INVALID_ARGUMENT, "NULL bitmap");
    // len < 64 compares the single words at offset/64 under a low-bit mask (simd.rs:523-528); longer windows
    // must start on a word (simd.rs:530-535). Either way the window starts at word offset/64.
    MA_REQUIRE(len < 64 || (a_offset % 64 == 0 && b_offset % 64 == 0), MA_ERR_INVALID_ARGUMENT,
               "all_eq_mask_simd: offsets must be 64-bit aligned (got a: %zu, b: %zu)", a_offset, b_offset);
    BitScan s{};
    MA_TRY(scan_op(ctx, a_bits, a_offset - a_offset % 64, b_bits, b_offset - b_offset % 64, len, &s));
    *out_bool = s.any_diff ? 0 : 1;
    return MA_OK;
}
ma_status ma_all_ne(ma_ctx* ctx, const uint8_t* a_bits, size_t a_offset, const uint8_t* b_bits, size_t b_offset,
                    size_t len, int32_t* out_bool) {
    // all_ne_mask_simd = !all_eq_mask_simd (simd.rs:490-494): "not all equal"
    MA_TRY(ma_all_eq(ctx, a_bits, a_offset, b_bits, b_offset, len, out_bool));
    *out_bool = !*out_bool;
    return MA_OK;
}

// ---- popcount — simd.rs:596-644: counting starts at WORD offset/64 -------------------------------------------
ma_status ma_popcount_mask(ma_ctx* ctx, const uint8_t* bits, size_t offset, size_t len, uint64_t* out_count) {
    MA_REQUIRE(ctx != nullptr && out_count != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx or out is NULL");
    *out_count = 0;
    if (len == 0) return MA_OK;
    MA_REQUIRE(bits != nullptr, MA_ERR_INVALID_ARGUMENT, "NULL bitmap");
    BitScan s{};
    MA_TRY(scan_op(ctx, bits, offset - offset % 64, nullptr, 0, len, &s));
    *out_count = s.pop;
    return MA_OK;
}

// ---- all_true / all_false — std.rs:300-366 (every logical bit set / clear) -----------------------------------
ma_status ma_all_true_mask(ma_ctx* ctx, const uint8_t* bits, size_t len, int32_t* out_bool) {
    MA_REQUIRE(ctx != nullptr && out_bool != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx or out is NULL");
    *out_bool = 1;
    if (len == 0) return MA_OK;
    MA_REQUIRE(bits != nullptr, MA_ERR_INVALID_ARGUMENT, "NULL bitmap");
    BitScan s{};
    MA_TRY(scan_op(ctx, bits, 0, nullptr, 0, len, &s));
    *out_bool = s.any_clear ? 0 : 1;
    return MA_OK;
}
ma_status ma_all_false_mask(ma_ctx* ctx, const uint8_t* bits, size_t len, int32_t* out_bool) {
    MA_REQUIRE(ctx != nullptr && out_bool != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx or out is NULL");
    *out_bool = 1;
    if (len == 0) return MA_OK;
    MA_REQUIRE(bits != nullptr, MA_ERR_INVALID_ARGUMENT, "NULL bitmap");
    BitScan s{};
    MA_TRY(scan_op(ctx, bits, 0, nullptr, 0, len, &s));
    *out_bool = s.any_set ? 0 : 1;
    return MA_OK;
}

// ---- in / not_in — simd.rs:327-398 ----------------------------------------------------------------------
ma_status ma_in_mask(ma_ctx* ctx, const uint8_t* lhs_bits, size_t lhs_offset, const uint8_t* rhs_bits,
                     size_t rhs_offset, size_t len, uint8_t* out_bits) {
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    if (len == 0) return MA_OK;
    MA_REQUIRE(lhs_bits && rhs_bits && out_bits, MA_ERR_INVALID_ARGUMENT, "NULL bitmap");
    // Which boolean values occur in rhs? The reference scans words from rhs_off/64 (simd.rs:345).
    BitScan s{};
    MA_TRY(scan_op(ctx, rhs_bits, rhs_offset - rhs_offset % 64, nullptr, 0, len, &s));
    if (s.any_set && s.any_clear) return fill_bits(ctx, out_bits, len, true);                 // both: every bit is a member
    if (s.any_set) return ma_bitmask_slice(ctx, lhs_bits, lhs_offset, len, out_bits);          // lhs.slice_clone
    if (s.any_clear) return ma_not_mask(ctx, lhs_bits, lhs_offset, len, out_bits);             // not_mask_simd
    return fill_bits(ctx, out_bits, len, false);
}
ma_status ma_not_in_mask(ma_ctx* ctx, const uint8_t* lhs_bits, size_t lhs_offset, const uint8_t* rhs_bits,
                         size_t rhs_offset, size_t len, uint8_t* out_bits) {
    MA_TRY(ma_in_mask(ctx, lhs_bits, lhs_offset, rhs_bits, rhs_offset, len, out_bits));
    if (len == 0) return MA_OK;
    return ma_not_mask(ctx, out_bits, 0, len, out_bits);  // word-for-word in place
}

// ---- merge_bitmasks_to_new — bitmask/mod.rs:171-196: AND of optional masks, both at bit 0 -------------------
ma_status ma_merge_bitmasks_to_new(ma_ctx* ctx, const uint8_t* lhs_bits, const uint8_t* rhs_bits, size_t len,
                                   uint8_t* out_bits, int32_t* out_is_some) {
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    if (out_is_some) *out_is_some = (lhs_bits || rhs_bits) ? 1 : 0;
    if (!lhs_bits && !rhs_bits) return MA_OK;  // (None, None) => None
    if (len == 0) return MA_OK;
    if (lhs_bits && rhs_bits) return words_op(ctx, kBitAnd, lhs_bits, 0, rhs_bits, 0, len, out_bits, 1);
    return words_op(ctx, kBitCopy, lhs_bits ? lhs_bits : rhs_bits, 0, nullptr, 0, len, out_bits, 1);
}

// ---- simd_eq_mask_u{8,16,32,64} — simd.rs:741-788 ------------------------------------------------------
#define MA_DEFINE_EQ_MASK(TAG, T)                                                                                \
    ma_status ma_simd_eq_mask_##TAG(ma_ctx* ctx, const T* data, size_t n, T field_mask, T target,                 \
                                    uint8_t* out_bits) {                                                         \
        MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");                                       \
        if (n == 0) return MA_OK;                                                                                \
        MA_REQUIRE(data != nullptr && out_bits != nullptr, MA_ERR_INVALID_ARGUMENT, "NULL buffer");               \
        MA_REQUIRE(((uintptr_t)data % sizeof(T)) == 0, MA_ERR_INVALID_ARGUMENT, "misaligned data pointer");       \
        MA_ENTER(ctx);                                                               \
        MA_HIP(hipSetDevice(ctx->device));                                                                       \
        CallScope scope(ctx);                                                                                    \
        const void* d = nullptr;                                                                                 \
        MA_TRY(scope.in(data, n * sizeof(T), &d));                                                               \
        uint64_t* ow = nullptr;                                                                                  \
        MA_TRY(scope.out_mask(out_bits, n, &ow));                                                                \
        launch_eq_mask<T>(ctx, (const T*)d, n, field_mask, target, ow);                                          \
        MA_HIP(hipGetLastError());                                                                               \
        return end_call(ctx, scope);                                                                             \
    }
MA_DEFINE_EQ_MASK(u8, uint8_t)
MA_DEFINE_EQ_MASK(u16, uint16_t)
MA_DEFINE_EQ_MASK(u32, uint32_t)
MA_DEFINE_EQ_MASK(u64, uint64_t)

}  // extern "C"
