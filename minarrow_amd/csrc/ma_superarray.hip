// Batched chunk fan-out: SuperArray (op) SuperArray in ONE launch (a few for very long chunk lists).
//
// route_super_array_broadcast (src/kernels/broadcast/super_array.rs:180-251) loops over chunk pairs sequentially
// ("// TODO: Parallelise", :193). SuperArrays are rechunked to 8192 rows by default
// (RechunkStrategy::Auto, src/structs/chunked/super_array.rs:51-59): a 10^9-row column is ~122 000 chunks, and a
// launch per chunk would be launch-bound by three orders of magnitude. Here compact descriptor tables are uploaded — 32
// bytes per chunk pair (+32 when the call carries validity) plus a tile prefix sum — and every workgroup binary-searches
// its tile's chunk in the prefix array (like concat_kernel) and runs the same 16-byte vector body as the single-array
// kernels (inputs on any element phase), or a row body for ragged tiles.
// Validity: the common mask of a chunk is lhs | rhs (Bitmask::union, :224) or whichever side has one; the wave that
// computes a run writes its words of the output bitmap, or a second launch assembles every chunk's bitmap word by word.
//
// Round 3, from a kernel trace of 60 000 x 8192-row chunk pairs (profiles/r03_super_array_trace.txt): the tile kernel
// itself runs at 1.0-1.07 of the same-process copy rate — the search is not what the chunked regime pays for. What it
// paid was (a) the 104-byte-per-chunk table crossing PCIe ON the stream in front of the kernel (6.2 MB = 0.11 ms of a
// 1.05 ms kernel) and (b) the host building that table (0.48 ms) before anything was enqueued. Hence the compact tables
// (2.4 MB dense), and long chunk lists go in SEGMENTS of growing size: the GPU starts on the first 4096 chunks while the
// host describes the next segment. And RechunkStrategy-sized chunk lists take a second kernel, chunk_binary_kernel: a whole
// chunk per workgroup, no search, its 32-byte descriptors read straight from the pinned staging buffer with wave-uniform
// loads a chunk ahead of their use — no copy on the stream at all. As a kernel it is 4-8 % slower than the tile kernel
// (every workgroup walks its own chunk: a four times wider access front), end to end it wins everywhere but dense f64,
// where the two tie: see batched_impl for the figures.
#include <vector>

#include "ma_binary.hpp"

namespace ma {

struct PairDesc {
    const void* lhs;
    const void* rhs;
    void* out;
    const uint64_t* lw;   // lhs validity words or nullptr
    size_t lo, l_last;
    const uint64_t* rw;   // rhs validity words or nullptr
    size_t ro, r_last;
    uint64_t* ow;         // output validity words or nullptr
    size_t len;
    size_t tile0;         // first tile of this chunk
    size_t word0;         // first output-bitmap word of this chunk in the global word numbering
    unsigned head;        // rows before the first 16-byte boundary of `out`
};

struct ChunkPair {  // 32 bytes
    const void* lhs;
    const void* rhs;
    void* out;
    uint64_t len;
};
struct ChunkMaskDesc {  // 32 bytes; entry c belongs to ChunkPair c
    const uint64_t* lw;  // 8-byte aligned word pointers (or nullptr) ...
    const uint64_t* rw;
    uint64_t* ow;
    uint32_t lo, ro;     // ... and the bit the chunk's validity starts at within the first word (0..63)
};


// The in-register description pair_tile works on, from the compact table entries.
template <typename T>
__device__ __forceinline__ PairDesc make_pair(const ChunkPair& e, const ChunkMaskDesc& m) {
    PairDesc d;
    d.lhs = e.lhs;
    d.rhs = e.rhs;
    d.out = e.out;
    d.len = (size_t)e.len;
    d.lw = m.lw;  // an all-zero entry = no validity on either side
    d.rw = m.rw;
    d.ow = m.ow;
    d.lo = m.lo;
    d.ro = m.ro;
    d.l_last = d.len ? (d.lo + d.len - 1) >> 6 : 0;
    d.r_last = d.len ? (d.ro + d.len - 1) >> 6 : 0;
    const unsigned mis = (unsigned)((uintptr_t)e.out & 15);
    d.head = mis ? (16 - mis) / (unsigned)sizeof(T) : 0;
    d.tile0 = 0;
    d.word0 = 0;
    return d;
}

// The chunk whose prefix value (first tile / first bitmap word) is the last one <= x: a dense u64 array, 8 bytes per
// chunk — the search touches nothing else.
__device__ __forceinline__ int find_by_prefix(const uint64_t* __restrict__ prefix, int n, size_t x) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (prefix[mid] <= x) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

__device__ __forceinline__ uint64_t window_word_at(const uint64_t* __restrict__ words, size_t bit_off, size_t last_word,
                                                   size_t j) {
    const size_t b = bit_off + (j << 6);
    const size_t w = b >> 6;
    const unsigned sh = (unsigned)(b & 63);
    uint64_t lo = w <= last_word ? words[w] : 0;
    if (sh == 0) return lo;
    uint64_t hi = (w + 1) <= last_word ? words[w + 1] : 0;
    return (lo >> sh) | (hi << (64 - sh));
}

// common validity of chunk row `row` (64-row word containing it, shifted so that bit 0 = row)
__device__ __forceinline__ unsigned pair_row_valid(const PairDesc& d, size_t row) {
    unsigned v = 0;
    bool any = false;
    if (d.lw) {
        v |= row_bit(d.lw, d.lo + row);
        any = true;
    }
    if (d.rw) {
        v |= row_bit(d.rw, d.ro + row);
        any = true;
    }
    return any ? v : 1u;
}

// One tile of one chunk pair: the 16-byte vector body of the single-array kernels (stores aligned by construction, inputs
// on any element phase), the guarded form for a ragged last tile, the rows in front of `out`'s first 16-byte boundary
// with tile 0. FUSE_MASK: the wave that computes a run of rows also writes the run's words of the chunk's output bitmap
// (the common mask it has in registers anyway) — possible when every masked chunk's `out` starts on a 16-byte boundary
// (head == 0), so that runs start on validity-word boundaries; otherwise batched_mask_kernel assembles the bitmaps in a
// second launch.
// smode: 0 = array (op) array; 1 = the LEFT operand is the scalar `sval` for every chunk (broadcast_scalar_to_superarray,
// src/kernels/broadcast/scalar.rs:214-243), 2 = the right one (broadcast_superarray_to_scalar, super_array.rs:87-116) —
// wave-uniform: the scalar side's loads are skipped, nothing else changes.
// DV (integer Div / Rem / FloorDiv of a call that carries validity): a masked chunk's OUTPUT validity depends on the data —
// a zero divisor clears the row's bit instead of raising (m & !div_zero, simd.rs:319-326) — so the computing wave packs the
// result bits of its rows into the chunk's output words itself (FUSE_MASK is then always on); dense chunks of the same call
// keep the latch.
template <typename T, int UNROLL, bool FUSE_MASK, bool DV = false>
__device__ __forceinline__ void pair_tile(const PairDesc& d, size_t lt, int op, bool& dz, unsigned lane, unsigned wave,
                                          int smode, T sval) {
    typedef typename Vec16<T>::type V;
    constexpr int R = 16 / (int)sizeof(T);
    constexpr int WPT = R * UNROLL;
    constexpr size_t WAVE_ROWS = (size_t)64 * R * UNROLL;
    constexpr size_t TILE_ROWS = WAVE_ROWS * kWaves;
    // global, not flat: these pointers come out of a table (see as_global)
    const auto lhs = as_global((const T*)d.lhs);
    const auto rhs = as_global((const T*)d.rhs);
    const auto out = as_global((T*)d.out);
    const bool masked = d.lw != nullptr || d.rw != nullptr;
    const size_t r0 = d.head + lt * TILE_ROWS;
    const size_t r1 = r0 + TILE_ROWS < d.len ? r0 + TILE_ROWS : d.len;
    if (lt == 0) {
        for (size_t i = threadIdx.x; i < d.head && i < d.len; i += kBlock) {
            T v = Elem<T>::apply_rt(op, smode == 1 ? sval : lhs[i], smode == 2 ? sval : rhs[i], dz);
            if (masked) v = pair_row_valid(d, i) ? v : (T)0;
            out[i] = v;
        }
    }
    V sv;
#pragma unroll
    for (int k = 0; k < R; ++k) sv[k] = sval;
    const size_t w0 = r0 + (size_t)wave * WAVE_ROWS;  // this wave's run of the tile
    if (w0 >= r1) return;
    // Stores are 16-byte aligned by construction; inputs may sit on any element phase (load16u). A ragged last tile
    // runs the same vector body over its whole vectors (loads and stores guarded per vector) and finishes the < R
    // rows that remain one by one; whole runs take the unguarded body.
    const size_t run_rows = r1 - w0 < WAVE_ROWS ? r1 - w0 : WAVE_ROWS;
    const V* __restrict__ p = (const V*)((const T*)d.lhs + w0) + lane;
    const V* __restrict__ q = (const V*)((const T*)d.rhs + w0) + lane;
    V* __restrict__ o = (V*)((T*)d.out + w0) + lane;
    uint64_t aw = ~(uint64_t)0;
    if (masked) {
        aw = 0;
        if (d.lw) aw |= load_run_words<WPT>(d.lw, d.lo + w0, d.l_last, lane);
        if (d.rw) aw |= load_run_words<WPT>(d.rw, d.ro + w0, d.r_last, lane);
        if (FUSE_MASK && !DV && lane < (unsigned)WPT) {  // lane k holds run word k = word w0 / 64 + k of the chunk's bitmap
            const size_t j = (w0 >> 6) + lane;
            const size_t first = j << 6;
            if (first < d.len) {
                uint64_t w = aw;
                if (d.len - first < 64) w &= (((uint64_t)1) << (d.len - first)) - 1;
                as_global(d.ow)[j] = w;
            }
        }
    }
    if (run_rows == WAVE_ROWS) {
        V va[UNROLL], vb[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) va[u] = smode == 1 ? sv : load16u<V, true>(p + (size_t)u * 64);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) vb[u] = smode == 2 ? sv : load16u<V, true>(q + (size_t)u * 64);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            unsigned bits = ~0u;
            if (masked) bits = lane_bits<R>(aw, u, lane);
            V r;
            [[maybe_unused]] unsigned out_bits = bits;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                T v;
                if constexpr (DV) {
                    bool dzk = false;
                    v = Elem<T>::apply_rt(op, (T)va[u][k], (T)vb[u][k], dzk);
                    if (masked) out_bits &= ~((dzk ? 1u : 0u) << k);
                    else dz |= dzk;
                } else {  // exactly the round-2 form: a per-element flag cost the masked i32 kernels 5-17 %
                    v = Elem<T>::apply_rt(op, (T)va[u][k], (T)vb[u][k], dz);
                }
                v = ((bits >> k) & 1u) ? v : (T)0;  // a zero divisor already gave 0
                r[k] = v;
            }
            store16<V, true>(o + (size_t)u * 64, r);
            if constexpr (DV) {
                if (masked) {  // wave-uniform: the run's 64 R rows of step u are exactly R words of the chunk's output bitmap
                    constexpr int LPW = 64 / R;
                    const uint64_t word = pack_lane_bits<R>(out_bits & ((1u << R) - 1u), lane);
                    if (lane % LPW == 0) as_global(d.ow)[(w0 >> 6) + (size_t)u * R + lane / LPW] = word;
                }
            }
        }
    } else if (DV && masked) {
        // the chunk's ragged last run: one row per lane, one ballot per 64 rows (its words hold bits of vector and of tail rows)
        const size_t end = w0 + run_rows;
        for (size_t base = w0; base < end; base += 64) {
            const size_t i = base + lane;
            bool ok = false;
            T v = (T)0;
            if (i < end && pair_row_valid(d, i)) {
                bool dzk = false;
                v = Elem<T>::apply_rt(op, smode == 1 ? sval : lhs[i], smode == 2 ? sval : rhs[i], dzk);
                ok = !dzk;
            }
            const unsigned long long word = __ballot(ok);
            if (i < end) out[i] = v;
            if (lane == 0) as_global(d.ow)[base >> 6] = word;
        }
    } else {
        const unsigned n_vec = (unsigned)(run_rows / R);
        for (int u = 0; u < UNROLL; ++u) {
            unsigned bits = ~0u;
            if (masked) bits = lane_bits<R>(aw, u, lane);  // wave-wide shuffle: outside the per-lane guard
            if ((unsigned)u * 64 + lane < n_vec) {
                const V a = smode == 1 ? sv : load16u<V, true>(p + (size_t)u * 64);
                const V b = smode == 2 ? sv : load16u<V, true>(q + (size_t)u * 64);
                V r;
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    T v = Elem<T>::apply_rt(op, (T)a[k], (T)b[k], dz);
                    v = ((bits >> k) & 1u) ? v : (T)0;
                    r[k] = v;
                }
                store16<V, true>(o + (size_t)u * 64, r);
            }
        }
        const size_t tail0 = w0 + (size_t)n_vec * R;
        if (tail0 + lane < w0 + run_rows) {
            const size_t i = tail0 + lane;
            T v = Elem<T>::apply_rt(op, smode == 1 ? sval : lhs[i], smode == 2 ? sval : rhs[i], dz);
            if (masked) v = pair_row_valid(d, i) ? v : (T)0;
            out[i] = v;
        }
    }
}

// Tiles dealt round-robin to workgroups (the grid sweeps the chunk list as one contiguous front); each workgroup
// binary-searches its tile's chunk in the tile prefix array.
template <typename T, int UNROLL, bool FUSE_MASK, bool DV = false>
__global__ __launch_bounds__(kBlock) void batched_binary_kernel(const ChunkPair* __restrict__ cd,
                                                                const ChunkMaskDesc* __restrict__ md,
                                                                const uint64_t* __restrict__ tile0, int n_chunks,
                                                                size_t n_tiles, int op, uint32_t* flags, int smode,
                                                                uint64_t sbits) {
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bool dz = false;
    T sval;
    __builtin_memcpy(&sval, &sbits, sizeof(T));
    // The chunk of the previous tile is remembered: with long chunks (the lists this kernel is chosen for) a workgroup's
    // next tile, gridDim.x tiles on, is nearly always in the same chunk, and the search + descriptor fetch — a chain of
    // dependent loads in front of the tile's data — is skipped (8 long i32 chunks against a scalar: 0.89 -> see §3.2b).
    size_t t_lo = 1, t_hi = 0;
    PairDesc d{};
    for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        if (t < t_lo || t >= t_hi) {
            const int c = find_by_prefix(tile0, n_chunks, t);
            const ChunkPair e = cd[c];
            ChunkMaskDesc m{};
            if (md) m = md[c];
            d = make_pair<T>(e, m);
            t_lo = tile0[c];
            t_hi = c + 1 < n_chunks ? tile0[c + 1] : n_tiles;
        }
        pair_tile<T, UNROLL, FUSE_MASK, DV>(d, t - t_lo, op, dz, lane, wave, smode, sval);
    }
    if constexpr (std::is_integral<T>::value) {
        // only dense chunks can latch (masked integer division is routed chunk by chunk on the host)
        if (__any(dz) && lane == 0) atomicOr(flags, 1u);
    }
}

// RechunkStrategy-sized chunks (src/structs/chunked/super_array.rs:51-59: 8192 rows by default — a 10^9-row column is
// 122 000 chunk pairs of one to four tiles): chunk c belongs to workgroup c mod grid, which walks the chunk's tiles
// itself. No search (the tile kernel pays log2(n_chunks) ~ 16 dependent loads per tile, two to four times per chunk), and
// the descriptor is 32 bytes (+32 when the call carries validity) instead of 104, read with wave-uniform (scalar) loads
// one chunk ahead of its use: the table that has to cross PCIe before the kernel may start shrinks from 6.2 MB to
// 1.9 MB for 60 000 dense pairs.
template <typename T, int UNROLL, bool ANY_MASK, bool DV = false>
__global__ __launch_bounds__(kBlock) void chunk_binary_kernel(const ChunkPair* __restrict__ cd,
                                                              const ChunkMaskDesc* __restrict__ md, int n_chunks, int op,
                                                              uint32_t* flags, int smode, uint64_t sbits) {
    constexpr int R = 16 / (int)sizeof(T);
    constexpr size_t TILE_ROWS = (size_t)64 * R * UNROLL * kWaves;
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bool dz = false;
    T sval;
    __builtin_memcpy(&sval, &sbits, sizeof(T));
    int c = blockIdx.x;
    if (c >= n_chunks) return;
    ChunkPair e = cd[c];
    ChunkMaskDesc m{};
    if (ANY_MASK) m = md[c];
    while (true) {
        const int next = c + (int)gridDim.x;
        ChunkPair en{};
        ChunkMaskDesc mn{};
        if (next < n_chunks) {  // wave-uniform addresses: scalar loads, in flight while this chunk's rows stream
            en = cd[next];
            if (ANY_MASK) mn = md[next];
        }
        const PairDesc d = make_pair<T>(e, m);
        if (d.len) {
            const size_t n_t = d.len > d.head ? (d.len - d.head + TILE_ROWS - 1) / TILE_ROWS : 1;
            for (size_t lt = 0; lt < n_t; ++lt) pair_tile<T, UNROLL, ANY_MASK, DV>(d, lt, op, dz, lane, wave, smode, sval);
        }
        if (next >= n_chunks) break;
        c = next;
        e = en;
        m = mn;
    }
    if constexpr (std::is_integral<T>::value) {
        if (__any(dz) && lane == 0) atomicOr(flags, 1u);
    }
}

// One thread per output validity word over all chunks: word = lhs window | rhs window (or the single one).
__global__ __launch_bounds__(kBlock) void batched_mask_kernel(const ChunkPair* __restrict__ cd,
                                                              const ChunkMaskDesc* __restrict__ md,
                                                              const uint64_t* __restrict__ word0, int n_chunks,
                                                              size_t n_words) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t g = (size_t)blockIdx.x * kBlock + threadIdx.x; g < n_words; g += stride) {
        const int c = find_by_prefix(word0, n_chunks, g);
        const ChunkMaskDesc m = md[c];
        if (m.ow == nullptr) continue;
        const size_t len = (size_t)cd[c].len;
        const size_t j = g - word0[c];
        const size_t chunk_words = (len + 63) >> 6;
        if (j >= chunk_words) continue;
        uint64_t w = 0;
        if (m.lw) w |= window_word_at(m.lw, m.lo, (m.lo + len - 1) >> 6, j);
        if (m.rw) w |= window_word_at(m.rw, m.ro, (m.ro + len - 1) >> 6, j);
        if (j == chunk_words - 1 && (len & 63)) w &= (((uint64_t)1) << (len & 63)) - 1;
        m.ow[j] = w;
    }
}

// Makes the buffers of chunk pair i device-reachable and describes them: pointers, validity words + first bit, head rows.
// A chunked column's pointers run through a few allocations: each operand role remembers the device range its last
// pointer fell into, so the common case is two compares per pointer (no classification call — six of those per chunk
// pair were most of the 38 ns a pair cost the host, as long as the kernel itself at 60 000 pairs).
struct PairRoles {
    DeviceRange lhs, rhs, out, lm, rm, om;
};
template <typename T>
static ma_status resolve_pair(CallScope& scope, PairRoles& roles, size_t i, size_t n, const void* lhs, const void* rhs,
                              void* out, const uint8_t* lm, const uint8_t* rm, uint8_t* om, PairDesc& d, int smode) {
    memset(&d, 0, sizeof(d));
    d.len = n;
    if (n == 0) return MA_OK;
    MA_REQUIRE((lhs || smode == 1) && (rhs || smode == 2) && out, MA_ERR_INVALID_ARGUMENT, "chunk %zu: NULL buffer", i);
    if (smode == 1) {
        d.lhs = nullptr;  // the scalar's side: never dereferenced
    } else if (roles.lhs.holds(lhs)) {
        d.lhs = lhs;
    } else {
        MA_TRY(scope.in(lhs, n * sizeof(T), &d.lhs));
        roles.lhs.learn(lhs);
    }
    if (smode == 2) {
        d.rhs = nullptr;
    } else if (roles.rhs.holds(rhs)) {
        d.rhs = rhs;
    } else {
        MA_TRY(scope.in(rhs, n * sizeof(T), &d.rhs));
        roles.rhs.learn(rhs);
    }
    if (roles.out.holds(out)) {
        d.out = out;
    } else {
        MA_TRY(scope.out(out, n * sizeof(T), &d.out));
        roles.out.learn(out);
    }
    if (lm) {
        if (roles.lm.holds(lm)) {
            const uintptr_t addr = (uintptr_t)lm, base = addr & ~(uintptr_t)7;  // CallScope::in_mask's re-basing
            d.lw = (const uint64_t*)base;
            d.lo = (size_t)(addr - base) * 8;
        } else {
            MA_TRY(scope.in_mask(lm, 0, n, &d.lw, &d.lo));
            roles.lm.learn(lm);
        }
        d.l_last = (d.lo + n - 1) >> 6;
    }
    if (rm) {
        if (roles.rm.holds(rm)) {
            const uintptr_t addr = (uintptr_t)rm, base = addr & ~(uintptr_t)7;
            d.rw = (const uint64_t*)base;
            d.ro = (size_t)(addr - base) * 8;
        } else {
            MA_TRY(scope.in_mask(rm, 0, n, &d.rw, &d.ro));
            roles.rm.learn(rm);
        }
        d.r_last = (d.ro + n - 1) >> 6;
    }
    if (lm || rm) {
        MA_REQUIRE(om != nullptr, MA_ERR_INVALID_ARGUMENT, "chunk %zu carries nulls but has no output bitmap", i);
        if (roles.om.holds(om) && ((uintptr_t)om & 7) == 0) {
            d.ow = (uint64_t*)om;
        } else {
            MA_TRY(scope.out_mask(om, n, &d.ow));
            roles.om.learn(om);
        }
    }
    const uintptr_t mis = (uintptr_t)d.out & 15;
    d.head = mis ? (unsigned)((16 - mis) / sizeof(T)) : 0;
    return MA_OK;
}

static ma_status finish_batched(ma_ctx* ctx, CallScope& scope, bool int_div) {
    MA_TRY(end_call(ctx, scope));
    if (int_div && is_async(ctx) && !scope.staged()) {
        ctx->pending_flags = true;  // like the dense kernels of an async context: reported by the next synchronize
        return MA_OK;
    }
    if (int_div) {
        uint32_t f = 0;
        MA_HIP(hipMemcpyAsync(&f, ctx->dev_flags, sizeof(f), hipMemcpyDeviceToHost, ctx->stream));
        MA_HIP(hipStreamSynchronize(ctx->stream));
        if (f & 1u) {
            MA_HIP(hipMemsetAsync(ctx->dev_flags, 0, sizeof(f), ctx->stream));
            MA_HIP(hipStreamSynchronize(ctx->stream));
            set_error("Super Array broadcasting error - division by zero in a dense integer chunk");
            return MA_ERR_DIVIDE_BY_ZERO;
        }
    }
    return MA_OK;
}

// One segment [c0, c1) of the chunk list: tables built in the pinned staging buffer, delivered (copied into `dev_tab`, or
// left in place for the chunk-per-workgroup kernel), kernels enqueued. `by_chunk` / `any_mask` were decided for the whole
// call. Layout of a segment's table: [ChunkPair x n][ChunkMaskDesc x n if any_mask][u64 tile0 x n][u64 word0 x n if any_mask]
// (the prefix arrays only for the tile form).
template <typename T, int U>
static ma_status batched_segment(ma_ctx* ctx, CallScope& scope, PairRoles& roles, int op, size_t c0, size_t c1,
                                 const void* const* lhs_data, const size_t* lens, const uint8_t* const* lhs_masks,
                                 const void* const* rhs_data, const uint8_t* const* rhs_masks, const uint8_t* override_mask,
                                 void* const* out_data, uint8_t* const* out_masks, int32_t* out_has_mask, bool any_mask,
                                 bool by_chunk, char* dev_tab, int smode, uint64_t sbits) {
    constexpr int R = 16 / (int)sizeof(T);
    constexpr size_t TILE_ROWS = (size_t)64 * R * U * kWaves;
    // integer Div / Rem / FloorDiv with validity: the output bitmaps depend on the data (pair_tile's DV form)
    const bool dv = std::is_integral<T>::value && any_mask && (op == MA_OP_DIVIDE || op == MA_OP_REMAINDER || op == MA_OP_FLOORDIV);
    const size_t n = c1 - c0;
    const size_t off_md = sizeof(ChunkPair) * n;
    const size_t off_t0 = off_md + (any_mask ? sizeof(ChunkMaskDesc) * n : 0);
    const size_t off_w0 = off_t0 + (by_chunk ? 0 : 8 * n);
    const size_t bytes = off_w0 + ((any_mask && !by_chunk) ? 8 * n : 0);
    char* host = nullptr;
    MA_TRY(table_begin(ctx, bytes, (void**)&host));
    ChunkPair* cd = (ChunkPair*)host;
    ChunkMaskDesc* md = any_mask ? (ChunkMaskDesc*)(host + off_md) : nullptr;
    uint64_t* tile0 = by_chunk ? nullptr : (uint64_t*)(host + off_t0);
    uint64_t* word0 = (any_mask && !by_chunk) ? (uint64_t*)(host + off_w0) : nullptr;
    size_t n_tiles = 0, n_words = 0, total = 0;
    bool masked_head = false;
    for (size_t k = 0; k < n; ++k) {
        const size_t i = c0 + k;
        const size_t len = lens[i];
        const uint8_t* lm = override_mask ? override_mask : (lhs_masks ? lhs_masks[i] : nullptr);
        const uint8_t* rm = override_mask ? nullptr : (rhs_masks ? rhs_masks[i] : nullptr);
        if (out_has_mask) out_has_mask[i] = (lm || rm) ? 1 : 0;
        PairDesc d;
        MA_TRY(resolve_pair<T>(scope, roles, i, len, lhs_data ? lhs_data[i] : nullptr, rhs_data ? rhs_data[i] : nullptr,
                               out_data[i], lm, rm, out_masks ? out_masks[i] : nullptr, d, smode));
        cd[k] = ChunkPair{d.lhs, d.rhs, d.out, (uint64_t)len};
        if (md) md[k] = ChunkMaskDesc{len ? d.lw : nullptr, len ? d.rw : nullptr, len ? d.ow : nullptr, (uint32_t)d.lo, (uint32_t)d.ro};
        if (tile0) tile0[k] = n_tiles;
        if (word0) word0[k] = n_words;
        if (len == 0) continue;
        if (d.head && (lm || rm)) masked_head = true;
        n_tiles += len > d.head ? (len - d.head + TILE_ROWS - 1) / TILE_ROWS : 1;
        n_words += (len + 63) >> 6;
        total += len;
    }
    if (total == 0) return MA_OK;
    if (by_chunk) {
        MA_REQUIRE(!masked_head, MA_ERR_DEVICE, "internal: chunk form chosen for a masked chunk whose output starts mid-vector");
        // The table is read where it was built (pinned host memory: one wave-uniform 32-byte read per chunk, issued a chunk
        // ahead of its use) unless variant bit 512 asks for the device copy.
        const void* tab = nullptr;
        TableSlotGuard guard(ctx);  // releases the mapped slot behind the launch on every way out
        if (tuning_variant(ctx) & 512) {
            MA_TRY(table_commit(ctx, host, bytes, dev_tab));
            tab = dev_tab;
        } else {
            MA_TRY(table_commit_mapped(ctx, host, &tab, &guard.slot));
        }
        const ChunkPair* dcd = (const ChunkPair*)tab;
        const ChunkMaskDesc* dmd = any_mask ? (const ChunkMaskDesc*)((const char*)tab + off_md) : nullptr;
        const int grid = grid_for(ctx, n, 6);
        bool launched = false;
        if constexpr (std::is_integral<T>::value) {
            if (dv) {
                hipLaunchKernelGGL((chunk_binary_kernel<T, U, true, true>), dim3(grid), dim3(kBlock), 0, ctx->stream, dcd, dmd, (int)n,
                                   op, ctx->dev_flags, smode, sbits);
                launched = true;
            }
        }
        if (launched) {
        } else if (any_mask)
            hipLaunchKernelGGL((chunk_binary_kernel<T, U, true>), dim3(grid), dim3(kBlock), 0, ctx->stream, dcd, dmd, (int)n, op,
                               ctx->dev_flags, smode, sbits);
        else
            hipLaunchKernelGGL((chunk_binary_kernel<T, U, false>), dim3(grid), dim3(kBlock), 0, ctx->stream, dcd, dmd, (int)n, op,
                               ctx->dev_flags, smode, sbits);
        MA_HIP(hipGetLastError());
        return MA_OK;
    }
    MA_TRY(table_commit(ctx, host, bytes, dev_tab));
    const ChunkPair* dcd = (const ChunkPair*)dev_tab;
    const ChunkMaskDesc* dmd = any_mask ? (const ChunkMaskDesc*)(dev_tab + off_md) : nullptr;
    const uint64_t* dt0 = (const uint64_t*)(dev_tab + off_t0);
    const uint64_t* dw0 = any_mask ? (const uint64_t*)(dev_tab + off_w0) : nullptr;
    const bool fuse = any_mask && !masked_head && (dv || !(tuning_variant(ctx) & 64));  // variant bit 64: always the separate bitmap launch
    MA_REQUIRE(!dv || fuse, MA_ERR_DEVICE, "internal: data-dependent validity needs every masked chunk's output on a 16-byte boundary");
    if (any_mask && !fuse) {
        int grid = grid_for(ctx, (n_words + kBlock - 1) / kBlock, 8);
        hipLaunchKernelGGL(batched_mask_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, dcd, dmd, dw0, (int)n, n_words);
        MA_HIP(hipGetLastError());
    }
    const int grid = grid_for(ctx, n_tiles, 6);
    bool launched = false;
    if constexpr (std::is_integral<T>::value) {
        if (dv) {
            hipLaunchKernelGGL((batched_binary_kernel<T, U, true, true>), dim3(grid), dim3(kBlock), 0, ctx->stream, dcd, dmd, dt0, (int)n,
                               n_tiles, op, ctx->dev_flags, smode, sbits);
            launched = true;
        }
    }
    if (launched) {
    } else if (fuse)
        hipLaunchKernelGGL((batched_binary_kernel<T, U, true>), dim3(grid), dim3(kBlock), 0, ctx->stream, dcd, dmd, dt0, (int)n, n_tiles,
                           op, ctx->dev_flags, smode, sbits);
    else
        hipLaunchKernelGGL((batched_binary_kernel<T, U, false>), dim3(grid), dim3(kBlock), 0, ctx->stream, dcd, dmd, dt0, (int)n, n_tiles,
                           op, ctx->dev_flags, smode, sbits);
    MA_HIP(hipGetLastError());
    return MA_OK;
}

template <typename T, int U>
static ma_status batched_impl_u(ma_ctx* ctx, int op, size_t n_chunks, const void* const* lhs_data, const size_t* lens,
                                const uint8_t* const* lhs_masks, const void* const* rhs_data,
                                const uint8_t* const* rhs_masks, const uint8_t* override_mask, void* const* out_data,
                                uint8_t* const* out_masks, int32_t* out_has_mask, bool any_mask, bool by_chunk, int smode,
                                uint64_t sbits) {
    MA_ENTER(ctx);
    MA_NO_CAPTURE(ctx, "route_super_array_broadcast (descriptor upload)");
    MA_HIP(hipSetDevice(ctx->device));
    CallScope scope(ctx);
    PairRoles roles;
    // Segments: a short list is one segment; a long one (RechunkStrategy-sized chunks of a big column) is cut into
    // segments of 4096, 8192, ... 32768 chunks, each with its own table and launch, so that the GPU works on segment k
    // while the host describes segment k + 1 (8 ns per chunk on the host against 17 ns of kernel per 8192-row i32 pair:
    // after the first, small segment the host stays ahead). variant bit 1024: one segment whatever the length (A/B).
    constexpr size_t kPerChunk = sizeof(ChunkPair) + sizeof(ChunkMaskDesc) + 16;
    const size_t per_round = (size_t)grid_for(ctx, (size_t)1 << 30, 6);  // segments are whole rounds of the chunk kernel's grid
    const size_t kFirst = 4 * per_round, kMax = 21 * per_round;
    const bool segmented = n_chunks > 2 * kFirst && !(tuning_variant(ctx) & 1024);
    char* dev_tab = nullptr;
    size_t n_segments = 1;
    if (segmented)
        for (size_t c = 0, sz = kFirst; c + sz < n_chunks; c += sz, sz = sz * 2 < kMax ? sz * 2 : kMax) ++n_segments;
    MA_TRY(ctx_scratch(ctx, kPerChunk * n_chunks + 256 * n_segments, (void**)&dev_tab));
    size_t c0 = 0, seg = segmented ? kFirst : n_chunks, dev_off = 0;
    while (c0 < n_chunks) {
        const size_t c1 = c0 + seg < n_chunks ? c0 + seg : n_chunks;
        MA_TRY((batched_segment<T, U>(ctx, scope, roles, op, c0, c1, lhs_data, lens, lhs_masks, rhs_data, rhs_masks, override_mask,
                                      out_data, out_masks, out_has_mask, any_mask, by_chunk, dev_tab + dev_off, smode, sbits)));
        dev_off += (kPerChunk * (c1 - c0) + 255) & ~(size_t)255;
        c0 = c1;
        if (seg < kMax) seg = seg * 2 < kMax ? seg * 2 : kMax;
    }
    return finish_batched(ctx, scope, std::is_integral<T>::value &&
                                          (op == MA_OP_DIVIDE || op == MA_OP_REMAINDER || op == MA_OP_FLOORDIV));
}

// Picks the form (chunk-per-workgroup kernel on a pinned-host table, or tile-search kernel on an uploaded table) and the
// rows per wave step: 8 x 16 bytes per lane (the single-array kernels' shape) or 4 x 16 bytes (fewer ragged tiles when
// chunks are short). variant bit 16 forces 4, bit 32 forces 8 (tuning).
template <typename T>
static ma_status batched_impl(ma_ctx* ctx, int op, size_t n_chunks, const void* const* lhs_data, const size_t* lens,
                              const uint8_t* const* lhs_masks, const void* const* rhs_data,
                              const uint8_t* const* rhs_masks, const uint8_t* override_mask, void* const* out_data,
                              uint8_t* const* out_masks, int32_t* out_has_mask, int smode, uint64_t sbits) {
    size_t total = 0;
    bool any_mask = false, masked_head = false;
    for (size_t i = 0; i < n_chunks; ++i) {
        total += lens[i];
        if (!lens[i]) continue;
        const bool m = override_mask || (lhs_masks && lhs_masks[i]) || (rhs_masks && rhs_masks[i]);
        any_mask |= m;
        // a staged (pageable) output lands on a 256-byte boundary: looking at the caller's pointer only over-estimates
        masked_head |= m && (((uintptr_t)out_data[i]) & 15) != 0;
    }
    if (total == 0) {
        if (out_has_mask)
            for (size_t i = 0; i < n_chunks; ++i)
                out_has_mask[i] = (override_mask || (lhs_masks && lhs_masks[i]) || (rhs_masks && rhs_masks[i])) ? 1 : 0;
        return MA_OK;
    }
    constexpr size_t kWideTileRows = (size_t)64 * (16 / sizeof(T)) * 8 * kWaves;
    const size_t avg = total / n_chunks;
    size_t longest = 0;
    for (size_t i = 0; i < n_chunks; ++i)
        if (lens[i] > longest) longest = lens[i];
    // Which form (profiles/r03_matrix_super_array.jsonl, 60 000 x 8192-row pairs, fraction of the same-process copy / one
    // call from an idle stream to complete results against the copy-equivalent time):
    //   chunk-per-workgroup kernel on the pinned-host table   i32 1.03 dense, 1.01 masked / 1.06-1.12 x;  f64 0.97, 0.98 / 1.05-1.11 x
    //   tile-search kernel on the uploaded table                i32 0.93, 0.94 / 1.18-1.20 x;               f64 1.00, 0.91 / 1.04-1.16 x
    // Many chunks of a few tiles each take the chunk form: enough chunks to keep every workgroup busy with an even share
    // (>= 4 per CU), none so long that its workgroup becomes the tail, every masked chunk's output on a 16-byte boundary
    // (runs must start on validity words). Everything else — a few long chunks above all — is dealt out tile by tile.
    // variant bit 128 forces the tile form, bit 256 the chunk form (tuning / tests).
    bool by_chunk = n_chunks >= (size_t)4 * (size_t)ctx->num_cus && avg <= ((size_t)1 << 16) && longest <= 8 * (avg ? avg : 1);
    if (form_variant(ctx) & 128) by_chunk = false;
    if (form_variant(ctx) & 256) by_chunk = true;
    by_chunk = by_chunk && !masked_head && n_chunks < ((size_t)1 << 31);
    // 8 x 16 bytes per lane when the tiles that gives are filled: 8-byte types in the chunk form (two 4096-row tiles per
    // 8192-row chunk; 4-byte types are faster with two 4 x 16-byte tiles), long chunks in the tile form.
    bool wide = by_chunk ? (sizeof(T) >= 8 && avg >= kWideTileRows) : avg >= 16 * kWideTileRows;
    if (form_variant(ctx) & 16) wide = false;
    if (form_variant(ctx) & 32) wide = true;
    if (wide)
        return batched_impl_u<T, 8>(ctx, op, n_chunks, lhs_data, lens, lhs_masks, rhs_data, rhs_masks, override_mask, out_data, out_masks, out_has_mask, any_mask, by_chunk, smode, sbits);
    return batched_impl_u<T, 4>(ctx, op, n_chunks, lhs_data, lens, lhs_masks, rhs_data, rhs_masks, override_mask, out_data, out_masks, out_has_mask, any_mask, by_chunk, smode, sbits);
}

}  // namespace ma

namespace ma {

// Batched form of route_super_array_broadcast for same-type chunk pairs (internal: declared in ma_common.hpp). The
// caller (ma_route_super_array_broadcast in ma_arrow.hip) decides which calls need the per-chunk path instead.
ma_status route_batched(ma_ctx* ctx, int32_t format_code, int32_t op, size_t n_chunks, const void* const* lhs_data,
                        const size_t* lens, const uint8_t* const* lhs_masks, const void* const* rhs_data,
                        const uint8_t* const* rhs_masks, const uint8_t* override_mask, void* const* out_data,
                        uint8_t* const* out_masks, int32_t* out_has_mask, int smode, uint64_t sbits) {
    switch (format_code) {
        case 'i': return batched_impl<int32_t>(ctx, op, n_chunks, lhs_data, lens, lhs_masks, rhs_data, rhs_masks, override_mask, out_data, out_masks, out_has_mask, smode, sbits);
        case 'I': return batched_impl<uint32_t>(ctx, op, n_chunks, lhs_data, lens, lhs_masks, rhs_data, rhs_masks, override_mask, out_data, out_masks, out_has_mask, smode, sbits);
        case 'l': return batched_impl<int64_t>(ctx, op, n_chunks, lhs_data, lens, lhs_masks, rhs_data, rhs_masks, override_mask, out_data, out_masks, out_has_mask, smode, sbits);
        case 'L': return batched_impl<uint64_t>(ctx, op, n_chunks, lhs_data, lens, lhs_masks, rhs_data, rhs_masks, override_mask, out_data, out_masks, out_has_mask, smode, sbits);
        case 'f': return batched_impl<float>(ctx, op, n_chunks, lhs_data, lens, lhs_masks, rhs_data, rhs_masks, override_mask, out_data, out_masks, out_has_mask, smode, sbits);
        case 'g': return batched_impl<double>(ctx, op, n_chunks, lhs_data, lens, lhs_masks, rhs_data, rhs_masks, override_mask, out_data, out_masks, out_has_mask, smode, sbits);
        default:
            set_error("unsupported element format '%c'", (char)format_code);
            return MA_ERR_UNSUPPORTED;
    }
}

}  // namespace ma
