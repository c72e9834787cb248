// SuperTable / SuperArray consolidation for gfx950: K chunks of one column -> one contiguous column (+ validity).
//
// Replaces, for numeric columns:
//   Consolidate::consolidate -> consolidate_concat / consolidate_arena   src/structs/chunked/super_table.rs:657-743
//   consolidate_{int,float}_variant!, extend_null_mask                   src/traits/consolidate.rs:80-207
//   Arena::write_slices (memcpy per chunk + mask build)                  src/structs/arena.rs:264-308
//   append_array (extend_from_slice + bit-by-bit mask extend)            src/macros.rs:311-345
//
// Semantics restated: values are concatenated in chunk order; the result has a validity bitmap iff at least one
// chunk has one, and a chunk without a bitmap contributes all-valid rows (consolidate.rs:80-105).
//
// One launch copies every chunk (a descriptor table with per-chunk tile prefix sums is binary-searched once per
// workgroup tile), so 100 x 10 000-row batches (benches/consolidate.rs:21-58) cost one launch, not 100 memcpys; a second
// launch assembles the output bitmap word by word from bit-granular pieces of the chunk bitmaps. HBM-bound:
// 2 x elem_size bytes per row (+ 2/8 for validity).
#include <vector>

#include "ma_device.hpp"

namespace ma {

struct ChunkDesc {
    const void* data;        // first element of the chunk window
    size_t start;            // first output row of this chunk
    size_t len;              // rows
    const uint64_t* words;   // validity words (8-byte aligned base) or nullptr = all valid
    size_t bit_off;          // bit index of the chunk's row 0 relative to `words`
    size_t last_word;        // last word index holding a window bit
    size_t tile0;            // index of this chunk's first copy tile (prefix sum over chunks)
    unsigned head;           // rows in front of the first 16-byte boundary of the DESTINATION
};

__device__ __forceinline__ int find_chunk_by_tile(const ChunkDesc* __restrict__ c, int n_chunks, size_t tile) {
    int lo = 0, hi = n_chunks - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (c[mid].tile0 <= tile) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

// Copy tiles: every chunk is cut into tiles of TILE_ROWS rows counted from its first 16-byte aligned destination
// row (tile 0 also takes the `head` rows in front of it). A full tile moves 16 bytes per lane per access (the
// bandwidth path) whatever the source's byte phase; partial tiles move whole vectors, then single rows.
template <typename T, int UNROLL>
__device__ __forceinline__ void concat_tile(const T* __restrict__ src, T* __restrict__ dst, unsigned head, size_t len,
                                            size_t lt, unsigned lane, unsigned wave) {
    typedef MaU4 V;  // a copy moves dwords whatever the element type (and 1-byte vectors lose the loads' nt hint: ma_device.hpp)
    constexpr int R = 16 / (int)sizeof(T);
    constexpr size_t WAVE_ROWS = (size_t)64 * R * UNROLL;
    constexpr size_t TILE_ROWS = WAVE_ROWS * kWaves;
    const size_t r0 = head + lt * TILE_ROWS;  // first row of the aligned part of this tile
    const size_t r1 = r0 + TILE_ROWS < len ? r0 + TILE_ROWS : len;
    if (lt == 0) {
        for (size_t i = threadIdx.x; i < head && i < len; i += kBlock) as_global(dst)[i] = as_global(src)[i];
    }
    if (r0 >= len) return;
    if (r1 - r0 == TILE_ROWS) {
        // Stores are 16-byte aligned by construction of the tiles; the source is read at whatever element-aligned
        // phase it has with the same global_load_dwordx4 (gfx950 runs with unaligned access mode on under HSA; a
        // misaligned wave access touches one extra cache line per KiB).
        const size_t w0 = r0 + (size_t)wave * WAVE_ROWS;
        const V* __restrict__ p = (const V*)(src + w0) + lane;
        V* __restrict__ q = (V*)(dst + w0) + lane;
        V v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = load16u<V, true>(p + (size_t)u * 64);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) store16<V, true>(q + (size_t)u * 64, v[u]);
    } else {
        // The chunk's last, partial tile (up to TILE_ROWS - 1 rows — 32 767 for 1-byte columns): whole 16-byte
        // vectors first (dst + r0 is 16-byte aligned like every tile start), then the few rows left, instead of one
        // element per lane per trip (128 trips of 1-byte accesses for one workgroup, ~10 % of a 1-byte consolidate).
        const size_t n_vec = (r1 - r0) / R;
        const V* __restrict__ p = (const V*)(src + r0);
        V* __restrict__ q = (V*)(dst + r0);
        for (size_t v = threadIdx.x; v < n_vec; v += kBlock) store16<V, true>(q + v, load16u<V, true>(p + v));
        for (size_t i = r0 + n_vec * R + threadIdx.x; i < r1; i += kBlock) as_global(dst)[i] = as_global(src)[i];
    }
}

template <typename T, int UNROLL>
__global__ __launch_bounds__(kBlock) void concat_kernel(const ChunkDesc* __restrict__ chunks, int n_chunks,
                                                        size_t n_tiles, T* __restrict__ out) {
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the previous tile's chunk is remembered (long chunks: the next tile of this workgroup is nearly always in it)
    size_t t_lo = 1, t_hi = 0;
    ChunkDesc d{};
    for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        if (t < t_lo || t >= t_hi) {
            const int c = find_chunk_by_tile(chunks, n_chunks, t);  // workgroup-uniform
            d = chunks[c];
            t_lo = d.tile0;
            t_hi = c + 1 < n_chunks ? chunks[c + 1].tile0 : n_tiles;
        }
        concat_tile<T, UNROLL>((const T*)d.data, out + d.start, d.head, d.len, t - t_lo, lane, wave);
    }
}

// RechunkStrategy-sized chunk lists (src/structs/chunked/super_array.rs:51-59: 8192 rows by default — a 10^9-row column is
// 122 000 chunks): a whole chunk per workgroup, no search, and the 32-byte descriptors are read where the host built
// them — in the pinned staging buffer, with wave-uniform loads issued a chunk ahead of their use — instead of crossing
// PCIe as a 64-byte-per-chunk table ON the stream in front of the kernel (ma_superarray.hip has the measurements that
// led here: the copy and the host's table building, not the search, are what the chunked regime pays for).
struct ConcatChunk {  // 32 bytes
    const void* data;
    uint64_t start;       // first output row
    uint64_t len_bit;     // rows (low 58 bits) | bit index of row 0 inside words[0] (high 6 bits)
    const uint64_t* words;  // the word of the chunk's bitmap that holds its row 0, or nullptr = all rows valid / no validity
};
constexpr uint64_t kConcatLenMask = (((uint64_t)1) << 58) - 1;
// The validity fields ride in the SAME 32-byte entry: a second table (16 bytes per chunk) doubled the number of reads
// that cross PCIe per chunk, and at 8192-row chunks of a 4-byte column — 18 us of rows per chunk per workgroup, 1536
// workgroups — the request rate, not the bytes, became the limit (kernel trace: +14 % with the second table, +1.5 % on an
// 8-byte column whose chunks take twice as long).

// MASK: the workgroup that copies chunk c also writes the output validity words that lie wholly inside the chunk's rows
// (for 8192-row chunks on 64-row boundaries: all 128 of them, by one pass of two waves) — one funnel shift of an unaligned
// 8-byte load each, no search. Words that hold a join of two chunks (a chunk starting in the middle of a word) are left to
// concat_mask_joins_kernel; the column's last, partial word belongs to the chunk it begins in.
template <typename T, int UNROLL, bool MASK>
__global__ __launch_bounds__(kBlock) void concat_chunk_kernel(const ConcatChunk* __restrict__ cd, int n_chunks,
                                                              T* __restrict__ out, uint64_t* __restrict__ out_words,
                                                              size_t total) {
    constexpr int R = 16 / (int)sizeof(T);
    constexpr size_t TILE_ROWS = (size_t)64 * R * UNROLL * kWaves;
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int c = blockIdx.x;
    if (c >= n_chunks) return;
    ConcatChunk e = cd[c];
    while (true) {
        const int next = c + (int)gridDim.x;
        ConcatChunk en{};
        if (next < n_chunks) en = cd[next];  // wave-uniform address: a scalar load, in flight while this chunk streams
        const size_t len = (size_t)(e.len_bit & kConcatLenMask);
        struct {
            const uint64_t* words;
            size_t bit_off;
        } m{e.words, (size_t)(e.len_bit >> 58)};
        if (len) {
            T* dst = out + e.start;
            const unsigned mis = (unsigned)((uintptr_t)dst & 15);
            const unsigned head = mis ? (16 - mis) / (unsigned)sizeof(T) : 0;
            const size_t n_t = len > head ? (len - head + TILE_ROWS - 1) / TILE_ROWS : 1;
            // validity words of this chunk: thread t takes word w0 + t (+ kBlock, ...). The first word's loads are issued
            // BEFORE the data tiles and consumed after them, so that their latency hides behind the chunk's rows instead of
            // following them (measured: +11 % on a 4-byte column of 8192-row chunks when the words were loaded afterwards)
            const size_t start = (size_t)e.start, end = start + len;
            const size_t w0 = (start + 63) >> 6, w1 = end >> 6;
            const bool tail = end == total && (end & 63) != 0 && w0 <= w1;  // the partial last word begins in this chunk
            const size_t n_w = MASK ? (w1 > w0 ? w1 - w0 : 0) + (tail ? 1 : 0) : 0;
            const size_t last_word = ((size_t)m.bit_off + len - 1) >> 6;
            auto fetch = [&](size_t t, uint64_t& lo, uint64_t& hi, unsigned& sh) {
                const size_t row = (w0 + t) << 6;
                const size_t b = (size_t)m.bit_off + (row - start);
                if (end - row >= 64) {
                    // bits b .. b + 63 are window bits: bytes b/8 .. b/8 + 7, and (b % 8 != 0) byte b/8 + 8, hold them
                    typedef uint64_t u64u __attribute__((aligned(1)));
                    typedef const u64u __attribute__((address_space(1)))* GP;  // global, not flat (ma_device.hpp: as_global)
                    const auto base = as_global((const uint8_t*)m.words + (b >> 3));
                    sh = (unsigned)(b & 7);
                    lo = *(GP)base;
                    hi = sh ? (uint64_t)base[8] : 0;
                } else {
                    const size_t wi = b >> 6;
                    sh = (unsigned)(b & 63);
                    const auto gw = as_global(m.words);
                    lo = gw[wi];
                    hi = (sh && wi + 1 <= last_word) ? gw[wi + 1] : 0;
                }
            };
            uint64_t lo0 = 0, hi0 = 0;
            unsigned sh0 = 0;
            const bool mine = MASK && threadIdx.x < n_w && m.words != nullptr;
            if (mine) fetch(threadIdx.x, lo0, hi0, sh0);
            for (size_t lt = 0; lt < n_t; ++lt) concat_tile<T, UNROLL>((const T*)e.data, dst, head, len, lt, lane, wave);
            if constexpr (MASK) {
                for (size_t t = threadIdx.x; t < n_w; t += kBlock) {
                    const size_t w = w0 + t, avail = end - (w << 6);
                    uint64_t v = ~(uint64_t)0;  // a chunk without a bitmap is all valid (consolidate.rs:91-96)
                    if (m.words != nullptr) {
                        uint64_t lo = lo0, hi = hi0;
                        unsigned sh = sh0;
                        if (t != threadIdx.x) fetch(t, lo, hi, sh);
                        v = sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
                    }
                    if (avail < 64) v &= (((uint64_t)1) << avail) - 1;  // bits >= total stay zero
                    __builtin_nontemporal_store(v, out_words + w);
                }
            }
        }
        if (next >= n_chunks) break;
        c = next;
        e = en;
    }
}

// The part of a chunk descriptor the validity concat needs (40 bytes; up to kLdsChunks of them are staged in LDS so
// that the per-pair binary search never leaves the CU).
struct MaskDesc {
    size_t start, len;
    const uint64_t* words;
    size_t bit_off, last_word;
};
constexpr int kLdsChunks = 256;

template <bool LDS>
struct DescTable {
    const ChunkDesc* g;   // the copy kernel's table (64-byte entries), or nullptr when only ...
    const MaskDesc* s;    // ... a compact MaskDesc table exists (LDS: staged in shared memory)
    __device__ __forceinline__ MaskDesc get(int i) const {
        if (LDS || g == nullptr) return s[i];
        const ChunkDesc& c = g[i];
        return MaskDesc{c.start, c.len, c.words, c.bit_off, c.last_word};
    }
    __device__ __forceinline__ size_t start(int i) const {
        if (LDS || g == nullptr) return s[i].start;
        return g[i].start;
    }
    // Index of the chunk that contains output row `row` (row < total).
    __device__ __forceinline__ int find(int n_chunks, size_t row) const {
        int lo = 0, hi = n_chunks - 1;
        while (lo < hi) {
            int mid = (lo + hi + 1) >> 1;
            if (start(mid) <= row) lo = mid;
            else hi = mid - 1;
        }
        return lo;
    }
};

// One output validity word starting at output row `row` (a multiple of 64): pieces of up to 64 bits are pulled from
// the chunk bitmaps, starting the search at chunk `c`.
template <bool LDS>
__device__ __forceinline__ uint64_t gather_word(const DescTable<LDS>& tab, int c, size_t row, size_t total) {
    const size_t row_end = row + 64 < total ? row + 64 : total;
    uint64_t word = 0;
    unsigned filled = 0;
    MaskDesc d = tab.get(c);
    while (row < row_end) {
        while (row >= d.start + d.len) d = tab.get(++c);
        const size_t in_chunk = row - d.start;
        size_t take = d.len - in_chunk;
        if (take > row_end - row) take = row_end - row;
        uint64_t piece;
        if (d.words == nullptr) {
            piece = ~(uint64_t)0;  // a chunk without a bitmap is all valid (consolidate.rs:91-96)
        } else {
            const size_t b = d.bit_off + in_chunk;
            const size_t w = b >> 6;
            const unsigned sh = (unsigned)(b & 63);
            piece = d.words[w] >> sh;
            if (sh && (w + 1) <= d.last_word) piece |= d.words[w + 1] << (64 - sh);
        }
        if (take < 64) piece &= (((uint64_t)1) << take) - 1;
        word |= piece << filled;
        filled += (unsigned)take;
        row += take;
    }
    return word;  // bits >= total stay zero
}

// One thread per PAIR of output words (a 16-byte store). When the pair's 128 rows lie inside one chunk — all but
// the few pairs at chunk joins — the two words are funnel-shifted out of three consecutive source words.
template <bool LDS>
__global__ __launch_bounds__(kBlock) void concat_mask_kernel(const ChunkDesc* __restrict__ chunks,
                                                             const MaskDesc* __restrict__ compact, int n_chunks,
                                                             size_t total, uint64_t* __restrict__ out_words) {
    typedef unsigned long long u2 __attribute__((ext_vector_type(2)));
    __shared__ MaskDesc staged[LDS ? kLdsChunks : 1];
    if constexpr (LDS) {
        for (int i = threadIdx.x; i < n_chunks; i += kBlock) {
            if (chunks != nullptr) {
                const ChunkDesc& c = chunks[i];
                staged[i] = MaskDesc{c.start, c.len, c.words, c.bit_off, c.last_word};
            } else {
                staged[i] = compact[i];
            }
        }
        __syncthreads();
    }
    const DescTable<LDS> tab{chunks, LDS ? staged : compact};
    const size_t n_words = (total + 63) >> 6;
    const size_t n_pairs = (n_words + 1) >> 1;
    const bool out16 = ((uintptr_t)out_words & 15) == 0;

    // one pair of output words whatever it straddles: the general form
    auto one_pair = [&](size_t p) {
        const size_t row = p << 7;
        const int c = tab.find(n_chunks, row);
        const MaskDesc d = tab.get(c);
        uint64_t w0, w1;
        if (row + 128 <= d.start + d.len) {
            if (d.words == nullptr) {
                w0 = w1 = ~(uint64_t)0;
            } else {
                const size_t b = d.bit_off + (row - d.start);
                const size_t byte0 = b >> 3;
                const unsigned sub = (unsigned)(b & 7);
                if (byte0 + 17 <= (d.last_word + 1) * 8) {
                    // one byte-aligned 16-byte load + the one byte that holds the last `sub` bits (two accesses, the
                    // wide one at the full 16-byte rate) instead of three 8-byte loads
                    typedef u2 u2u __attribute__((aligned(1)));
                    const uint8_t* base = (const uint8_t*)d.words + byte0;
                    const u2 pr = *(const u2u*)base;
                    const uint64_t nx = sub ? (uint64_t)base[16] : 0;
                    w0 = sub ? (pr.x >> sub) | (pr.y << (64 - sub)) : pr.x;
                    w1 = sub ? (pr.y >> sub) | (nx << (64 - sub)) : pr.y;
                } else {
                    const size_t w = b >> 6;
                    const unsigned sh = (unsigned)(b & 63);
                    const uint64_t s0 = d.words[w];
                    const uint64_t s1 = d.words[w + 1];  // row + 64 is inside the chunk: word w + 1 holds window bits
                    const uint64_t s2 = (sh && w + 2 <= d.last_word) ? d.words[w + 2] : 0;
                    w0 = sh ? (s0 >> sh) | (s1 << (64 - sh)) : s0;
                    w1 = sh ? (s1 >> sh) | (s2 << (64 - sh)) : s1;
                }
            }
        } else {
            w0 = gather_word<LDS>(tab, c, row, total);
            w1 = row + 64 < total ? gather_word<LDS>(tab, c, row + 64, total) : 0;
        }
        if (2 * p + 1 < n_words) {
            if (out16) {
                u2 v = {w0, w1};
                *(u2*)(out_words + 2 * p) = v;
            } else {
                out_words[2 * p] = w0;
                out_words[2 * p + 1] = w1;
            }
        } else {
            out_words[2 * p] = w0;
        }
    };

    // A wave takes runs of 64 x kRun consecutive pairs (4 KiB of output). Almost every run lies inside ONE chunk: the chunk
    // is then looked up once for the wave (not once per 16 bytes — a binary search and a 40-byte descriptor read, both in
    // LDS, per pair were what kept a bit-granular join at 0.90 of the copy rate), the bit phase is the same for every lane
    // (pairs are 128 rows apart), and the kRun loads of a lane are in flight together.
    constexpr int kRun = 4;
    const unsigned lane = threadIdx.x & 63;
    const size_t wave_id = ((size_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * kBlock) >> 6;
    const size_t n_runs = (n_pairs + 64 * kRun - 1) / (64 * kRun);
    for (size_t run = wave_id; run < n_runs; run += n_waves) {
        const size_t p0 = run * 64 * kRun;
        const size_t row0 = p0 << 7;
        const size_t row_end = (p0 + 64 * kRun) << 7;  // one past the run's last row, if the run is whole
        const int c = __builtin_amdgcn_readfirstlane(tab.find(n_chunks, row0));
        const MaskDesc d = tab.get(c);
        const size_t b0 = d.bit_off + (row0 - d.start);
        const bool whole = p0 + 64 * kRun <= n_pairs && 2 * (p0 + 64 * kRun) <= n_words && row_end <= d.start + d.len &&
                           out16 && (d.words == nullptr || ((b0 + ((size_t)(64 * kRun - 1) << 7)) >> 3) + 17 <= (d.last_word + 1) * 8);
        if (whole) {  // wave-uniform
            u2 v[kRun];
            if (d.words == nullptr) {
#pragma unroll
                for (int u = 0; u < kRun; ++u) v[u] = u2{~(uint64_t)0, ~(uint64_t)0};
            } else {
                typedef u2 u2u __attribute__((aligned(1)));
                const unsigned sub = (unsigned)(b0 & 7);  // the same for every pair of the run
                u2 pr[kRun];
                uint64_t nx[kRun];
                uint64_t last_byte = 0;  // lane 63 of the last load needs the one byte behind the run
#pragma unroll
                for (int u = 0; u < kRun; ++u) {
                    const uint8_t* base = (const uint8_t*)d.words + ((b0 + ((size_t)(u * 64 + lane) << 7)) >> 3);
                    pr[u] = __builtin_nontemporal_load((const u2u __attribute__((address_space(1)))*)base);  // a stream: read once
                    if (u == kRun - 1 && sub && lane == 63) last_byte = base[16];
                }
                // the byte that holds a pair's last `sub` bits is the first byte of the NEXT pair's load: one lane up, or
                // lane 0 of the next load (a byte load per pair doubled the number of memory instructions)
#pragma unroll
                for (int u = 0; u < kRun; ++u) {
                    const unsigned mine = (unsigned)pr[u].x & 0xFFu;
                    unsigned up = (unsigned)__shfl_down((int)mine, 1, 64);
                    const unsigned wrap = u + 1 < kRun ? (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)pr[u + 1 < kRun ? u + 1 : u].x & 0xFFu))
                                                       : (unsigned)last_byte;
                    nx[u] = lane == 63 ? wrap : up;
                }
#pragma unroll
                for (int u = 0; u < kRun; ++u) {
                    v[u].x = sub ? (pr[u].x >> sub) | (pr[u].y << (64 - sub)) : pr[u].x;
                    v[u].y = sub ? (pr[u].y >> sub) | (nx[u] << (64 - sub)) : pr[u].y;
                }
            }
#pragma unroll
            for (int u = 0; u < kRun; ++u) __builtin_nontemporal_store(v[u], (u2 __attribute__((address_space(1)))*)(out_words + 2 * (p0 + u * 64 + lane)));
        } else {
            for (int u = 0; u < kRun; ++u) {
                const size_t p = p0 + u * 64 + lane;
                if (p < n_pairs) one_pair(p);
            }
        }
    }
}

static void launch_concat_mask(ma_ctx* ctx, const ChunkDesc* d, size_t n_chunks, size_t total, uint64_t* ow,
                               const MaskDesc* compact = nullptr) {
    const size_t n_pairs = (((total + 63) >> 6) + 1) >> 1;
    const int grid = grid_for(ctx, (n_pairs + kBlock * 4 - 1) / (kBlock * 4), 8);  // a wave takes runs of 64 x 4 pairs
    if (n_chunks <= (size_t)kLdsChunks)
        hipLaunchKernelGGL(concat_mask_kernel<true>, dim3(grid), dim3(kBlock), 0, ctx->stream, d, compact, (int)n_chunks, total, ow);
    else
        hipLaunchKernelGGL(concat_mask_kernel<false>, dim3(grid), dim3(kBlock), 0, ctx->stream, d, compact, (int)n_chunks, total, ow);
}

// The words concat_chunk_kernel<MASK> leaves out: those a chunk STARTS in the middle of. One thread per chunk; the first
// non-empty chunk that starts inside a word owns it and gathers it from the chunks either side (gather_word). Launched
// only when the host saw such a start (chunk lengths that are not multiples of 64 rows).
__global__ __launch_bounds__(kBlock) void concat_mask_joins_kernel(const MaskDesc* __restrict__ compact, int n_chunks,
                                                                   size_t total, uint64_t* __restrict__ out_words) {
    const DescTable<false> tab{nullptr, compact};
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < (size_t)n_chunks; i += stride) {
        const int c = (int)i;
        const size_t start = compact[c].start;
        if (compact[c].len == 0 || (start & 63) == 0) continue;
        const size_t row = start & ~(size_t)63;
        int first = c;  // the chunk that holds the word's first row
        bool owner = true;
        for (int k = c - 1; k >= 0; --k) {
            if (compact[k].len == 0) continue;
            if (compact[k].start > row) owner = false;  // an earlier chunk starts inside this word too: it (or one before it) owns
            else first = k;
            break;
        }
        if (owner) out_words[row >> 6] = gather_word<false>(tab, first, row, total);
    }
}

// Bit-packed columns cut into many short chunks (a Boolean column rechunked at 8192 rows: 122 000 chunks per 10^9 rows): a
// WAVE writes the output words that lie wholly inside its chunk — no search at all —, concat_mask_joins_kernel the words a
// chunk starts inside. concat_mask_kernel's run-wise lookup pays a 17-step search of the chunk table per 4 KiB there
// (131 072 x 8192-bit chunks: 306 us of kernel for 256 MB of traffic).
__global__ __launch_bounds__(kBlock) void concat_bits_chunk_kernel(const MaskDesc* __restrict__ compact, int n_chunks, size_t total,
                                                                   uint64_t* __restrict__ out_words) {
    const unsigned lane = threadIdx.x & 63;
    const size_t wave_id = ((size_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * kBlock) >> 6;
    for (size_t ci = wave_id; ci < (size_t)n_chunks; ci += n_waves) {
        const MaskDesc d = compact[__builtin_amdgcn_readfirstlane((int)ci)];
        if (d.len == 0) continue;
        const size_t start = d.start, end = start + d.len;
        const size_t w0 = (start + 63) >> 6, w1 = end >> 6;
        const bool tail = end == total && (end & 63) != 0 && w0 <= w1;  // the partial last word begins in this chunk
        const size_t n_w = (w1 > w0 ? w1 - w0 : 0) + (tail ? 1 : 0);
        const auto gw = as_global(d.words);
        for (size_t t = lane; t < n_w; t += 64) {
            const size_t w = w0 + t, row = w << 6, avail = end - row;
            uint64_t v = ~(uint64_t)0;  // a chunk without a bitmap is all valid (consolidate.rs:91-96)
            if (d.words != nullptr) {
                const size_t b = d.bit_off + (row - start);
                uint64_t lo, hi;
                unsigned sh;
                if (avail >= 64) {  // bits b .. b + 63 are window bits: bytes b/8 .. b/8 + 7 and, off a byte boundary, b/8 + 8
                    typedef uint64_t u64u __attribute__((aligned(1)));
                    typedef const u64u __attribute__((address_space(1)))* GP;
                    const auto base = as_global((const uint8_t*)d.words + (b >> 3));
                    sh = (unsigned)(b & 7);
                    lo = *(GP)base;
                    hi = sh ? (uint64_t)base[8] : 0;
                } else {
                    const size_t wi = b >> 6;
                    sh = (unsigned)(b & 63);
                    lo = gw[wi];
                    hi = (sh && wi + 1 <= d.last_word) ? gw[wi + 1] : 0;
                }
                v = sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
            }
            if (avail < 64) v &= (((uint64_t)1) << avail) - 1;  // bits >= total stay zero
            __builtin_nontemporal_store(v, out_words + w);
        }
    }
}

// Picks the form for a compact (device-resident) table: many short chunks -> chunk-owned words + the join pass (only when a
// chunk starts inside a word: `has_join`, decided by the host); otherwise concat_mask_kernel.
static void launch_concat_bits(ma_ctx* ctx, const MaskDesc* compact, size_t n_chunks, size_t total, uint64_t* ow, bool has_join) {
    const bool many_short = n_chunks >= 1024 && total / n_chunks <= ((size_t)1 << 20) && n_chunks < ((size_t)1 << 31);
    if (!many_short || (form_variant(ctx) & 128)) {  // variant bit 128: the searching form (A/B, tests)
        launch_concat_mask(ctx, nullptr, n_chunks, total, ow, compact);
        return;
    }
    const int grid = grid_for(ctx, (n_chunks + kWaves - 1) / kWaves, 8);
    hipLaunchKernelGGL(concat_bits_chunk_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, compact, (int)n_chunks, total, ow);
    if (has_join) {
        const int gj = grid_for(ctx, (n_chunks + kBlock - 1) / kBlock, 8);
        hipLaunchKernelGGL(concat_mask_joins_kernel, dim3(gj), dim3(kBlock), 0, ctx->stream, compact, (int)n_chunks, total, ow);
    }
}

template <typename T>
static void launch_concat(ma_ctx* ctx, const ChunkDesc* d, int n_chunks, size_t n_tiles, void* out) {
    int grid = grid_for(ctx, n_tiles, 6);  // store stream in the mix: more workgroups (profiles/r01_sweep_grid.json)
    hipLaunchKernelGGL((concat_kernel<T, 8>), dim3(grid), dim3(kBlock), 0, ctx->stream, d, n_chunks, n_tiles, (T*)out);
}

template <typename T>
static size_t tile_rows_of() {
    return (size_t)64 * (16 / sizeof(T)) * 8 * kWaves;
}

}  // namespace ma

using namespace ma;

// The chunk-per-workgroup form of one column's concatenation (see ma_consolidate_column for when it is chosen): descriptors in
// the pinned staging buffer, segments of growing size, validity words by the chunk's own workgroup + the join pass.
// `po` / `ow`: device-reachable destination of the values / of the validity words (nullptr: the column has none).
static ma_status concat_column_by_chunks(ma_ctx* ctx, CallScope& scope, size_t elem_size, size_t n_chunks,
                                         const void* const* chunk_data, const size_t* chunk_lens,
                                         const uint8_t* const* chunk_masks, const size_t* chunk_mask_offsets, void* po,
                                         uint64_t* ow, size_t total) {
    const bool has_mask = ow != nullptr;
    // validity: the words inside a chunk are written by the chunk's workgroup; only when some chunk starts in the
    // middle of a word (lengths that are not multiples of 64) is there a join pass, and only then a whole-list table
    bool has_join = false;
    if (has_mask) {
        size_t r = 0;
        for (size_t i = 0; i < n_chunks && !has_join; ++i) {
            has_join = chunk_lens[i] != 0 && (r & 63) != 0;
            r += chunk_lens[i];
        }
    }
    std::vector<MaskDesc> mdesc;
    if (has_join) mdesc.resize(n_chunks);
    DeviceRange data_role, mask_role;
    // segments are whole multiples of the grid (workgroups deal chunks round-robin: 4096 chunks over 1536 workgroups left a
    // third of them a chunk short for a third of the segment). As FEW segments as keep the GPU fed: on output blocks that
    // write fast, a copy kernel of under ~20 rounds runs 5-8 % below the rate the same rows reach inside one long sweep (the
    // plain copy does too: profiles/r04_copy_windows.jsonl), so 4, 8, 16, 21, 21, 9 rounds (122 000 8-byte chunks; rounds
    // 1-3's shape) cost 6.5 % where 4, 16, 59 cost 2 %. The host describes a round in ~6 us whatever the element width; the
    // GPU copies one in 34 us (8192-row chunks of 8 bytes) or 17 us (4 bytes): the next segment may be 4x / 2x the one the
    // GPU is working on without starving it.
    const size_t per_round = (size_t)grid_for(ctx, (size_t)1 << 30, 6);
    const size_t avg_chunk_bytes = total / n_chunks * elem_size;
    const size_t growth = avg_chunk_bytes >= ((size_t)48 << 10) ? 4 : 2;
    const size_t kFirst = 4 * per_round, kMax = 64 * per_round;
    size_t c0 = 0, seg = (n_chunks > 2 * kFirst && !(tuning_variant(ctx) & 1024)) ? kFirst : n_chunks, row = 0;
    while (c0 < n_chunks) {
        const size_t c1 = c0 + seg < n_chunks ? c0 + seg : n_chunks;
        ConcatChunk* cd = nullptr;
        MA_TRY(table_begin(ctx, sizeof(ConcatChunk) * (c1 - c0), (void**)&cd));
        for (size_t i = c0; i < c1; ++i) {
            const void* p = chunk_data[i];
            if (!data_role.holds(p)) {
                MA_TRY(scope.in(chunk_data[i], chunk_lens[i] * elem_size, &p));
                if (chunk_lens[i]) data_role.learn(chunk_data[i]);
            }
            MA_REQUIRE(chunk_lens[i] <= kConcatLenMask, MA_ERR_INVALID_ARGUMENT, "chunk %zu is too long", i);
            const uint64_t* words = nullptr;
            size_t bit_off = 0;
            if (has_mask) {
                if (chunk_masks && chunk_masks[i] && chunk_lens[i]) {
                    const size_t mo = chunk_mask_offsets ? chunk_mask_offsets[i] : 0;
                    if (mask_role.holds(chunk_masks[i])) {
                        const uintptr_t addr = (uintptr_t)chunk_masks[i], base = addr & ~(uintptr_t)7;
                        words = (const uint64_t*)base;
                        bit_off = mo + (size_t)(addr - base) * 8;
                    } else {
                        MA_TRY(scope.in_mask(chunk_masks[i], mo, chunk_lens[i], &words, &bit_off));
                        mask_role.learn(chunk_masks[i]);
                    }
                }
                if (words) {  // to the word that holds row 0: six bits of offset are left
                    words += bit_off >> 6;
                    bit_off &= 63;
                }
                if (has_join)
                    mdesc[i] = MaskDesc{row, chunk_lens[i], words, bit_off,
                                        words ? (bit_off + chunk_lens[i] - 1) >> 6 : 0};
            }
            cd[i - c0] = ConcatChunk{p, (uint64_t)row, (uint64_t)chunk_lens[i] | ((uint64_t)bit_off << 58), words};
            row += chunk_lens[i];
        }
        const void* tab = nullptr;
        TableSlotGuard guard(ctx);  // the slot is released (an event behind the launch) on every way out of this iteration
        MA_TRY(table_commit_mapped(ctx, cd, &tab, &guard.slot));
        const int n = (int)(c1 - c0);
        const int grid = grid_for(ctx, (size_t)n, 6);
        const ConcatChunk* tcd = (const ConcatChunk*)tab;
        // two 4 x 16-byte tiles per 8192-row chunk (4-byte), two 8 x 16-byte tiles (8-byte); one tile for the 1- and
        // 2-byte columns (2 / 4 x 16 bytes per lane)
#define MA_CONCAT_CHUNKS(T, U)                                                                                              \
    do {                                                                                                                     \
if (has_mask)                                                                                                        \
    hipLaunchKernelGGL((concat_chunk_kernel<T, U, true>), dim3(grid), dim3(kBlock), 0, ctx->stream, tcd, n, (T*)po, ow, \
                       total);                                                                                       \
else                                                                                                                 \
    hipLaunchKernelGGL((concat_chunk_kernel<T, U, false>), dim3(grid), dim3(kBlock), 0, ctx->stream, tcd, n, (T*)po, ow, \
                       total);                                                                                       \
    } while (0)
        switch (elem_size) {
            case 1: MA_CONCAT_CHUNKS(uint8_t, 2); break;
            case 2: MA_CONCAT_CHUNKS(uint16_t, 4); break;
            case 4: MA_CONCAT_CHUNKS(uint32_t, 4); break;
            default: MA_CONCAT_CHUNKS(uint64_t, 8); break;
        }
#undef MA_CONCAT_CHUNKS
        MA_HIP(hipGetLastError());
        c0 = c1;
        if (seg < kMax) seg = seg * growth < kMax ? seg * growth : kMax;
    }
    if (has_join) {  // the join pass walks the list either side of a chunk: one compact table, uploaded
        void* dm = nullptr;
        MA_TRY(ctx_scratch(ctx, sizeof(MaskDesc) * n_chunks, &dm));
        MA_TRY(upload_table(ctx, mdesc.data(), sizeof(MaskDesc) * n_chunks, dm));
        const int grid = grid_for(ctx, (n_chunks + kBlock - 1) / kBlock, 8);
        hipLaunchKernelGGL(concat_mask_joins_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, (const MaskDesc*)dm,
                           (int)n_chunks, total, ow);
        MA_HIP(hipGetLastError());
    }
    return MA_OK;
}

// Whether a chunk list takes the chunk-per-workgroup form: enough chunks to keep every workgroup busy with an even share
// (>= 4 per CU), short on average, none so long that its workgroup becomes the tail. variant bit 128 keeps the tile form,
// bit 256 forces the chunk form (tuning / tests).
static bool chunk_form_wanted(const ma_ctx* ctx, size_t n_chunks, const size_t* chunk_lens, size_t total) {
    size_t longest = 0;
    for (size_t i = 0; i < n_chunks; ++i)
        if (chunk_lens[i] > longest) longest = chunk_lens[i];
    const size_t avg = total / n_chunks;
    bool by_chunk = n_chunks >= (size_t)4 * (size_t)ctx->num_cus && avg <= ((size_t)1 << 16) && longest <= 8 * (avg ? avg : 1);
    if (form_variant(ctx) & 128) by_chunk = false;
    if (form_variant(ctx) & 256) by_chunk = true;
    return by_chunk;
}

extern "C" ma_status ma_consolidate_column(ma_ctx* ctx, size_t elem_size, size_t n_chunks, const void* const* chunk_data,
                                           const size_t* chunk_lens, const uint8_t* const* chunk_masks,
                                           const size_t* chunk_mask_offsets, void* out_data, uint8_t* out_mask,
                                           int32_t* out_has_mask) {
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    MA_REQUIRE(elem_size == 1 || elem_size == 2 || elem_size == 4 || elem_size == 8, MA_ERR_UNSUPPORTED,
               "element size %zu is not a numeric column width", elem_size);
    // super_table.rs:693-696 / :728-731: "consolidate() called on empty SuperTable"
    MA_REQUIRE(n_chunks > 0, MA_ERR_INVALID_ARGUMENT, "consolidate() called on empty SuperTable");
    MA_REQUIRE(n_chunks < ((size_t)1 << 30), MA_ERR_INVALID_ARGUMENT, "too many chunks");
    MA_REQUIRE(chunk_data != nullptr && chunk_lens != nullptr, MA_ERR_INVALID_ARGUMENT, "NULL chunk table");
    bool has_mask = false;
    size_t total = 0;
    for (size_t i = 0; i < n_chunks; ++i) {
        MA_REQUIRE(chunk_lens[i] == 0 || chunk_data[i] != nullptr, MA_ERR_INVALID_ARGUMENT, "chunk %zu data is NULL", i);
        MA_REQUIRE(((uintptr_t)chunk_data[i] % elem_size) == 0, MA_ERR_INVALID_ARGUMENT, "chunk %zu is misaligned", i);
        if (chunk_masks && chunk_masks[i]) has_mask = true;
        total += chunk_lens[i];
    }
    if (out_has_mask) *out_has_mask = has_mask ? 1 : 0;
    if (total == 0) return MA_OK;
    MA_REQUIRE(out_data != nullptr, MA_ERR_INVALID_ARGUMENT, "out_data is NULL");
    MA_REQUIRE(!has_mask || out_mask != nullptr, MA_ERR_INVALID_ARGUMENT, "a chunk carries nulls but out_mask is NULL");
    // every argument check comes BEFORE the first enqueue: a failure past that point would return with kernels in flight
    MA_REQUIRE(!has_mask || ((uintptr_t)out_mask & 7) == 0, MA_ERR_INVALID_ARGUMENT,
               "output bitmap must be 8-byte aligned (got %p)", (const void*)out_mask);

    MA_ENTER(ctx);
    MA_NO_CAPTURE(ctx, "consolidation (descriptor upload)");
    MA_HIP(hipSetDevice(ctx->device));
    CallScope scope(ctx);
    void* po = nullptr;
    MA_TRY(scope.out(out_data, total * elem_size, &po));
    uint64_t* ow = nullptr;
    if (has_mask) MA_TRY(scope.out_mask(out_mask, total, &ow));
    // Many chunks of a few tiles each: the chunk-per-workgroup kernel on pinned-host descriptors (1- and 2-byte columns too:
    // 60 000 x 8192 rows 0.26 / 0.37 ms against 0.30-0.44 / 0.42 for the tile search — the host's 4.4 ns per chunk is the floor),
    // the list cut into segments of 4096, 8192, ... 32768 chunks so that the GPU copies segment k while the host describes
    // segment k + 1. variant bit 128 keeps the tile form, bit 256 forces the chunk form (tuning / tests).
    if (chunk_form_wanted(ctx, n_chunks, chunk_lens, total)) {
        MA_TRY(concat_column_by_chunks(ctx, scope, elem_size, n_chunks, chunk_data, chunk_lens, chunk_masks, chunk_mask_offsets, po,
                                       has_mask ? ow : nullptr, total));
        return end_call(ctx, scope);
    }
    ChunkDesc* desc = nullptr;  // built in the context's pinned staging buffer (ma::table_begin / table_commit)
    MA_TRY(table_begin(ctx, sizeof(ChunkDesc) * n_chunks, (void**)&desc));
    const size_t tile_rows = elem_size == 1 ? tile_rows_of<uint8_t>() : elem_size == 2 ? tile_rows_of<uint16_t>()
                           : elem_size == 4 ? tile_rows_of<uint32_t>() : tile_rows_of<uint64_t>();
    size_t row = 0, n_tiles = 0;
    DeviceRange data_role, mask_role;
    for (size_t i = 0; i < n_chunks; ++i) {
        ChunkDesc& d = desc[i];
        if (data_role.holds(chunk_data[i])) {  // same device allocation as the previous chunk: no classification call
            d.data = chunk_data[i];
        } else {
            const void* p = nullptr;
            MA_TRY(scope.in(chunk_data[i], chunk_lens[i] * elem_size, &p));
            d.data = p;
            if (chunk_lens[i]) data_role.learn(chunk_data[i]);
        }
        d.start = row;
        d.len = chunk_lens[i];
        d.words = nullptr;
        d.bit_off = 0;
        d.last_word = 0;
        if (chunk_masks && chunk_masks[i] && chunk_lens[i]) {
            const size_t mo = chunk_mask_offsets ? chunk_mask_offsets[i] : 0;
            if (mask_role.holds(chunk_masks[i])) {
                const uintptr_t addr = (uintptr_t)chunk_masks[i], base = addr & ~(uintptr_t)7;  // CallScope::in_mask's re-basing
                d.words = (const uint64_t*)base;
                d.bit_off = mo + (size_t)(addr - base) * 8;
            } else {
                MA_TRY(scope.in_mask(chunk_masks[i], mo, chunk_lens[i], &d.words, &d.bit_off));
                mask_role.learn(chunk_masks[i]);
            }
            d.last_word = (d.bit_off + d.len - 1) >> 6;
        }
        // copy tiles of this chunk
        const uintptr_t dst_addr = (uintptr_t)po + row * elem_size;
        const uintptr_t mis = dst_addr & 15;
        d.head = mis ? (unsigned)((16 - mis) / elem_size) : 0;
        d.tile0 = n_tiles;
        if (d.len) n_tiles += d.len > d.head ? (d.len - d.head + tile_rows - 1) / tile_rows : 1;
        row += chunk_lens[i];
    }
    void* ddesc = nullptr;  // descriptor table -> device scratch, through the context's pinned staging (no stream drain)
    MA_TRY(ctx_scratch(ctx, sizeof(ChunkDesc) * n_chunks, &ddesc));
    MA_TRY(table_commit(ctx, desc, sizeof(ChunkDesc) * n_chunks, ddesc));

    const ChunkDesc* d = (const ChunkDesc*)ddesc;
    switch (elem_size) {
        case 1: launch_concat<uint8_t>(ctx, d, (int)n_chunks, n_tiles, po); break;
        case 2: launch_concat<uint16_t>(ctx, d, (int)n_chunks, n_tiles, po); break;
        case 4: launch_concat<uint32_t>(ctx, d, (int)n_chunks, n_tiles, po); break;
        default: launch_concat<uint64_t>(ctx, d, (int)n_chunks, n_tiles, po); break;
    }
    MA_HIP(hipGetLastError());
    if (has_mask) {
        launch_concat_mask(ctx, d, n_chunks, total, ow);
        MA_HIP(hipGetLastError());
    }
    return end_call(ctx, scope);
}

// ------------------------------------------------------------------------------------------------
// Whole-table consolidation into one arena: consolidate_tables_arena (src/structs/arena.rs:1187-1340) for numeric
// columns. The layout is the reference's cursor rule (align_cursor + reserve_slice, arena.rs:152-232).
// ------------------------------------------------------------------------------------------------
extern "C" ma_status ma_arena_layout(size_t n_cols, const size_t* elem_sizes, const int32_t* has_nulls, size_t n_rows,
                                     size_t* out_data_offsets, size_t* out_mask_offsets, size_t* out_capacity_bytes,
                                     size_t* out_used_bytes) {
    MA_REQUIRE(n_cols == 0 || elem_sizes != nullptr, MA_ERR_INVALID_ARGUMENT, "elem_sizes is NULL");
    auto align64 = [](size_t b) { return (b + 63) & ~(size_t)63; };  // src/utils.rs:178-180
    const size_t mask_bytes = (n_rows + 7) / 8;
    size_t cursor = 0, capacity = 0;
    for (size_t c = 0; c < n_cols; ++c) {
        const size_t e = elem_sizes[c];
        MA_REQUIRE(e == 1 || e == 2 || e == 4 || e == 8, MA_ERR_UNSUPPORTED, "column %zu: element size %zu is not a numeric width", c, e);
        cursor = align64(cursor);
        if (out_data_offsets) out_data_offsets[c] = cursor;
        cursor += n_rows * e;
        capacity += align64(n_rows * e);
        const bool nulls = has_nulls && has_nulls[c];
        if (nulls) {
            cursor = align64(cursor);
            if (out_mask_offsets) out_mask_offsets[c] = cursor;
            cursor += mask_bytes;
            capacity += align64(mask_bytes);
        } else if (out_mask_offsets) {
            out_mask_offsets[c] = SIZE_MAX;
        }
    }
    if (out_capacity_bytes) *out_capacity_bytes = capacity;
    if (out_used_bytes) *out_used_bytes = cursor;
    return MA_OK;
}

extern "C" ma_status ma_consolidate_table_arena(ma_ctx* ctx, size_t n_cols, size_t n_batches, const size_t* elem_sizes,
                                                const size_t* batch_rows, const void* const* cell_data,
                                                const uint8_t* const* cell_masks, const size_t* cell_mask_offsets,
                                                void* arena, size_t arena_bytes, size_t* out_data_offsets,
                                                size_t* out_mask_offsets, size_t* out_used_bytes) {
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    // arena.rs:1196: "consolidate called on empty table set"
    MA_REQUIRE(n_batches > 0, MA_ERR_INVALID_ARGUMENT, "consolidate called on empty table set");
    MA_REQUIRE(n_cols > 0 && elem_sizes && batch_rows && cell_data, MA_ERR_INVALID_ARGUMENT, "NULL or empty table description");
    MA_REQUIRE(n_cols < ((size_t)1 << 20) && n_batches < ((size_t)1 << 30) && n_cols * n_batches < ((size_t)1 << 30),
               MA_ERR_INVALID_ARGUMENT, "too many cells");
    size_t n_rows = 0;
    for (size_t b = 0; b < n_batches; ++b) n_rows += batch_rows[b];
    std::vector<int32_t> has_nulls(n_cols, 0);
    if (cell_masks)
        for (size_t c = 0; c < n_cols; ++c)
            for (size_t b = 0; b < n_batches; ++b)
                if (cell_masks[c * n_batches + b]) has_nulls[c] = 1;
    std::vector<size_t> data_off(n_cols), mask_off(n_cols);
    size_t capacity = 0, used = 0;
    MA_TRY(ma_arena_layout(n_cols, elem_sizes, has_nulls.data(), n_rows, data_off.data(), mask_off.data(), &capacity, &used));
    if (out_data_offsets) memcpy(out_data_offsets, data_off.data(), sizeof(size_t) * n_cols);
    if (out_mask_offsets) memcpy(out_mask_offsets, mask_off.data(), sizeof(size_t) * n_cols);
    if (out_used_bytes) *out_used_bytes = used;
    if (n_rows == 0) return MA_OK;
    MA_REQUIRE(arena != nullptr, MA_ERR_INVALID_ARGUMENT, "arena is NULL");
    MA_REQUIRE(((uintptr_t)arena & 63) == 0, MA_ERR_INVALID_ARGUMENT, "the arena must be 64-byte aligned (Vec64)");
    // Arena::reserve_slice asserts "Arena overflow" (arena.rs:210-216); the reference sizes it with the rounded regions.
    MA_REQUIRE(arena_bytes >= capacity, MA_ERR_INVALID_ARGUMENT, "Arena overflow: need %zu bytes, capacity is %zu", capacity, arena_bytes);
    for (size_t c = 0; c < n_cols; ++c)
        for (size_t b = 0; b < n_batches; ++b) {
            const void* p = cell_data[c * n_batches + b];
            MA_REQUIRE(batch_rows[b] == 0 || p != nullptr, MA_ERR_INVALID_ARGUMENT, "column %zu batch %zu: data is NULL", c, b);
            MA_REQUIRE(((uintptr_t)p % elem_sizes[c]) == 0, MA_ERR_INVALID_ARGUMENT, "column %zu batch %zu is misaligned", c, b);
        }

    MA_ENTER(ctx);
    MA_NO_CAPTURE(ctx, "consolidation (descriptor upload)");
    MA_HIP(hipSetDevice(ctx->device));
    CallScope scope(ctx);
    void* pa = nullptr;
    MA_TRY(scope.out(arena, capacity, &pa));
    // A pageable arena is staged through a temporary that is copied back whole: give its padding defined (zero) bytes,
    // as Arena::with_capacity pre-fills (arena.rs:125-130). A device-reachable arena keeps whatever its padding held.
    if (pa != arena) MA_HIP(hipMemsetAsync(pa, 0, capacity, ctx->stream));
    // A SuperTable of many short batches (RechunkStrategy::Auto: 8192 rows): every column takes the chunk-per-workgroup form
    // of the single-column consolidate (in-place descriptors, segments, validity by the chunk's own workgroup) — cells of one
    // column are contiguous in the cell tables. 20 000 batches x 4 columns: 0.69 -> see DESIGN.md 3.2b.
    if (chunk_form_wanted(ctx, n_batches, batch_rows, n_rows)) {
        for (size_t c = 0; c < n_cols; ++c)
            MA_TRY(concat_column_by_chunks(ctx, scope, elem_sizes[c], n_batches, cell_data + c * n_batches, batch_rows,
                                           cell_masks ? cell_masks + c * n_batches : nullptr,
                                           cell_mask_offsets ? cell_mask_offsets + c * n_batches : nullptr, (char*)pa + data_off[c],
                                           has_nulls[c] ? (uint64_t*)((char*)pa + mask_off[c]) : nullptr, n_rows));
        return end_call(ctx, scope);
    }
    // Descriptor table: first the copy descriptors grouped by element width (one launch per width; `start` counts
    // elements from the arena base — regions are 64-byte aligned, so every data offset is a whole number of elements),
    // then one run of validity descriptors per nullable column (`start` = the batch's first row in the column).
    struct Group { size_t first = 0, count = 0, n_tiles = 0; };
    Group groups[4];
    std::vector<ChunkDesc> desc;
    desc.reserve(n_cols * n_batches * 2);
    const size_t widths[4] = {1, 2, 4, 8};
    for (int g = 0; g < 4; ++g) {
        const size_t e = widths[g];
        const size_t tile_rows = e == 1 ? tile_rows_of<uint8_t>() : e == 2 ? tile_rows_of<uint16_t>()
                               : e == 4 ? tile_rows_of<uint32_t>() : tile_rows_of<uint64_t>();
        groups[g].first = desc.size();
        for (size_t c = 0; c < n_cols; ++c) {
            if (elem_sizes[c] != e) continue;
            size_t row = 0;
            for (size_t b = 0; b < n_batches; ++b) {
                ChunkDesc d{};
                const void* p = nullptr;
                MA_TRY(scope.in(cell_data[c * n_batches + b], batch_rows[b] * e, &p));
                d.data = p;
                d.start = data_off[c] / e + row;
                d.len = batch_rows[b];
                const uintptr_t mis = ((uintptr_t)pa + d.start * e) & 15;
                d.head = mis ? (unsigned)((16 - mis) / e) : 0;
                d.tile0 = groups[g].n_tiles;
                if (d.len) groups[g].n_tiles += d.len > d.head ? (d.len - d.head + tile_rows - 1) / tile_rows : 1;
                desc.push_back(d);
                row += batch_rows[b];
            }
        }
        groups[g].count = desc.size() - groups[g].first;
    }
    std::vector<size_t> mask_first(n_cols, 0);
    for (size_t c = 0; c < n_cols; ++c) {
        if (!has_nulls[c]) continue;
        mask_first[c] = desc.size();
        size_t row = 0;
        for (size_t b = 0; b < n_batches; ++b) {
            ChunkDesc d{};
            d.start = row;
            d.len = batch_rows[b];
            const uint8_t* m = cell_masks[c * n_batches + b];
            if (m && d.len) {
                MA_TRY(scope.in_mask(m, cell_mask_offsets ? cell_mask_offsets[c * n_batches + b] : 0, d.len, &d.words, &d.bit_off));
                d.last_word = (d.bit_off + d.len - 1) >> 6;
            }
            desc.push_back(d);
            row += batch_rows[b];
        }
    }
    void* ddesc = nullptr;
    MA_TRY(ctx_scratch(ctx, sizeof(ChunkDesc) * desc.size(), &ddesc));
    MA_TRY(upload_table(ctx, desc.data(), sizeof(ChunkDesc) * desc.size(), ddesc));
    const ChunkDesc* d = (const ChunkDesc*)ddesc;
    for (int g = 0; g < 4; ++g) {
        if (!groups[g].n_tiles) continue;
        const ChunkDesc* dg = d + groups[g].first;
        const int k = (int)groups[g].count;
        switch (widths[g]) {
            case 1: launch_concat<uint8_t>(ctx, dg, k, groups[g].n_tiles, pa); break;
            case 2: launch_concat<uint16_t>(ctx, dg, k, groups[g].n_tiles, pa); break;
            case 4: launch_concat<uint32_t>(ctx, dg, k, groups[g].n_tiles, pa); break;
            default: launch_concat<uint64_t>(ctx, dg, k, groups[g].n_tiles, pa); break;
        }
        MA_HIP(hipGetLastError());
    }
    for (size_t c = 0; c < n_cols; ++c) {
        if (!has_nulls[c]) continue;
        launch_concat_mask(ctx, d + mask_first[c], n_batches, n_rows, (uint64_t*)((char*)pa + mask_off[c]));
        MA_HIP(hipGetLastError());
    }
    return end_call(ctx, scope);
}

// ------------------------------------------------------------------------------------------------
// Bit-packed columns: BooleanArray data and stand-alone bitmaps.
//   Bitmask::extend_from_bitmask_range / extend_from_slice      src/structs/bitmask.rs:520-592
//   BooleanArray::append_range (data bits, then the mask rules)  src/structs/variants/boolean.rs:627-653
//   Arena::write_boolean_slices                                  src/structs/arena.rs:391-430
// Chunk i contributes bits [offset_i, offset_i + len_i) of its bitmap; joins fall on arbitrary bit positions. The
// reference's unaligned paths go bit by bit (bitmask.rs:571-583); here every output word is assembled from at most
// a few funnel-shifted source words (concat_mask_kernel above). Validity follows the numeric rule: present iff any
// chunk has one, chunks without one contribute all-valid rows (boolean.rs:638-650, arena.rs:416-421).
// ------------------------------------------------------------------------------------------------
extern "C" ma_status ma_consolidate_boolean_column(ma_ctx* ctx, size_t n_chunks, const uint8_t* const* chunk_bits,
                                                   const size_t* chunk_bit_offsets, const size_t* chunk_lens,
                                                   const uint8_t* const* chunk_masks, const size_t* chunk_mask_offsets,
                                                   uint8_t* out_bits, uint8_t* out_mask, int32_t* out_has_mask) {
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    MA_REQUIRE(n_chunks > 0, MA_ERR_INVALID_ARGUMENT, "consolidate() called on empty SuperTable");
    MA_REQUIRE(n_chunks < ((size_t)1 << 30), MA_ERR_INVALID_ARGUMENT, "too many chunks");
    MA_REQUIRE(chunk_bits != nullptr && chunk_lens != nullptr, MA_ERR_INVALID_ARGUMENT, "NULL chunk table");
    bool has_mask = false;
    size_t total = 0;
    for (size_t i = 0; i < n_chunks; ++i) {
        MA_REQUIRE(chunk_lens[i] == 0 || chunk_bits[i] != nullptr, MA_ERR_INVALID_ARGUMENT, "chunk %zu bits are NULL", i);
        if (chunk_masks && chunk_masks[i]) has_mask = true;
        total += chunk_lens[i];
    }
    if (out_has_mask) *out_has_mask = has_mask ? 1 : 0;
    if (total == 0) return MA_OK;
    MA_REQUIRE(out_bits != nullptr, MA_ERR_INVALID_ARGUMENT, "out_bits is NULL");
    MA_REQUIRE(!has_mask || out_mask != nullptr, MA_ERR_INVALID_ARGUMENT, "a chunk carries nulls but out_mask is NULL");

    MA_ENTER(ctx);
    MA_NO_CAPTURE(ctx, "consolidation (descriptor upload)");
    MA_HIP(hipSetDevice(ctx->device));
    CallScope scope(ctx);
    // Compact 40-byte descriptors (a Boolean column rechunked at 8192 rows is 122 000 chunks per 10^9 rows: the call is
    // host- and table-bound, not data-bound); a chunked column's bitmaps run through a few allocations, so each operand
    // role remembers the device range its last pointer fell into (two compares instead of a classification call).
    MaskDesc* data_desc = nullptr;  // both tables are built in the context's pinned staging buffer (no second copy of 5 MB)
    MA_TRY(table_begin(ctx, sizeof(MaskDesc) * n_chunks * (has_mask ? 2 : 1), (void**)&data_desc));
    MaskDesc* mask_desc = has_mask ? data_desc + n_chunks : nullptr;
    DeviceRange data_role, mask_role;
    auto describe = [&](DeviceRange& role, const uint8_t* bits, size_t off, size_t len, MaskDesc& d) -> ma_status {
        if (role.holds(bits)) {
            const uintptr_t addr = (uintptr_t)bits, base = addr & ~(uintptr_t)7;  // CallScope::in_mask's re-basing
            d.words = (const uint64_t*)base;
            d.bit_off = off + (size_t)(addr - base) * 8;
        } else {
            MA_TRY(scope.in_mask(bits, off, len, &d.words, &d.bit_off));
            role.learn(bits);
        }
        d.last_word = (d.bit_off + len - 1) >> 6;
        return MA_OK;
    };
    size_t row = 0;
    bool has_join = false;  // a non-empty chunk that starts inside an output word
    for (size_t i = 0; i < n_chunks; ++i) {
        MaskDesc d{row, chunk_lens[i], nullptr, 0, 0};
        MaskDesc m = d;
        has_join |= chunk_lens[i] != 0 && (row & 63) != 0;
        if (d.len) {
            MA_TRY(describe(data_role, chunk_bits[i], chunk_bit_offsets ? chunk_bit_offsets[i] : 0, d.len, d));
            if (has_mask && chunk_masks[i])
                MA_TRY(describe(mask_role, chunk_masks[i], chunk_mask_offsets ? chunk_mask_offsets[i] : 0, m.len, m));
        }
        data_desc[i] = d;
        if (has_mask) mask_desc[i] = m;
        row += chunk_lens[i];
    }
    // both descriptor tables in one scratch region
    void* tables = nullptr;
    MA_TRY(ctx_scratch(ctx, sizeof(MaskDesc) * n_chunks * 2, &tables));
    MaskDesc* dd = (MaskDesc*)tables;
    MaskDesc* md = dd + n_chunks;
    MA_TRY(table_commit(ctx, data_desc, sizeof(MaskDesc) * n_chunks * (has_mask ? 2 : 1), dd));
    uint64_t *ow = nullptr, *mw = nullptr;  // both outputs are validated before the first launch
    MA_TRY(scope.out_mask(out_bits, total, &ow));
    if (has_mask) MA_TRY(scope.out_mask(out_mask, total, &mw));
    launch_concat_bits(ctx, dd, n_chunks, total, ow, has_join);
    MA_HIP(hipGetLastError());
    if (has_mask) {
        launch_concat_bits(ctx, md, n_chunks, total, mw, has_join);
        MA_HIP(hipGetLastError());
    }
    return end_call(ctx, scope);
}
