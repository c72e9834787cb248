// Per-column sums of MANY columns in two launches — the "per-column reduce" of a wide or chunked table
// (BASELINE config 5; the reference sums column by column, one loop per Array: benches/hotloop_benchmark_std.rs:
// 109-127 through the enum, src/structs/chunked/super_table.rs columns). One kernel launch per column costs ~4 us of
// dispatch on MI355X whatever its size (profiles/r01_launch_bound.json), so a 1000-column table of 1000-row columns is
// launch-bound by three orders of magnitude; here the launch count is independent of the column count.
//
//   pass 1  every column is cut into segments of seg_rows(element size) rows; one workgroup reduces one segment at a time
//           (16-byte loads, validity words per wave run — the body of ma_reduce.hip's sum_kernel) and stores a
//           32-byte partial per segment;
//   pass 2  one wave per column folds that column's partials in index order and writes {sum, count}.
// Same accumulators as the single-column kernel (wrapping u64 / double-double), so integer results are bit-exact and
// float results within 1 ULP of the exactly rounded sum, independent of the segmentation.
#include <type_traits>
#include <vector>

#include "ma_acc.hpp"
#include "ma_device.hpp"

namespace ma {

struct ColDesc {
    const void* data;       // first element of the column window
    size_t len;             // rows
    const uint64_t* words;  // validity words (8-byte aligned base) or nullptr = dense
    size_t bit_off;         // bit index of row 0 relative to `words`
    size_t last_word;       // last word index holding a window bit
    size_t seg0;            // index of this column's first segment (prefix sum)
};

// Columns of a segment or less each: 32 bytes, wave-uniform loads a column ahead of their use — column c IS partial c, nothing
// to search. The table reaches the device on the context's upload stream, beside the stream's work (TableUpload): no copy on the
// stream in front of the kernel (0.1 ms of a 0.7-ms call for 60 000 columns), no PCIe read per column either (rounds 3-5 read
// it in place).
struct ShortCol {
    const void* data;
    size_t len;
    const uint64_t* words;  // validity words (8-byte aligned base) or nullptr = dense
    size_t bit_off;
};

// Rows per segment: 65 536 for 4- and 8-byte values (256 / 512 KiB); the 1- and 2-byte types take 524 288 / 262 144 rows
// (512 KiB): at 65 536 rows a u8 segment is eight tiles and the per-segment work (search, partial, fold) kept the scan at
// 4.4 TB/s against the single-column kernel's 6.1.
constexpr size_t seg_rows(size_t elem) { return elem >= 4 ? ((size_t)1 << 16) : ((size_t)1 << 19) / elem; }

// The column of segment `seg`: the LAST c with cols[c].seg0 <= seg (an empty column shares its seg0 with the next one).
// 64 probes at a time, one per lane, instead of a bisection: two dependent loads for up to 4096 columns where the bisection
// made twelve — a workgroup streams nothing while it looks for its next segment.
__device__ __forceinline__ int find_col(const ColDesc* __restrict__ c, int n_cols, size_t seg, unsigned lane) {
    int lo = 0, n = n_cols;  // the answer lies in [lo, lo + n); c[lo].seg0 <= seg throughout
    while (n > 1) {
        const int step = (n + 63) >> 6;
        const int idx = lo + (int)lane * step;
        const bool le = idx < lo + n && c[idx].seg0 <= seg;
        const int k = __popcll(__ballot(le)) - 1;  // the predicate is monotone over the lanes; lane 0 holds
        const int end = lo + n;
        lo += k * step;
        n = end - lo < step ? end - lo : step;
    }
    return lo;
}

// PER_WAVE: a WAVE (not a workgroup) takes a segment and writes its partial itself — no barrier, no LDS merge, four segments
// in flight per workgroup. For columns of a segment or less each (a chunked column handed over chunk by chunk: 8192-row
// "columns"), where a workgroup's time per segment was latency — descriptor, barriers, partial — rather than bandwidth.
template <typename T, int UNROLL, bool PER_WAVE>
__global__ __launch_bounds__(kBlock) void column_segments_kernel(const void* __restrict__ table, int n_cols,
                                                                 size_t n_segs, Partial* __restrict__ partials) {
    // 1- and 2-byte columns keep their 16 bytes as four dwords: narrow_vec_sum works on dwords anyway, and a vector of 1-byte
    // elements lost the loads' non-temporal hint on the way through the optimiser (u8 / i8 read at 6.2 TB/s, u16 at 6.9)
    typedef typename std::conditional<(sizeof(T) <= 2), MaU4, typename Vec16<T>::type>::type V;
    typedef typename AccOf<T>::type Acc;
    constexpr int R = 16 / (int)sizeof(T);
    constexpr int WPT = R * UNROLL;
    constexpr bool kNarrow = sizeof(T) <= 2;  // 8 / 16 rows per load, summed inside 32-bit registers (narrow_vec_sum)
    constexpr size_t WAVE_ROWS = (size_t)64 * R * UNROLL;
    constexpr size_t TILE_ROWS = PER_WAVE ? WAVE_ROWS : WAVE_ROWS * kWaves;  // rows per trip of the unit that owns the segment
    static_assert(WPT < 64, "a wave must be able to load its validity words in one instruction");
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned unit_tid = PER_WAVE ? lane : tid;        // index inside the owning unit ...
    constexpr unsigned kUnit = PER_WAVE ? 64u : (unsigned)kBlock;  // ... and its size
    __shared__ Partial lds[PER_WAVE ? 1 : kWaves];

    const size_t seg_first = PER_WAVE ? (size_t)blockIdx.x * kWaves + wave : (size_t)blockIdx.x;
    const size_t seg_stride = PER_WAVE ? (size_t)gridDim.x * kWaves : (size_t)gridDim.x;
    const ColDesc* __restrict__ cols = (const ColDesc*)table;
    const ShortCol* __restrict__ shorts = (const ShortCol*)table;
    ShortCol e{}, e_next{};
    if constexpr (PER_WAVE) {
        if (seg_first < n_segs) e = shorts[__builtin_amdgcn_readfirstlane((int)seg_first)];
    }
    for (size_t seg = seg_first; seg < n_segs; seg += seg_stride) {
        ColDesc d;
        if constexpr (PER_WAVE) {  // segment s = column s: the descriptor was fetched during the previous column's rows
            if (seg + seg_stride < n_segs) e_next = shorts[__builtin_amdgcn_readfirstlane((int)(seg + seg_stride))];
            d.data = e.data;
            d.len = e.len;
            d.words = e.len ? e.words : nullptr;
            d.bit_off = e.bit_off;
            d.last_word = e.len ? (e.bit_off + e.len - 1) >> 6 : 0;
            d.seg0 = seg;
            e = e_next;
        } else {
            d = cols[__builtin_amdgcn_readfirstlane(find_col(cols, n_cols, seg, lane))];  // workgroup-uniform
        }
        constexpr size_t kSegRows = seg_rows(sizeof(T));
        const size_t r_begin = (seg - d.seg0) * kSegRows;
        const size_t r_end = r_begin + kSegRows < d.len ? r_begin + kSegRows : d.len;
        const T* __restrict__ data = (const T*)d.data;
        const bool masked = d.words != nullptr;

        Acc acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r].init();
        uint64_t cnt = 0;

        // rows in front of the first 16-byte boundary of this segment, full tiles, then the rest
        size_t head = 0;
        if (r_begin < r_end) {
            const uintptr_t mis = (uintptr_t)(data + r_begin) & 15;
            head = mis ? (16 - mis) / sizeof(T) : 0;
            if (head > r_end - r_begin) head = r_end - r_begin;
        }
        const size_t body0 = r_begin + head;
        const size_t n_tiles = (r_end - body0) / TILE_ROWS;
        // The tiles of the segment with the NEXT tile's rows (and raw validity words) requested before this one is consumed —
        // round 4; ma_reduce.hip's masked kernels have the reasoning and the rules (always UNROLL + 1 loads per request, with
        // dummies from `partials` when nothing is left; two register sets that swap roles; the funnel shift deferred) — once
        // per validity (uniform over the segment).
        auto run = [&](auto masked_c) {
            constexpr bool M = decltype(masked_c)::value;
            auto issue = [&](size_t t, V (&v)[UNROLL], uint64_t& raw, size_t& row0) {
                const bool real = t < n_tiles;
                row0 = body0 + (real ? t : 0) * TILE_ROWS + (PER_WAVE ? 0 : (size_t)wave * WAVE_ROWS);
                const V* __restrict__ p = real ? (const V*)(data + row0) + lane : (const V*)partials;
                const size_t stride = real ? 64 : 0;
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) v[u] = load16<V, true>(p + (size_t)u * stride);
                if constexpr (M) {
                    size_t idx = ((d.bit_off + row0) >> 6) + (lane < (unsigned)WPT ? lane : (unsigned)WPT);
                    idx = idx < d.last_word ? idx : d.last_word;
                    raw = as_global(real ? d.words : (const uint64_t*)partials)[real ? idx : 0];
                } else {
                    raw = 0;
                }
            };
            auto use = [&](const V (&v)[UNROLL], uint64_t raw, size_t row0) {
                uint64_t aw = ~(uint64_t)0;
                if constexpr (M) {
                    aw = finish_run_words(raw, d.bit_off + row0);
                    if (lane < (unsigned)WPT) cnt += (uint64_t)__popcll(aw);
                }
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    const unsigned bits = M ? lane_bits<R>(aw, u, lane) : ~0u;
                    if constexpr (kNarrow) {
                        acc[0].add(narrow_vec_sum<T>(v[u], bits));
                    } else {
#pragma unroll
                        for (int r = 0; r < R; ++r) acc[r].add(((bits >> r) & 1u) ? (T)v[u][r] : (T)0);
                    }
                }
            };
            V va[UNROLL], vb[UNROLL];
            uint64_t ra, rb;
            size_t row_a, row_b;
            issue(0, va, ra, row_a);
            for (size_t t = 0; t < n_tiles; t += 2) {
                issue(t + 1, vb, rb, row_b);
                use(va, ra, row_a);
                issue(t + 2, va, ra, row_a);
                if (t + 1 < n_tiles) use(vb, rb, row_b);
            }
        };
        if (masked) {
            if (n_tiles) run(std::true_type{});
        } else {
            // dense: UNROLL loads per lane from two waves per SIMD already fill the pipe; a tile ahead only queues (8 long i32 /
            // u8 columns 7.05 -> 6.93 TB/s with it, where the masked form gains 6.5-6.8 -> 6.9-7.0)
            for (size_t t = 0; t < n_tiles; ++t) {
                const size_t row0 = body0 + t * TILE_ROWS + (PER_WAVE ? 0 : (size_t)wave * WAVE_ROWS);
                const V* __restrict__ p = (const V*)(data + row0) + lane;
                V v[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) v[u] = load16<V, true>(p + (size_t)u * 64);
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    if constexpr (kNarrow) {
                        acc[0].add(narrow_vec_sum<T>(v[u], ~0u));
                    } else {
#pragma unroll
                        for (int r = 0; r < R; ++r) acc[r].add((T)v[u][r]);
                    }
                }
            }
        }
        const size_t tail0 = body0 + n_tiles * TILE_ROWS;
        const size_t n_ragged = head + (r_end - tail0);
        for (size_t i = unit_tid; i < n_ragged; i += kUnit) {
            const size_t row = i < head ? r_begin + i : tail0 + (i - head);
            T x = data[row];
            if (masked) {
                const unsigned valid = row_bit(d.words, d.bit_off + row);
                cnt += valid;
                x = valid ? x : (T)0;
            }
            acc[0].add(x);
        }
        if (!masked && unit_tid == 0) cnt = r_end - r_begin;

#pragma unroll
        for (int r = 1; r < R; ++r) acc[0].merge(acc[r]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            acc[0].shfl_down_merge(off);
            cnt += (uint64_t)__shfl_down((unsigned long long)cnt, off, 64);
        }
        if constexpr (PER_WAVE) {
            if (lane == 0) {
                Partial p;
                acc[0].to_partial(p);
                p.cnt = cnt;
                p.pad = 0;
                partials[seg] = p;
            }
            continue;
        }
        __syncthreads();  // lds[] of the previous segment has been consumed
        if (lane == 0) {
            acc[0].to_partial(lds[wave]);
            lds[wave].cnt = cnt;
        }
        __syncthreads();
        if (tid == 0) {
            Acc s;
            s.from_words(lds[0].a, lds[0].b);
            uint64_t n = lds[0].cnt;
#pragma unroll
            for (int w = 1; w < kWaves; ++w) {
                Acc o;
                o.from_words(lds[w].a, lds[w].b);
                s.merge(o);
                n += lds[w].cnt;
            }
            Partial p;
            s.to_partial(p);
            p.cnt = n;
            p.pad = 0;
            partials[seg] = p;
        }
    }
}

// ---- columns of a segment or less, one WAVE per column, loads a tile ahead ---------------------------------------------------
// The chunked regime (8192-row "columns": 8 or 16 tiles a wave's worth each). A scan is fastest with ~8 KiB of loads in
// flight per SIMD from ONE or TWO waves (ma_reduce.hip's launch shape; eight workgroups per CU on this path read at 6.2 TB/s,
// one at 6.7: tools/probe_sum_chunks.py) — but with so few waves nobody covers for a wave that has drained its loads at a
// column's end. So every wave keeps the NEXT tile's loads in flight while it accumulates this one, across column boundaries
// too: the first tile of column c + W is requested before column c is reduced and stored. Descriptors come a column ahead
// from the table (wave-uniform loads), requested right after a column's result is out so that no cross-lane step
// waits for them.
// TOTAL: the columns are chunks of ONE column (ma_sum_chunks) — nothing is reduced per chunk; the wave's accumulators run on
// and its one partial goes to partials[wave].
// DIRECT (per-column form only): a column is ONE wave's work here, so what its flush holds is the column's result — it goes
// straight to the caller's arrays (three 8-byte stores; an array the caller did not ask for points into `partials`, unused in this
// form) instead of into a Partial that a second launch turns into the same three values: that launch cost 19.5 us per 60 000
// columns (a wave per column, seven dependent round trips each: profiles/r06_column_waves.md) on a 280-us scan.
struct ColOut {
    double* f64;
    uint64_t* i64;
    uint64_t* cnt;
    int direct, is_signed;
};

template <typename T, int UNROLL, bool TOTAL>
__global__ __launch_bounds__(kBlock) void column_waves_kernel(const ShortCol* __restrict__ table, size_t n_cols_,
                                                              Partial* __restrict__ partials, int stagger, ColOut out) {
    typedef typename std::conditional<(sizeof(T) <= 2), MaU4, typename Vec16<T>::type>::type V;
    typedef typename AccOf<T>::type Acc;
    constexpr int R = 16 / (int)sizeof(T);
    constexpr int WPT = R * UNROLL;
    constexpr bool kNarrow = sizeof(T) <= 2;
    constexpr size_t TILE_ROWS = (size_t)64 * R * UNROLL;
    static_assert(WPT < 64, "a wave must be able to load its validity words in one instruction");
    const unsigned lane = threadIdx.x & 63;
    // Wave-uniform by construction, and said so: the column state lives in scalar registers. 32-bit wherever the value allows
    // (column indices < 2^30, a column is a segment or less): the scalar unit has no 64-bit ordered compare, and uniform
    // values that took a detour through vector registers ended up in the tiles' registers — with waits for their loads.
    const unsigned w_id = (unsigned)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * kWaves + (threadIdx.x >> 6)));
    const unsigned n_w = gridDim.x * kWaves;
    const unsigned n_cols = (unsigned)n_cols_;
    // Wave w starts every column at tile w mod (tiles of the column) and wraps: the waves advance in step, and without this the
    // four waves of a CU read the SAME offset of four adjacent chunks at any moment (i64 6.81 -> 7.05 TB/s, with validity 6.65 ->
    // 6.97, i32 6.48 -> 6.72; the same start for a workgroup's four waves and a different one per workgroup: 6.70 — it is the
    // waves that share a CU that must differ). variant bit 32768: every wave from tile 0, A/B.
    auto tile_of = [&](unsigned t, unsigned rot, unsigned n_tiles) -> unsigned {
        const unsigned r = t + rot;
        return r >= n_tiles ? r - n_tiles : r;
    };

    struct Col {
        const T* data;
        const uint64_t* words;
        size_t bit_off, last_word;
        unsigned len, head, n_tiles, rot;
    };
    auto derive = [&](const ShortCol& e) -> Col {
        Col c;
        c.data = (const T*)e.data;
        c.len = (unsigned)e.len;
        c.words = c.len ? e.words : nullptr;
        c.bit_off = e.bit_off;
        c.last_word = c.len ? (e.bit_off + c.len - 1) >> 6 : 0;
        const unsigned mis = (unsigned)(uintptr_t)c.data & 15u;
        c.head = (mis && c.len) ? (16u - mis) / (unsigned)sizeof(T) : 0u;
        if (c.head > c.len) c.head = c.len;
        c.n_tiles = (c.len - c.head) / (unsigned)TILE_ROWS;
        c.rot = (stagger && c.n_tiles) ? w_id % c.n_tiles : 0u;
        return c;
    };
    // always a load (the index is clamped instead of the load being skipped): a conditional one ends in register moves that
    // wait for the value where it was requested
    auto fetch = [&](unsigned col) -> ShortCol {
        const unsigned i = col < n_cols ? col : n_cols - 1;
        return table[i];
    };
    // Requests tile t of column c — ALWAYS nine loads: with nothing to request (`real` false: the wave's last tile, a column
    // without a full tile) every lane reads the first bytes of `partials` instead, and a dense column's "validity word" comes
    // from there too. Loads that are issued on some paths only would make the compiler's wait counts assume the shortest
    // queue (vmcnt counts in order): the wait for THIS tile would then also wait for most of the next one.
    auto issue = [&](const Col& c, unsigned t, bool real, V (&v)[UNROLL], uint64_t& raw) {
        const unsigned row0 = c.head + tile_of(t, c.rot, c.n_tiles) * (unsigned)TILE_ROWS;
        const V* __restrict__ p = real ? (const V*)(c.data + row0) + lane : (const V*)partials;
        const size_t stride = real ? 64 : 0;
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = load16<V, true>(p + (size_t)u * stride);
        const bool masked = real && c.words != nullptr;
        // lane l <= WPT holds run word l (the last one only feeds the funnel shift); clamped, not skipped, past the column's
        // last word — a full tile never needs a word behind it
        size_t idx = ((c.bit_off + row0) >> 6) + (lane < (unsigned)WPT ? lane : (unsigned)WPT);
        idx = idx < c.last_word ? idx : c.last_word;
        raw = as_global(masked ? c.words : (const uint64_t*)partials)[masked ? idx : 0];
    };

    // 32-bit integers accumulate in halves (SplitAcc: two full-rate adds per value instead of a shift, a copy and a 64-bit add —
    // this loop's issue time is what bounds it: profiles/r06_column_waves.md) and are widened once per column: a column is at most
    // 65 536 rows here, 256 adds per accumulator.
    constexpr bool kSplit = sizeof(T) == 4 && std::is_integral<T>::value;
    typedef typename std::conditional<kSplit, SplitAcc<typename std::conditional<kSplit, T, int32_t>::type>, Acc>::type TileAcc;
    TileAcc acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r].init();
    uint64_t cnt = 0;
    Acc run;  // (TOTAL with SplitAcc only) the columns so far
    run.init();
    auto widened = [&]() -> Acc {  // the R accumulators as one; they start over
#pragma unroll
        for (int r = 1; r < R; ++r) acc[0].merge(acc[r]);
        Acc w;
        if constexpr (kSplit) w = acc[0].widen();
        else w = acc[0];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r].init();
        return w;
    };
    auto flush = [&](unsigned slot) {  // (TOTAL only) lanes and slots -> the wave's one partial; the accumulators start over
        Acc w = widened();
        if constexpr (kSplit) {
            w.merge(run);
            run.init();
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            w.shfl_down_merge(off);
            cnt += (uint64_t)__shfl_down((unsigned long long)cnt, off, 64);
        }
        if (lane == 0) {
            Partial p;
            w.to_partial(p);
            p.cnt = cnt;
            p.pad = 0;
            partials[slot] = p;
        }
        cnt = 0;
    };

    // ---- per-column results without a cross-lane reduction per column (round 6) ---------------------------------------------------
    // A column of 8192 rows is four tiles; reducing it across the wave the moment it ends is six dependent shuffle stages (twice, for
    // the valid count) in front of the next column's first tile — on ONE wave per SIMD, with nobody to cover. Instead a lane PARKS
    // its partial of the column in the wave's own LDS rows (one 8-byte store per word: no dependency, no wait) and goes on; every
    // kBatch columns the wave reduces the batch transposed: lane l adds up 64 / G of column l / G's 64 parked partials (G = 64 /
    // kBatch lanes per column), log2(G) shuffle stages finish kBatch columns at once, and G-th lanes write the results. A wave's
    // columns are col, col + n_w, ...: the batch's first column and a count describe it.
    constexpr int kBatch = 8;
    constexpr bool kDD = std::is_same<Acc, DDAcc>::value;
    __shared__ uint64_t park_a[TOTAL ? 1 : kWaves][TOTAL ? 1 : kBatch][TOTAL ? 1 : 64];
    __shared__ uint64_t park_b[(TOTAL || !kDD) ? 1 : kWaves][(TOTAL || !kDD) ? 1 : kBatch][(TOTAL || !kDD) ? 1 : 64];
    __shared__ uint64_t park_c[TOTAL ? 1 : kWaves][TOTAL ? 1 : kBatch][TOTAL ? 1 : 64];
    const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    unsigned in_batch = 0, batch_first = 0;
    auto park = [&](unsigned column) {
        if (in_batch == 0) batch_first = column;
        const Acc w = widened();
        Partial p;
        w.to_partial(p);
        park_a[TOTAL ? 0 : wave][TOTAL ? 0 : in_batch][TOTAL ? 0 : lane] = p.a;
        if constexpr (kDD && !TOTAL) park_b[wave][in_batch][lane] = p.b;
        park_c[TOTAL ? 0 : wave][TOTAL ? 0 : in_batch][TOTAL ? 0 : lane] = cnt;
        cnt = 0;
        ++in_batch;
    };
    auto reduce_batch = [&]() {
        constexpr unsigned G = 64 / kBatch, kPer = 64 / G;  // G lanes per column, kPer parked partials per lane
        __builtin_amdgcn_wave_barrier();  // (a wave's LDS operations complete in order; this keeps the compiler from moving the reads up)
        const unsigned jj = lane / G, part = lane % G;
        Acc t;
        t.init();
        uint64_t n = 0;
#pragma unroll
        for (unsigned i = 0; i < kPer; ++i) {
            const unsigned idx = part * kPer + ((i + lane) & (kPer - 1));  // rotated: the G lanes of a column start on different banks
            Acc o;
            uint64_t b = 0;
            if constexpr (kDD && !TOTAL) b = park_b[wave][jj][idx];
            o.from_words(park_a[TOTAL ? 0 : wave][TOTAL ? 0 : jj][TOTAL ? 0 : idx], b);
            t.merge(o);
            n += park_c[TOTAL ? 0 : wave][TOTAL ? 0 : jj][TOTAL ? 0 : idx];
        }
#pragma unroll
        for (int off = (int)G / 2; off > 0; off >>= 1) {
            t.shfl_down_merge(off);
            n += (uint64_t)__shfl_down((unsigned long long)n, off, 64);
        }
        if (part == 0 && jj < in_batch) {
            const unsigned slot = batch_first + jj * n_w;
            if (out.direct) {
                if constexpr (kDD) {
                    t.normalise();
                    out.f64[slot] = t.hi;
                    out.i64[slot] = 0;  // (points into `partials`: float formats have no integer sums)
                } else {
                    out.i64[slot] = t.s;
                    out.f64[slot] = out.is_signed ? (double)(int64_t)t.s : (double)t.s;
                }
                out.cnt[slot] = n;
            } else {
                Partial p;
                t.to_partial(p);
                p.cnt = n;
                p.pad = 0;
                partials[slot] = p;
            }
        }
        __builtin_amdgcn_wave_barrier();  // the rows are written again from here on
        in_batch = 0;
    };

    unsigned col = w_id;
    if (col >= n_cols) {
        if (TOTAL) flush(w_id);
        return;
    }
    Col c = derive(fetch(col));
    ShortCol e_next = fetch(col + n_w);
    unsigned t = 0;
    bool have = c.n_tiles > 0;
    // One step: request the tile after (c, t) into (nv, nraw), then consume (cv, craw) = tile (c, t). Two register sets that
    // swap roles from step to step (a copy from "next" to "current" would wait for the loads it copies).
    // (Both steps of the loop below always run and the loop leaves at its end only: a step after the wave's last column
    // requests and "uses" nine loads like any other and skips the column's end. With an exit between the steps the exit shared
    // its branch with the loop's end, and the compiler's wait counts — which follow every path the branches allow, taken or
    // not — held each step's requests back until most of the tile before them had arrived.)
    bool finished = false;
    auto step = [&](V (&cv)[UNROLL], uint64_t& craw, V (&nv)[UNROLL], uint64_t& nraw) __attribute__((always_inline)) {
        // the tile after this one: the same column's, or the first of this wave's next column
        const bool col_done = !(have && t + 1 < c.n_tiles);
        const unsigned col_n = col + n_w;
        Col cn = c;
        unsigned tn = t + 1;
        bool have_n = !col_done;
        if (col_done && col_n < n_cols) {
            cn = derive(e_next);
            tn = 0;
            have_n = cn.n_tiles > 0;
        }
        issue(cn, tn, have_n, nv, nraw);

        if (have) {
            const unsigned row0 = c.head + tile_of(t, c.rot, c.n_tiles) * (unsigned)TILE_ROWS;
            if (c.words) {
                const uint64_t aw = finish_run_words(craw, c.bit_off + row0);
                if (lane < (unsigned)WPT) cnt += (uint64_t)__popcll(aw);
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    const unsigned bits = lane_bits<R>(aw, u, lane);
                    if constexpr (kNarrow) {
                        acc[0].add(narrow_vec_sum<T>(cv[u], bits));
                    } else {
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            if constexpr (kSplit) acc[r].add_if((T)cv[u][r], (bits >> r) & 1u);
                            else acc[r].add(((bits >> r) & 1u) ? (T)cv[u][r] : (T)0);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    if constexpr (kNarrow) {
                        acc[0].add(narrow_vec_sum<T>(cv[u], ~0u));
                    } else {
#pragma unroll
                        for (int r = 0; r < R; ++r) acc[r].add((T)cv[u][r]);
                    }
                }
                asm volatile("" ::"v"(craw));  // loaded, not needed: "used" all the same (see below)
            }
        } else {
            // nothing was requested for real, but the registers were loaded: "use" them, so that on this path too they are
            // known to have arrived — otherwise every path's next request waits before it overwrites them
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) asm volatile("" ::"v"(cv[u]));
            asm volatile("" ::"v"(craw));
        }
        if (col_done) {
            if (!finished) {
                // rows in front of the first 16-byte boundary and behind the last full tile
                const unsigned tail0 = c.head + c.n_tiles * (unsigned)TILE_ROWS;
                const unsigned n_ragged = c.head + (c.len - tail0);
                for (unsigned i = lane; i < n_ragged; i += 64) {
                    const unsigned row = i < c.head ? i : tail0 + (i - c.head);
                    T x = c.data[row];
                    if (c.words) {
                        const unsigned valid = row_bit(c.words, c.bit_off + row);
                        cnt += valid;
                        x = valid ? x : (T)0;
                    }
                    acc[0].add(x);
                }
                if (!c.words && lane == 0) cnt += c.len;
                if constexpr (!TOTAL) {
                    park(col);
                    if (in_batch == (unsigned)kBatch || col_n >= n_cols) reduce_batch();  // a full batch, or the wave's last column
                } else if constexpr (kSplit) {
                    run.merge(widened());  // the halves hold one chunk at most
                }
            }
            if (col_n >= n_cols) {
                finished = true;
            } else {
                col = col_n;
                e_next = fetch(col + n_w);  // a column ahead, and behind the cross-lane steps of flush()
                c = cn;
            }
        }
        t = tn;
        have = have_n;
    };
    V va[UNROLL], vb[UNROLL];
    uint64_t raw_a = 0, raw_b = 0;
    issue(c, 0, have, va, raw_a);
    do {
        step(va, raw_a, vb, raw_b);
        step(vb, raw_b, va, raw_a);
    } while (!finished);
    if (TOTAL) flush(w_id);
}

// One wave per column: fold the column's partials in index order per lane, then across lanes.
template <typename T>
__global__ __launch_bounds__(kBlock) void column_fold_kernel(const ColDesc* __restrict__ cols, int n_cols, size_t n_segs,
                                                             const Partial* __restrict__ partials, int is_signed,
                                                             double* __restrict__ out_f64, uint64_t* __restrict__ out_i64,
                                                             uint64_t* __restrict__ out_cnt, int partial_per_column) {
    typedef typename AccOf<T>::type Acc;
    const unsigned lane = threadIdx.x & 63;
    const size_t wave_id = ((size_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const size_t n_waves = ((size_t)gridDim.x * kBlock) >> 6;
    for (size_t c = wave_id; c < (size_t)n_cols; c += n_waves) {
        const size_t s0 = partial_per_column ? c : cols[c].seg0;
        const size_t s1 = partial_per_column ? c + 1 : (c + 1 < (size_t)n_cols ? cols[c + 1].seg0 : n_segs);
        Acc tot;
        tot.init();
        uint64_t cnt = 0;
        for (size_t s = s0 + lane; s < s1; s += 64) {
            Acc o;
            o.from_words(partials[s].a, partials[s].b);
            tot.merge(o);
            cnt += partials[s].cnt;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            tot.shfl_down_merge(off);
            cnt += (uint64_t)__shfl_down((unsigned long long)cnt, off, 64);
        }
        if (lane == 0) {
            if constexpr (std::is_same<Acc, DDAcc>::value) {
                tot.normalise();
                if (out_f64) out_f64[c] = tot.hi;
            } else {
                if (out_i64) out_i64[c] = tot.s;
                if (out_f64) out_f64[c] = is_signed ? (double)(int64_t)tot.s : (double)tot.s;
            }
            if (out_cnt) out_cnt[c] = cnt;
        }
    }
}

// ---- the total over ALL partials (ma_sum_chunks: the chunks are one logical column) -----------------------------------
// Level 1 (many partials): wave w folds the contiguous slice [w * per, (w + 1) * per) in index order per lane, then across
// lanes, and writes one partial. Level 2: one workgroup folds what is left and writes the result. Slices, lanes and waves
// are always combined in the same order: the double-double total is reproducible for a given chunk list.
template <typename T>
__global__ __launch_bounds__(kBlock) void partial_fold_kernel(const Partial* __restrict__ in, size_t n, size_t per,
                                                              Partial* __restrict__ out) {
    typedef typename AccOf<T>::type Acc;
    const unsigned lane = threadIdx.x & 63;
    const size_t w = ((size_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    const size_t s0 = w * per, s1 = s0 + per < n ? s0 + per : n;
    Acc tot;
    tot.init();
    uint64_t cnt = 0;
    for (size_t s = s0 + lane; s < s1; s += 64) {
        Acc o;
        o.from_words(in[s].a, in[s].b);
        tot.merge(o);
        cnt += in[s].cnt;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        tot.shfl_down_merge(off);
        cnt += (uint64_t)__shfl_down((unsigned long long)cnt, off, 64);
    }
    if (lane == 0) {
        Partial p;
        tot.to_partial(p);
        p.cnt = cnt;
        p.pad = 0;
        out[w] = p;
    }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void total_fold_kernel(const Partial* __restrict__ in, size_t n, int is_signed,
                                                            double* __restrict__ out_f64, uint64_t* __restrict__ out_i64,
                                                            uint64_t* __restrict__ out_cnt, double* __restrict__ out_lo) {
    typedef typename AccOf<T>::type Acc;
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ Partial lds[kWaves];
    Acc tot;
    tot.init();
    uint64_t cnt = 0;
    for (size_t s = tid; s < n; s += kBlock) {
        Acc o;
        o.from_words(in[s].a, in[s].b);
        tot.merge(o);
        cnt += in[s].cnt;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        tot.shfl_down_merge(off);
        cnt += (uint64_t)__shfl_down((unsigned long long)cnt, off, 64);
    }
    if (lane == 0) {
        tot.to_partial(lds[wave]);
        lds[wave].cnt = cnt;
    }
    __syncthreads();
    if (tid != 0) return;
    Acc s;
    s.from_words(lds[0].a, lds[0].b);
    uint64_t c = lds[0].cnt;
#pragma unroll
    for (int w = 1; w < kWaves; ++w) {
        Acc o;
        o.from_words(lds[w].a, lds[w].b);
        s.merge(o);
        c += lds[w].cnt;
    }
    if constexpr (std::is_same<Acc, DDAcc>::value) {
        s.normalise();
        if (out_f64) *out_f64 = s.hi;
        if (out_lo) *out_lo = s.lo;  // the pair, for a further error-free fold (a group's exchange)
    } else {
        if (out_i64) *out_i64 = s.s;
        if (out_f64) *out_f64 = is_signed ? (double)(int64_t)s.s : (double)s.s;
    }
    if (out_cnt) *out_cnt = c;
}

// ---- long columns through the same wave kernel ------------------------------------------------------------------------------
// Rows per PIECE (64 KiB) when a table of MANY long 1- or 2-byte columns is large enough to keep every wave busy with many
// pieces: the chunked regime's shape. In-process A/B at 1000 columns of 8.2 segments each (tools/probe_sum_chunks.py
// 0,4096 0 u8,i16 1000x4294912): u8 6.44 -> 6.76 TB/s dense and 5.09 -> 5.96 with validity, i16 with validity 5.72 -> 6.26
// (dense 6.84 -> 6.65: stays on segments); i32 / i64 / f64 gain nothing (6.55 either way) and stay on segments, as do few very
// long columns. The pieces are described ON the device: one thread per piece finds its column and writes a ShortCol;
// ColDesc::seg0 then counts pieces.
constexpr size_t piece_rows(size_t elem) { return ((size_t)1 << 16) / elem; }
constexpr size_t kMinPieces = 16384;  // 8 per wave at two workgroups per CU: below that, segments and workgroups

__global__ __launch_bounds__(kBlock) void expand_pieces_kernel(const ColDesc* __restrict__ cols, int n_cols, size_t n_pieces,
                                                               size_t rows_per_piece, size_t elem, ShortCol* __restrict__ out) {
    const size_t s = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (s >= n_pieces) return;
    int lo = 0, hi = n_cols - 1;  // the last column whose first piece is <= s (an empty column shares its seg0 with the next)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (cols[mid].seg0 <= s) lo = mid;
        else hi = mid - 1;
    }
    const ColDesc d = cols[lo];
    const size_t off = (s - d.seg0) * rows_per_piece;
    ShortCol e;
    e.data = (const char*)d.data + off * elem;
    e.len = d.len - off < rows_per_piece ? d.len - off : rows_per_piece;
    e.words = d.words;
    e.bit_off = d.bit_off + off;
    out[s] = e;
}

// short_table != nullptr: every column is a segment or less — a wave per column on the ShortCol table,
// partial c = column c; otherwise a workgroup per segment on the uploaded ColDesc table.
// total: one {sum, count} over all columns (they are the chunks of ONE logical column) instead of one per column;
// `partials2` then has room for 4096 partials.
template <typename T>
static void launch_columns(ma_ctx* ctx, const ColDesc* d, const ShortCol* short_table, size_t n_cols, size_t n_segs,
                           Partial* partials, bool is_signed, double* of, uint64_t* oi, uint64_t* oc, bool total = false,
                           Partial* partials2 = nullptr, double* olo = nullptr, ShortCol* expand_to = nullptr,
                           bool any_masked = false) {
    constexpr int UNROLL = sizeof(T) == 8 ? 8 : sizeof(T) == 1 ? 2 : 4;  // R * UNROLL <= 32 validity words per wave
    size_t n_short = n_cols;  // entries of the short table: columns, or the pieces of long columns (expand_to)
    if (expand_to) {
        hipLaunchKernelGGL(expand_pieces_kernel, dim3((unsigned)((n_segs + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, d,
                           (int)n_cols, n_segs, piece_rows(sizeof(T)), sizeof(T), expand_to);
        short_table = expand_to;
        n_short = n_segs;
    }
    size_t n_short_partials = n_short;  // short form: partial c = entry c, or (total) partial w = wave w
    bool direct = false;                // the waves wrote the per-column results themselves (column_waves_kernel's ColOut)
    if (short_table && (expand_to || !(tuning_variant(ctx) & 4096))) {
        // one or two waves per SIMD with ~8 KiB of loads each in flight AND a tile requested ahead (column_waves_kernel);
        // ctx->variant bits 1-3 / blocks_per_cu override the shape for sweeps (tools/probe_sum_chunks.py)
        // Swept at 60 000 x 8192 rows (profiles/r03_sweep_sum_chunks*.jsonl): ONE wave per SIMD with eight loads per tile — sixteen
        // in flight with the tile ahead — for 4- and 8-byte rows: i64 / f64 / i32 6.8-6.9 TB/s dense, 6.7-6.9 with validity, end
        // to end, against 7.2-7.3 for the plain sum of the same bytes; 122 000 chunks: 7.1 = 0.97 of the plain sum (two waves per
        // SIMD: 6.4-6.7; the round's first shape, eight workgroups per CU and no tile ahead: 6.3-6.4, i32 5.8). The 1- and 2-byte
        // types keep their shallower tiles (the validity words of a tile must fit one load instruction) on two waves per SIMD.
        // Sixteen loads per tile (thirty-two in flight) read no faster: i32 289.8 -> 291.7 us, i64 566.7 -> 571.2 (profiles/r06_column_waves.md).
        constexpr int U2 = sizeof(T) >= 4 ? 8 : UNROLL;  // the deeper of the two shapes of 4- and 8-byte types
        const int sel = (tuning_variant(ctx) >> 1) & 7;
        const bool deep = sizeof(T) >= 4 && sel != 2;
        // (f32 with validity — widened to f64, four double-double accumulators per load — is ALU-bound on one wave: 5.2 -> 5.9)
        const bool two = sizeof(T) < 4 || (std::is_same<T, float>::value && any_masked);
        const int bpc = ctx->blocks_per_cu > 0 ? ctx->blocks_per_cu : two ? 2 : 1;
        const int stagger = (tuning_variant(ctx) & 32768) ? 0 : 1;
        const int grid1 = grid_for(ctx, (n_short + kWaves - 1) / kWaves, bpc);
        if (total) n_short_partials = (size_t)grid1 * kWaves;
        // entry c = column c (no pieces): the waves write the columns' results themselves — no second launch
        ColOut out{};
        if (!total && !expand_to) {
            uint64_t* spare = (uint64_t*)partials;  // n_short x 32 bytes, unused in this form: three arrays of n_short words fit
            out.f64 = of ? of : (double*)spare;
            out.i64 = (oi && !std::is_same<typename AccOf<T>::type, DDAcc>::value) ? oi : spare + n_short;
            out.cnt = oc ? oc : spare + 2 * n_short;
            out.direct = 1;
            out.is_signed = is_signed ? 1 : 0;
            direct = true;
        }
        constexpr int U1 = sizeof(T) == 8 ? 4 : UNROLL;  // the shallower shape: what the 1- and 2-byte types run (U2 == U1 for them)
        if (deep || !MA_TUNING) {  // (shipped build: the deep shape for 4- and 8-byte rows; U2 == U1 for the narrow types)
            if (total) hipLaunchKernelGGL((column_waves_kernel<T, U2, true>), dim3(grid1), dim3(kBlock), 0, ctx->stream, short_table, n_short, partials, stagger, out);
            else hipLaunchKernelGGL((column_waves_kernel<T, U2, false>), dim3(grid1), dim3(kBlock), 0, ctx->stream, short_table, n_short, partials, stagger, out);
        } else if constexpr (MA_TUNING != 0) {
            if (total) hipLaunchKernelGGL((column_waves_kernel<T, U1, true>), dim3(grid1), dim3(kBlock), 0, ctx->stream, short_table, n_short, partials, stagger, out);
            else hipLaunchKernelGGL((column_waves_kernel<T, U1, false>), dim3(grid1), dim3(kBlock), 0, ctx->stream, short_table, n_short, partials, stagger, out);
        }
    } else if (short_table) {  // variant bit 4096: round 3's first shape (eight workgroups per CU, no tile ahead), for A/B
        if constexpr (MA_TUNING != 0) {
            const int grid1 = grid_for(ctx, (n_cols + kWaves - 1) / kWaves, ctx->blocks_per_cu > 0 ? ctx->blocks_per_cu : 8);
            hipLaunchKernelGGL((column_segments_kernel<T, UNROLL, true>), dim3(grid1), dim3(kBlock), 0, ctx->stream,
                               (const void*)short_table, (int)n_cols, n_cols, partials);
        }
    } else {
        // 1-byte rows: two loads in flight per lane, more waves (ma_reduce.hip)
        const int grid1 = grid_for(ctx, n_segs, ctx->blocks_per_cu > 0 ? ctx->blocks_per_cu : sizeof(T) == 1 ? 3 : 2);
        hipLaunchKernelGGL((column_segments_kernel<T, UNROLL, false>), dim3(grid1), dim3(kBlock), 0, ctx->stream, (const void*)d,
                           (int)n_cols, n_segs, partials);
    }
    if (total) {
        const Partial* src = partials;
        size_t n = short_table ? n_short_partials : n_segs;
        if (n > 4096) {  // level 1: up to 1024 workgroups x 4 waves, each wave a contiguous slice
            const size_t waves = 4096;
            const size_t per = (n + waves - 1) / waves;
            const size_t used = (n + per - 1) / per;
            hipLaunchKernelGGL((partial_fold_kernel<T>), dim3((unsigned)((used + kWaves - 1) / kWaves)), dim3(kBlock), 0, ctx->stream,
                               src, n, per, partials2);
            src = partials2;
            n = ((used + kWaves - 1) / kWaves) * kWaves;  // waves past `used` wrote empty partials
        }
        hipLaunchKernelGGL((total_fold_kernel<T>), dim3(1), dim3(kBlock), 0, ctx->stream, src, n, is_signed ? 1 : 0, of, oi, oc, olo);
        return;
    }
    if (direct) return;
    const int grid2 = grid_for(ctx, (n_cols + kWaves - 1) / kWaves, 8);
    hipLaunchKernelGGL((column_fold_kernel<T>), dim3(grid2), dim3(kBlock), 0, ctx->stream, d, (int)n_cols, n_segs,
                       (const Partial*)partials, is_signed ? 1 : 0, of, oi, oc, (short_table && !expand_to) ? 1 : 0);
}

}  // namespace ma

using namespace ma;

namespace ma {
ma_status sum_fused_impl(ma_ctx* ctx, size_t n_cols, const ma_fused_column* cols, uint64_t* stamp, uint64_t stamp_value,
                         bool as_partials, uint64_t* early_stamp);  // ma_reduce_fused.hip
}

static ma_status sum_columns_impl(ma_ctx* ctx, int32_t format_code, size_t n_cols, const void* const* col_data,
                                  const size_t* col_lens, const uint8_t* const* col_masks, const size_t* col_mask_offsets,
                                  double* out_sums_f64, int64_t* out_sums_i64, uint64_t* out_valid_counts, bool total,
                                  double* out_lo = nullptr) {
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    size_t elem = 0;
    switch (format_code) {
        case 'c': case 'C': elem = 1; break;
        case 's': case 'S': elem = 2; break;
        case 'i': case 'I': case 'f': elem = 4; break;
        case 'l': case 'L': case 'g': elem = 8; break;
        default:
            set_error("unsupported element format '%c' (numeric primitives only)", (char)format_code);
            return MA_ERR_UNSUPPORTED;
    }
    if (n_cols == 0 && !total) return MA_OK;
    MA_REQUIRE(n_cols < ((size_t)1 << 30), MA_ERR_INVALID_ARGUMENT, "too many columns");
    MA_REQUIRE(n_cols == 0 || (col_data != nullptr && col_lens != nullptr), MA_ERR_INVALID_ARGUMENT, "NULL column table");
    // One pass over the caller's table for everything that is decided per column in front of the launch (a chunked column at
    // RechunkStrategy::Auto is 122 000 "columns" per 10^9 rows, and a 32-KB chunk is scanned in 4.7 ns: the host's time per
    // column is on the critical path of i32 and narrower rows — profiles/r06_column_waves.md).
    size_t longest = 0;
    {
        const uintptr_t mis = (uintptr_t)elem - 1;  // elem is 1, 2, 4 or 8
        uintptr_t bad_align = 0;
        size_t first_null = n_cols;
        for (size_t i = 0; i < n_cols; ++i) {
            const size_t len = col_lens[i];
            const uintptr_t a = (uintptr_t)col_data[i];
            bad_align |= a & mis;
            if (len != 0 && a == 0 && first_null == n_cols) first_null = i;
            longest = len > longest ? len : longest;
        }
        MA_REQUIRE(first_null == n_cols, MA_ERR_INVALID_ARGUMENT, "column %zu data is NULL", first_null);
        if (bad_align)
            for (size_t i = 0; i < n_cols; ++i)
                MA_REQUIRE(((uintptr_t)col_data[i] & mis) == 0, MA_ERR_INVALID_ARGUMENT, "column %zu is misaligned", i);
    }
    MA_ENTER(ctx);
    MA_NO_CAPTURE(ctx, "ma_sum_columns (descriptor upload)");
    MA_HIP(hipSetDevice(ctx->device));
    CallScope scope(ctx);
    if (n_cols == 0) {  // ma_sum_chunks of an empty chunk list: {0, 0}
        void *zf = nullptr, *zi = nullptr, *zc = nullptr, *zl = nullptr;
        MA_TRY(scope.out(out_sums_f64, 8, &zf));
        MA_TRY(scope.out(out_sums_i64, 8, &zi));
        MA_TRY(scope.out(out_valid_counts, 8, &zc));
        MA_TRY(scope.out(out_lo, 8, &zl));
        if (format_code == 'f' || format_code == 'g')
            hipLaunchKernelGGL((total_fold_kernel<double>), dim3(1), dim3(kBlock), 0, ctx->stream, (const Partial*)nullptr, (size_t)0, 1,
                               (double*)zf, (uint64_t*)zi, (uint64_t*)zc, (double*)zl);
        else
            hipLaunchKernelGGL((total_fold_kernel<int64_t>), dim3(1), dim3(kBlock), 0, ctx->stream, (const Partial*)nullptr, (size_t)0, 1,
                               (double*)zf, (uint64_t*)zi, (uint64_t*)zc, (double*)zl);
        MA_HIP(hipGetLastError());
        return end_call(ctx, scope);
    }
    // A FEW LONG dense 8-byte columns on the device (the batches of a SuperTable column, a table of a few wide columns): the
    // fused scan of ma_reduce_fused.hip, four columns per launch — the single-column sum's shape (paced loads, one
    // cross-workgroup hand-off per launch) instead of a workgroup per 64-Ki-row segment: 8 x 67 M rows 6.85 -> 7.03 TB/s, 8 x 125 M
    // rows 6.85 -> 7.15.
    // Each launch leaves one Partial per column; the folds below are the general path's. variant bit 16384: general path (A/B);
    // bit 65536: the fused scan whatever the total size (tests of the multi-launch form at sizes a CPU check can follow).
    if (elem == 8 && n_cols <= 16 && !(form_variant(ctx) & 16384)) {
        bool few_long = true;
        size_t total_rows = 0;
        for (size_t i = 0; i < n_cols && few_long; ++i) {
            few_long = col_lens[i] >= ((size_t)1 << 21) && pointer_kind(col_data[i]) == kDevice;
            if (few_long && col_masks && col_masks[i]) few_long = pointer_kind(col_masks[i]) == kDevice;
            total_rows += col_lens[i];
        }
        // a fused launch costs ~4.5 us beyond its bytes and reads them 5 % faster than the two launches of the segment path
        // (~6 us): one launch (<= 4 columns) always pays, more only from ~384 MiB per launch
        const size_t launches = (n_cols + MA_FUSED_MAX_COLUMNS - 1) / MA_FUSED_MAX_COLUMNS;
        if (launches > 1 && total_rows * 8 < launches * ((size_t)384 << 20) && !(form_variant(ctx) & 65536)) few_long = false;  // bit 65536: tests
        if (few_long) {
            void *of = nullptr, *oi = nullptr, *oc = nullptr, *olo = nullptr;
            const size_t n_out = total ? 1 : n_cols;
            MA_TRY(scope.out(out_sums_f64, n_out * 8, &of));
            MA_TRY(scope.out(out_sums_i64, n_out * 8, &oi));
            MA_TRY(scope.out(out_valid_counts, n_out * 8, &oc));
            if (total) MA_TRY(scope.out(out_lo, 8, &olo));
            void* scratch = nullptr;
            MA_TRY(ctx_scratch(ctx, sizeof(Partial) * n_cols, &scratch));
            Partial* partials = (Partial*)scratch;
            {
                NoSync enqueue_only;  // the composed launches are waited for once, by end_call below
                for (size_t c0 = 0; c0 < n_cols; c0 += MA_FUSED_MAX_COLUMNS) {
                    const size_t k = n_cols - c0 < MA_FUSED_MAX_COLUMNS ? n_cols - c0 : MA_FUSED_MAX_COLUMNS;
                    ma_fused_column fc[MA_FUSED_MAX_COLUMNS] = {};
                    for (size_t j = 0; j < k; ++j) {
                        const size_t i = c0 + j;
                        fc[j].data = col_data[i];
                        fc[j].n = col_lens[i];
                        fc[j].mask_bits = col_masks ? col_masks[i] : nullptr;
                        fc[j].mask_bit_offset = col_mask_offsets ? col_mask_offsets[i] : 0;
                        fc[j].null_count = -1;
                        fc[j].format_code = format_code;
                        fc[j].out = (uint64_t*)&partials[i];
                    }
                    MA_TRY(sum_fused_impl(ctx, k, fc, nullptr, 0, true, nullptr));
                }
            }
            const bool is_signed = format_code != 'L';
            if (total) {
                if (format_code == 'g')
                    hipLaunchKernelGGL((total_fold_kernel<double>), dim3(1), dim3(kBlock), 0, ctx->stream, (const Partial*)partials, n_cols,
                                       1, (double*)of, (uint64_t*)oi, (uint64_t*)oc, (double*)olo);
                else
                    hipLaunchKernelGGL((total_fold_kernel<int64_t>), dim3(1), dim3(kBlock), 0, ctx->stream, (const Partial*)partials, n_cols,
                                       is_signed ? 1 : 0, (double*)of, (uint64_t*)oi, (uint64_t*)oc, (double*)olo);
            } else {
                if (format_code == 'g')
                    hipLaunchKernelGGL((column_fold_kernel<double>), dim3(1), dim3(kBlock), 0, ctx->stream, (const ColDesc*)nullptr, (int)n_cols,
                                       n_cols, (const Partial*)partials, 1, (double*)of, (uint64_t*)oi, (uint64_t*)oc, 1);
                else
                    hipLaunchKernelGGL((column_fold_kernel<int64_t>), dim3(1), dim3(kBlock), 0, ctx->stream, (const ColDesc*)nullptr, (int)n_cols,
                                       n_cols, (const Partial*)partials, is_signed ? 1 : 0, (double*)of, (uint64_t*)oi, (uint64_t*)oc, 1);
            }
            MA_HIP(hipGetLastError());
            return end_call(ctx, scope);
        }
    }
    // Every column a segment or less (a chunked column handed over chunk by chunk): the short form (ShortCol).
    const bool all_short = n_cols >= 256 && longest <= seg_rows(elem);  // below 256 the table copy is a few microseconds and the segment form is as good
    // Long columns, enough of them for every wave to take many pieces: described per column, cut into pieces on the device and
    // summed by the same wave kernel (expand_pieces_kernel). variant bit 4096: segments and workgroups, for A/B.
    size_t n_pieces = 0, n_long_segs = 0;
    // variant bit 8192, A/B: pieces for 4- and 8-byte rows and for few very long columns too (1000 x 537 k rows: i64 / f64 6.6 TB/s
    // either way, i32 6.47 -> 6.2 dense, 5.87 -> 6.08 with validity; 8 x 67 M rows: the same picture)
    const bool any_width = (tuning_variant(ctx) & 8192) != 0;
    if (!all_short && (elem <= 2 || any_width) && !(tuning_variant(ctx) & 4096))
        for (size_t i = 0; i < n_cols; ++i) {
            n_pieces += (col_lens[i] + piece_rows(elem) - 1) / piece_rows(elem);
            n_long_segs += (col_lens[i] + seg_rows(elem) - 1) / seg_rows(elem);
        }
    const bool expand = (any_width || ((elem == 1 || (elem == 2 && col_masks != nullptr)) && n_long_segs < 32 * n_cols)) &&
                        n_pieces >= kMinPieces && n_pieces < ((size_t)1 << 30);
    const size_t rows_per_seg = expand ? piece_rows(elem) : seg_rows(elem);
    ColDesc* desc = nullptr;  // either table is built in the context's pinned staging buffer: no second copy of 60 000 entries
    ShortCol* sdesc = nullptr;
    // The short table goes to the device on the context's upload stream, in pieces while this loop still writes the rest and beside
    // the kernels of the call before (TableUpload, ma_common.hpp): read in place — a PCIe read per chunk, 200 M/s under an i32 scan —
    // the same kernel ran 288 us or 310-370, by the staging slot the table happened to sit in (profiles/r06_column_waves.md).
    // A table of a few thousand columns is still read where it was built: its kernel is over before a copy has been waited for.
    TableUpload upload(ctx);
    constexpr size_t kUploadPiece = 16384;  // columns per piece (512 KiB: ~10 us of copy, ~4 us of host time)
    const bool uploaded = all_short && n_cols >= kUploadPiece / 2;
    if (all_short) {
        MA_TRY(table_begin(ctx, sizeof(ShortCol) * n_cols, (void**)&sdesc));
        if (uploaded) MA_TRY(upload.begin(sdesc, sizeof(ShortCol) * n_cols));
    } else {
        MA_TRY(table_begin(ctx, sizeof(ColDesc) * n_cols, (void**)&desc));
    }
    size_t n_segs = 0;
    // the chunks of a chunked column (one "column" each: 122 000 per 10^9 rows at RechunkStrategy::Auto) run through a few
    // allocations: each role remembers the device range of its last pointer — two compares instead of a classification
    DeviceRange data_role, mask_role;
    bool any_masked = false;
    for (size_t i = 0; i < n_cols; ++i) {
        const void* data = col_data[i];
        if (!data_role.holds(data)) {
            MA_TRY(scope.in(col_data[i], col_lens[i] * elem, &data));
            if (col_lens[i]) data_role.learn(col_data[i]);
        }
        const uint64_t* words = nullptr;
        size_t bit_off = 0;
        if (col_masks && col_masks[i] && col_lens[i]) {
            const size_t mo = col_mask_offsets ? col_mask_offsets[i] : 0;
            if (mask_role.holds(col_masks[i])) {
                const uintptr_t addr = (uintptr_t)col_masks[i], base = addr & ~(uintptr_t)7;  // CallScope::in_mask's re-basing
                words = (const uint64_t*)base;
                bit_off = mo + (size_t)(addr - base) * 8;
            } else {
                MA_TRY(scope.in_mask(col_masks[i], mo, col_lens[i], &words, &bit_off));
                mask_role.learn(col_masks[i]);
            }
        }
        any_masked |= words != nullptr;
        if (all_short) {
            sdesc[i] = ShortCol{data, col_lens[i], words, bit_off};
            if (uploaded && (i + 1) % kUploadPiece == 0 && n_cols - (i + 1) >= kUploadPiece / 2) MA_TRY(upload.push(sizeof(ShortCol) * (i + 1)));
        } else {
            ColDesc& d = desc[i];
            d.data = data;
            d.len = col_lens[i];
            d.words = words;
            d.bit_off = bit_off;
            d.last_word = words ? (bit_off + d.len - 1) >> 6 : 0;
            d.seg0 = n_segs;
            n_segs += d.len ? (d.len + rows_per_seg - 1) / rows_per_seg : 0;  // an empty column has no segment: its fold is {0, 0}
        }
    }
    void *of = nullptr, *oi = nullptr, *oc = nullptr;
    const size_t n_out = total ? 1 : n_cols;
    MA_TRY(scope.out(out_sums_f64, n_out * 8, &of));
    MA_TRY(scope.out(out_sums_i64, n_out * 8, &oi));
    MA_TRY(scope.out(out_valid_counts, n_out * 8, &oc));
    void* olo = nullptr;
    if (total) MA_TRY(scope.out(out_lo, 8, &olo));

    // descriptors (segment form) + the pieces' table + partials in one scratch allocation
    const size_t n_partials = all_short ? n_cols + kWaves : (n_segs ? n_segs : 1) + kWaves;  // + kWaves: a total's partial per WAVE, grid rounded up
    const size_t col_bytes = all_short ? 0 : ((sizeof(ColDesc) * n_cols + 255) / 256) * 256;
    const size_t desc_bytes = col_bytes + (expand ? ((sizeof(ShortCol) * n_segs + 255) / 256) * 256 : 0);
    void* scratch = nullptr;
    MA_TRY(ctx_scratch(ctx, desc_bytes + sizeof(Partial) * (n_partials + (total ? 4096 : 0)), &scratch));
    const ColDesc* d = nullptr;
    const ShortCol* sd = nullptr;
    ShortCol* pieces = expand ? (ShortCol*)((char*)scratch + col_bytes) : nullptr;
    TableSlotGuard guard(ctx);  // a table read in place: its slot is released behind the launches on every way out
    if (all_short) {
        const void* alias = nullptr;
        if (uploaded) MA_TRY(upload.finish(&alias));  // (its last piece, waited for)
        else MA_TRY(table_commit_mapped(ctx, sdesc, &alias, &guard.slot));
        sd = (const ShortCol*)alias;
    } else {
        MA_TRY(table_commit(ctx, desc, sizeof(ColDesc) * n_cols, scratch));
        d = (const ColDesc*)scratch;
    }
    Partial* partials = (Partial*)((char*)scratch + desc_bytes);
    Partial* partials2 = partials + n_partials;  // level-1 results of the total fold
    switch (format_code) {
        case 'c': launch_columns<int8_t>(ctx, d, sd, n_cols, n_segs, partials, true, (double*)of, (uint64_t*)oi, (uint64_t*)oc, total, partials2, (double*)olo, pieces, any_masked); break;
        case 'C': launch_columns<uint8_t>(ctx, d, sd, n_cols, n_segs, partials, false, (double*)of, (uint64_t*)oi, (uint64_t*)oc, total, partials2, (double*)olo, pieces, any_masked); break;
        case 's': launch_columns<int16_t>(ctx, d, sd, n_cols, n_segs, partials, true, (double*)of, (uint64_t*)oi, (uint64_t*)oc, total, partials2, (double*)olo, pieces, any_masked); break;
        case 'S': launch_columns<uint16_t>(ctx, d, sd, n_cols, n_segs, partials, false, (double*)of, (uint64_t*)oi, (uint64_t*)oc, total, partials2, (double*)olo, pieces, any_masked); break;
        case 'i': launch_columns<int32_t>(ctx, d, sd, n_cols, n_segs, partials, true, (double*)of, (uint64_t*)oi, (uint64_t*)oc, total, partials2, (double*)olo, pieces, any_masked); break;
        case 'I': launch_columns<uint32_t>(ctx, d, sd, n_cols, n_segs, partials, false, (double*)of, (uint64_t*)oi, (uint64_t*)oc, total, partials2, (double*)olo, pieces, any_masked); break;
        case 'l': launch_columns<int64_t>(ctx, d, sd, n_cols, n_segs, partials, true, (double*)of, (uint64_t*)oi, (uint64_t*)oc, total, partials2, (double*)olo, pieces, any_masked); break;
        case 'L': launch_columns<uint64_t>(ctx, d, sd, n_cols, n_segs, partials, false, (double*)of, (uint64_t*)oi, (uint64_t*)oc, total, partials2, (double*)olo, pieces, any_masked); break;
        case 'f': launch_columns<float>(ctx, d, sd, n_cols, n_segs, partials, true, (double*)of, (uint64_t*)oi, (uint64_t*)oc, total, partials2, (double*)olo, pieces, any_masked); break;
        default: launch_columns<double>(ctx, d, sd, n_cols, n_segs, partials, true, (double*)of, (uint64_t*)oi, (uint64_t*)oc, total, partials2, (double*)olo, pieces, any_masked); break;
    }
    MA_HIP(hipGetLastError());
    return end_call(ctx, scope);
}

extern "C" ma_status ma_sum_columns(ma_ctx* ctx, int32_t format_code, size_t n_cols, const void* const* col_data,
                                    const size_t* col_lens, const uint8_t* const* col_masks,
                                    const size_t* col_mask_offsets, double* out_sums_f64, int64_t* out_sums_i64,
                                    uint64_t* out_valid_counts) {
    return sum_columns_impl(ctx, format_code, n_cols, col_data, col_lens, col_masks, col_mask_offsets, out_sums_f64, out_sums_i64,
                            out_valid_counts, false);
}

// ma_sum_chunks with the float total as a (hi, lo) pair: what a group member contributes to the exchange (ma_group.hip).
namespace ma {
ma_status sum_chunks_dd(ma_ctx* ctx, int32_t format_code, size_t n_chunks, const void* const* chunk_data, const size_t* chunk_lens,
                        const uint8_t* const* chunk_masks, const size_t* chunk_mask_offsets, double* out_hi, double* out_lo,
                        int64_t* out_sum_i64, uint64_t* out_valid_count) {
    return sum_columns_impl(ctx, format_code, n_chunks, chunk_data, chunk_lens, chunk_masks, chunk_mask_offsets, out_hi, out_sum_i64,
                            out_valid_count, true, out_lo);
}
}  // namespace ma

// The sum of ONE column held as a list of chunks (a SuperArray's chunks, one column of a SuperTable's batches): the same two
// passes, then the partials of ALL chunks folded into one {sum, count} — the double-double fold keeps the f64 total within
// 1 ULP of the exactly rounded sum, which a host-side addition of per-chunk rounded sums would not.
extern "C" ma_status ma_sum_chunks(ma_ctx* ctx, int32_t format_code, size_t n_chunks, const void* const* chunk_data,
                                   const size_t* chunk_lens, const uint8_t* const* chunk_masks, const size_t* chunk_mask_offsets,
                                   double* out_sum_f64, int64_t* out_sum_i64, uint64_t* out_valid_count) {
    return sum_columns_impl(ctx, format_code, n_chunks, chunk_data, chunk_lens, chunk_masks, chunk_mask_offsets, out_sum_f64,
                            out_sum_i64, out_valid_count, true);
}

// ------------------------------------------------------------------------------------------------
// Fold of per-rank reduction records — the step after the all-gather of a row-chunk partitioned reduction
// (minarrow_amd/parallel.py; the Rayon `.sum()` over per-chunk partials, benches/benchmark_parallel_simd.rs:87).
// record r = 8 x u64: [0] integer sum, [1] integer valid count, [2] f64 hi bits, [3] f64 lo bits, [4] float valid
// count. Folded strictly in record order by ONE lane, so every rank computes bit-identical finals:
// wrapping u64 adds; (hi, lo) pairs by Knuth two-sum, lo += e + l, final = hi + lo when both are finite.
// out = 4 x u64: [0] integer sum, [1] integer count, [2] f64 sum bits, [3] float count.
// ------------------------------------------------------------------------------------------------
namespace ma {
// blockIdx.x = column: its records start kRecordWords further on, its finals 4 words further on.
__global__ void fold_records_kernel(const uint64_t* __restrict__ rec, size_t n, size_t stride, uint64_t* __restrict__ out) {
    if (threadIdx.x != 0) return;
    rec += (size_t)blockIdx.x * kRecordWords;
    out += (size_t)blockIdx.x * 4;
    uint64_t isum = 0, icnt = 0, fcnt = 0;
    double hi = 0.0, lo = 0.0;
    for (size_t r = 0; r < n; ++r) {
        const uint64_t* p = rec + r * stride;
        isum += p[0];
        icnt += p[1];
        const double h = __longlong_as_double((long long)p[2]), l = __longlong_as_double((long long)p[3]);
        const double s = hi + h;
        const double bp = s - hi;
        const double e = (hi - (s - bp)) + (h - bp);
        hi = s;
        lo += e + l;
        fcnt += p[4];
    }
    const double total = (isfinite(hi) && isfinite(lo)) ? hi + lo : hi;
    out[0] = isum;
    out[1] = icnt;
    out[2] = (uint64_t)__double_as_longlong(total);
    out[3] = fcnt;
}

ma_status enqueue_fold_columns(ma_ctx* ctx, const uint64_t* rec, size_t n_records, size_t stride_words, size_t n_columns,
                               uint64_t* out) {
    if (n_columns == 0) return MA_OK;
    hipLaunchKernelGGL(fold_records_kernel, dim3((unsigned)n_columns), dim3(64), 0, ctx->stream, rec, n_records, stride_words, out);
    MA_HIP(hipGetLastError());
    return MA_OK;
}
}  // namespace ma

extern "C" ma_status ma_fold_sum_records(ma_ctx* ctx, const uint64_t* records, size_t n_records, size_t stride_words,
                                         uint64_t* out4) {
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    MA_REQUIRE(records != nullptr && out4 != nullptr, MA_ERR_INVALID_ARGUMENT, "NULL buffer");
    MA_REQUIRE(stride_words >= 5, MA_ERR_INVALID_ARGUMENT, "a record has 5 words; stride_words = %zu", stride_words);
    MA_REQUIRE(((uintptr_t)records & 7) == 0 && ((uintptr_t)out4 & 7) == 0, MA_ERR_INVALID_ARGUMENT, "misaligned buffer");
    MA_ENTER(ctx);
    MA_HIP(hipSetDevice(ctx->device));
    CallScope scope(ctx);
    const void* r = nullptr;
    void* o = nullptr;
    MA_TRY(scope.in(records, n_records * stride_words * 8, &r));
    MA_TRY(scope.out(out4, 32, &o));
    MA_TRY(enqueue_fold_columns(ctx, (const uint64_t*)r, n_records, stride_words, 1, (uint64_t*)o));
    return end_call(ctx, scope);
}
