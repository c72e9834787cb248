// ma_sum_fused: the sums of up to MA_FUSED_MAX_COLUMNS long 8-byte columns (i64 / u64 / f64, each dense or Bitmask-gated) in
// ONE launch — a per-column reduce of a table's columns (BASELINE config 5: src/structs/chunked/super_table.rs:657-743 +
// a reduce per column), or the two loops of the reference's sum bench over one row chunk
// (benches/benchmark_parallel_simd.rs:99-125, `rayon_simd_sum_i64` then `rayon_simd_sum_f64`).
//
// Why one launch. tools/probe_epilogue.hip (profiles/r04_probe_epilogue.jsonl) splits a sum launch's fixed cost on MI355X:
// ~1.5 us until the first tile's data has arrived (the ramp), ~1.8 us from the last workgroup's last row to the final store
// (wave reduce -> LDS -> partial published with sc1 stores and drained -> ticket(s) -> the last arrival loads and folds every
// partial), and the next dependent dispatch starts behind that. A 10^9-row column partitioned over 8 GPUs leaves each GPU
// 137 us of scan per column: two launches pay the 3.3 us twice per step, one pays it once — the columns' tiles run through
// the same workgroups back to back with no drain between them (tile index space = the columns' tiles concatenated,
// tile g -> workgroup g mod grid), every workgroup keeps one accumulator set PER COLUMN in registers, and the epilogue
// publishes and folds all columns' partials behind ONE ticket.
//
// Semantics per column are those of ma_i64_sum / ma_u64_sum / ma_f64_sum_dd (ma_reduce.hip): wrapping u64 accumulation for
// integers (bit-exact in any order), double-double (Knuth two-sum) for f64 — within 1 ULP of the exactly rounded sum and
// bit-reproducible for a fixed launch shape —, valid count = popcount of the window's validity bits (or n when dense).
#include <limits>

#include <type_traits>

#include "ma_acc.hpp"
#include "ma_device.hpp"

namespace ma {

constexpr int kFusedMax = MA_FUSED_MAX_COLUMNS;

struct FusedCol {
    const void* data;       // element pointer of the window (8-byte elements)
    size_t n;               // rows
    size_t head;            // rows in front of the first 16-byte boundary (0 or 1)
    size_t n_tiles;         // full workgroup tiles behind `head`
    size_t tile0;           // index of this column's tile 0 in the launch's concatenated tile space
    const uint64_t* words;  // validity words (8-byte aligned base) or nullptr = dense
    size_t bit_off;         // bit index of row 0 relative to `words`
    size_t last_word;       // index of the last word that holds a window bit
    uint64_t* out;          // integers: out[0] = sum, out[1] = valid count; f64: out[0], out[1] = (hi, lo), out[2] = valid count
    int is_float;
    int as_partial;         // internal callers (ma_sum_columns): `out` is a ma::Partial {a, b, count, 0} whatever the type
};

struct FusedArgs {
    FusedCol col[kFusedMax];
    int n_cols;
    size_t total_tiles;     // all columns' full tiles
    Partial* partials;      // column c's partial of workgroup b at [c * gridDim.x + b]
    unsigned int* ticket;
    uint64_t* done_word;    // pinned word stamped after the results (synchronous call that polls), or nullptr
    uint64_t done_seq;
    // "This launch has begun to drain": the word the NEXT scan, on another stream, waits for before it starts (done_seq is stored).
    // WHEN it is stored decides whether consecutive scans overlap by their tails only or run away into overlapping for most of
    // their length (a scan that starts early slows the stragglers of the one before, whose own successor then starts earlier
    // still; from 10^8 rows on that costs what the overlap gains). early_mode, over 5 boxes x 4 processes x 2 launch modes at
    // 125 M rows per column (tools/run_early_quarters.sh; step on one scan stream: 0.286-0.293 ms; "ran away" = span of a scan
    // between its marks >= 0.43 ms instead of ~0.30):
    //   0  the first workgroup that has scanned its rows                  0.271-0.300, ran away in 16 of 24 processes
    //   1-3  the arrival that completes 1/4, 1/2, 3/4 of a ticket shard   (3/4:) 0.271-0.300, ran away in 8 of 36
    //   4  the first whole shard (the workgroups of one XCD) has arrived  0.273-0.280, ran away in 1 of 48
    //   5  two whole shards have arrived — the default                    0.274-0.279, never in 36
    //   6  four / 7: six whole shards                                      0.275-0.281 / 0.278-0.280, never
    // Grids of up to 96 workgroups arrive on one ticket: mode 0 there (short scans; free-running overlap is a gain below 10^8 rows).
    uint64_t* early_word;
    unsigned early_mode;
};

// One accumulator per (column, row-of-a-load): two 64-bit words that are a wrapping integer sum (a) or a double-double
// (a = hi, b = lo) depending on the column's kind — a wave-uniform runtime choice, so one register set serves both.
struct Acc2 {
    uint64_t a, b;
    __device__ __forceinline__ void init() { a = 0; b = 0; }
    __device__ __forceinline__ void add_int(uint64_t x) { a += x; }
    __device__ __forceinline__ void add_f64(double v) {
        double hi = __longlong_as_double((long long)a), lo = __longlong_as_double((long long)b);
        const double t = hi + v;
        const double bp = t - hi;
        const double e = (hi - (t - bp)) + (v - bp);
        a = (uint64_t)__double_as_longlong(t);
        b = (uint64_t)__double_as_longlong(lo + e);
    }
    __device__ __forceinline__ void merge(const Acc2& o, bool is_float) {
        if (is_float) {
            DDAcc x, y;
            x.from_words(a, b);
            y.from_words(o.a, o.b);
            x.merge(y);
            a = (uint64_t)__double_as_longlong(x.hi);
            b = (uint64_t)__double_as_longlong(x.lo);
        } else {
            a += o.a;
        }
    }
    __device__ __forceinline__ Acc2 shfl_down(int off) const {
        Acc2 o;
        o.a = (uint64_t)__shfl_down((unsigned long long)a, off, 64);
        o.b = (uint64_t)__shfl_down((unsigned long long)b, off, 64);
        return o;
    }
};

constexpr unsigned kFTicketShards = 8;
constexpr unsigned kFTicketShardWord0 = 64;   // same layout as ma_reduce.hip's arrival counters (the context's zeroed block)
constexpr unsigned kFTicketShardStride = 16;
constexpr unsigned kFShardFrom = 96;

template <int UNROLL, bool ANY_MASKED, int PACE>
__global__ __launch_bounds__(kBlock) void sum_fused_kernel(FusedArgs a) {
    typedef unsigned long long V __attribute__((ext_vector_type(2)));  // 16 bytes = two 8-byte rows
    constexpr int R = 2;
    constexpr int WPT = R * UNROLL;
    constexpr size_t WAVE_ROWS = (size_t)64 * R * UNROLL;
    constexpr size_t TILE_ROWS = WAVE_ROWS * kWaves;
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned G = gridDim.x, b = blockIdx.x;

    Acc2 acc[kFusedMax][R];
    uint64_t cnt[kFusedMax];
#pragma unroll
    for (int c = 0; c < kFusedMax; ++c) {
        acc[c][0].init();
        acc[c][1].init();
        cnt[c] = 0;
    }

#pragma unroll
    for (int c = 0; c < kFusedMax; ++c) {
        if (c >= a.n_cols) continue;
        const FusedCol& col = a.col[c];
        const bool is_float = col.is_float != 0;
        const bool masked = ANY_MASKED && col.words != nullptr;
        const uint64_t* __restrict__ data = (const uint64_t*)col.data;
        // this workgroup's first tile of the column: the smallest t >= 0 with (tile0 + t) mod G == b
        const size_t first = (size_t)((b + G - (unsigned)(col.tile0 % G)) % G);
        // one tile's rows into the column's accumulators; `aw` = the run's validity words (bit j of word k = row 64 k + j)
        auto consume = [&](const V (&v)[UNROLL], uint64_t aw) {
            if (is_float) {
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    const unsigned bits = (ANY_MASKED && masked) ? lane_bits<R>(aw, u, lane) : 3u;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const double x = __longlong_as_double((long long)v[u][r]);
                        acc[c][r].add_f64(((bits >> r) & 1u) ? x : 0.0);
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    const unsigned bits = (ANY_MASKED && masked) ? lane_bits<R>(aw, u, lane) : 3u;
#pragma unroll
                    for (int r = 0; r < R; ++r) acc[c][r].add_int(((bits >> r) & 1u) ? (uint64_t)v[u][r] : 0);
                }
            }
        };
        if constexpr (ANY_MASKED) {
            // Validity work sits between a tile's loads and the next tile's: with four loads per lane and nothing requested
            // ahead, the wave's bytes in flight drop to zero once per tile (two such columns of 125 M rows: 311 us against
            // 2 x 144 for the single-column kernels). So the NEXT tile's rows and raw validity words are requested before
            // this tile is consumed, the way ma_reduce_batch.hip's wave kernel does it and under the same rules (learnt
            // there): a request is ALWAYS five loads — with nothing left to request every lane reads the first bytes of
            // `partials` instead, and a dense column's "validity word" comes from there too — because loads issued on some
            // paths only make the compiler's in-order wait counts assume the shortest queue; two register sets swap roles
            // (a copy from "next" to "current" would wait for what it copies); the funnel shift of the words waits until
            // they are wanted (finish_run_words).
            const size_t n_mine = first < col.n_tiles ? (col.n_tiles - first + G - 1) / G : 0;  // this workgroup's tiles
            auto issue = [&](size_t k, V (&v)[UNROLL], uint64_t& raw, size_t& row0) {
                const bool real = k < n_mine;
                row0 = col.head + (real ? first + k * G : 0) * TILE_ROWS + (size_t)wave * WAVE_ROWS;
                const V* __restrict__ p = real ? (const V*)(data + row0) + lane : (const V*)a.partials;
                const size_t stride = real ? 64 : 0;
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) v[u] = load16<V, true>(p + (size_t)u * stride);
                const bool m = real && masked;
                // lane l <= WPT holds run word l (the last one only feeds the funnel shift); clamped, not skipped, past the
                // window's last word — a full tile never needs a word behind it
                size_t idx = ((col.bit_off + row0) >> 6) + (lane < (unsigned)WPT ? lane : (unsigned)WPT);
                idx = idx < col.last_word ? idx : col.last_word;
                raw = as_global(m ? col.words : (const uint64_t*)a.partials)[m ? idx : 0];
            };
            // the loop once per (kind, validity) — both are uniform over the column, and left inside the loop they cost a
            // dozen scalar branches per tile
            auto run = [&](auto float_c, auto masked_c) {
                constexpr bool F = decltype(float_c)::value, M = decltype(masked_c)::value;
                auto use = [&](const V (&v)[UNROLL], uint64_t raw, size_t row0) {
                    uint64_t aw = ~(uint64_t)0;
                    if constexpr (M) {
                        aw = finish_run_words(raw, col.bit_off + row0);
                        if (lane < (unsigned)WPT) cnt[c] += (uint64_t)__popcll(aw);
                    }
#pragma unroll
                    for (int u = 0; u < UNROLL; ++u) {
                        const unsigned bits = M ? lane_bits<R>(aw, u, lane) : 3u;
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            if constexpr (F) {
                                const double x = __longlong_as_double((long long)v[u][r]);
                                acc[c][r].add_f64(((bits >> r) & 1u) ? x : 0.0);
                            } else {
                                acc[c][r].add_int(((bits >> r) & 1u) ? (uint64_t)v[u][r] : 0);
                            }
                        }
                    }
                };
                V va[UNROLL], vb[UNROLL];
                uint64_t ra, rb;
                size_t row_a, row_b;
                issue(0, va, ra, row_a);
                for (size_t k = 0; k < n_mine; k += 2) {
                    issue(k + 1, vb, rb, row_b);
                    use(va, ra, row_a);
                    issue(k + 2, va, ra, row_a);
                    if (k + 1 < n_mine) use(vb, rb, row_b);
                }
            };
            if (is_float) {
                if (masked) run(std::true_type{}, std::true_type{});
                else run(std::true_type{}, std::false_type{});
            } else {
                if (masked) run(std::false_type{}, std::true_type{});
                else run(std::false_type{}, std::false_type{});
            }
        } else {
            for (size_t t = first; t < col.n_tiles; t += G) {
                const size_t row0 = col.head + t * TILE_ROWS + (size_t)wave * WAVE_ROWS;
                const V* __restrict__ p = (const V*)(data + row0) + lane;
                V v[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    v[u] = load16<V, true>(p + (size_t)u * 64);
                    if (u + 1 < UNROLL) pace_loads<PACE>();
                }
                if constexpr (PACE > 0) __builtin_amdgcn_sched_barrier(0);
                consume(v, ~(uint64_t)0);
            }
        }
        if (!masked && b == 0 && tid == 0) cnt[c] = col.n;  // dense: every row is valid; credited once
    }
    // Ragged rows (a column's unaligned head, whatever follows its last full tile), behind ALL columns' tiles so that no
    // workgroup stops between two columns for them, and on workgroups that have a tile less than the others: the concatenated
    // tile space ends at workgroup total_tiles mod G - 1, column c's ragged rows go to workgroup (total_tiles + c) mod G.
#pragma unroll
    for (int c = 0; c < kFusedMax; ++c) {
        if (c >= a.n_cols) continue;
        const FusedCol& col = a.col[c];
        if (b != (unsigned)((a.total_tiles + (size_t)c) % G)) continue;
        const bool is_float = col.is_float != 0;
        const bool masked = ANY_MASKED && col.words != nullptr;
        const uint64_t* __restrict__ data = (const uint64_t*)col.data;
        const size_t tail_start = col.head + col.n_tiles * TILE_ROWS;
        const size_t n_ragged = col.head + (col.n - tail_start);
        for (size_t i = tid; i < n_ragged; i += kBlock) {
            const size_t row = i < col.head ? i : tail_start + (i - col.head);
            uint64_t x = as_global(data)[row];
            unsigned valid = 1;
            if (ANY_MASKED && masked) {
                valid = row_bit(col.words, col.bit_off + row);
                cnt[c] += valid;
            }
            if (is_float) acc[c][0].add_f64(valid ? __longlong_as_double((long long)x) : 0.0);
            else acc[c][0].add_int(valid ? x : 0);
        }
    }

    // this workgroup's rows are scanned: the launch has begun to drain (see FusedArgs::early_word)
    if (a.early_word && a.early_mode == 0 && tid == 0)
        __hip_atomic_store(a.early_word, a.done_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);

    // ---- workgroup reduce, all columns ------------------------------------------------------------------------------
    __shared__ Partial lds[kFusedMax][kWaves];
    __shared__ int is_last;
#pragma unroll
    for (int c = 0; c < kFusedMax; ++c) {
        if (c >= a.n_cols) continue;
        const bool is_float = a.col[c].is_float != 0;
        acc[c][0].merge(acc[c][1], is_float);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            acc[c][0].merge(acc[c][0].shfl_down(off), is_float);
            cnt[c] += (uint64_t)__shfl_down((unsigned long long)cnt[c], off, 64);
        }
        if (lane == 0) {
            lds[c][wave].a = acc[c][0].a;
            lds[c][wave].b = acc[c][0].b;
            lds[c][wave].cnt = cnt[c];
        }
    }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int c = 0; c < kFusedMax; ++c) {
            if (c >= a.n_cols) continue;
            const bool is_float = a.col[c].is_float != 0;
            Acc2 s;
            s.a = lds[c][0].a;
            s.b = lds[c][0].b;
            uint64_t n = lds[c][0].cnt;
#pragma unroll
            for (int w = 1; w < kWaves; ++w) {
                Acc2 o;
                o.a = lds[c][w].a;
                o.b = lds[c][w].b;
                s.merge(o, is_float);
                n += lds[c][w].cnt;
            }
            uint64_t* q = (uint64_t*)&a.partials[(size_t)c * G + b];
            store_agent(q, s.a);
            store_agent(q + 1, s.b);
            store_agent(q + 2, n);
        }
        // publish (write-through stores, drained), then arrive: ma_reduce.hip's hand-off, once for all columns
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int last;
        if (G <= kFShardFrom) {
            last = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == G - 1;
        } else {
            const unsigned sh = b & (kFTicketShards - 1);
            const unsigned members = (G - sh + kFTicketShards - 1) / kFTicketShards;
            unsigned int* shard = a.ticket + kFTicketShardWord0 + sh * kFTicketShardStride;
            last = 0;
            const unsigned before = __hip_atomic_fetch_add(shard, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.early_word && a.early_mode >= 1 && a.early_mode <= 3 && before + 1 == (members * a.early_mode + 3) / 4)
                __hip_atomic_store(a.early_word, a.done_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (before == members - 1) {
                __hip_atomic_store(shard, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned shards_done = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
                last = shards_done == kFTicketShards;
                // early_mode 4..7: the early stamp when 1 / 2 / 4 / 6 whole shards (XCDs) have arrived
                if (a.early_word && a.early_mode >= 4 && shards_done == (a.early_mode == 4 ? 1u : a.early_mode == 5 ? 2u : a.early_mode == 6 ? 4u : 6u))
                    __hip_atomic_store(a.early_word, a.done_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        is_last = last;
    }
    __syncthreads();
    if (!is_last) return;
    // ---- the last workgroup folds the partials. Every thread loads its share of EVERY column's partials first (index order
    // per thread; the columns' loads are independent, so they are all in flight in one round trip — a wave per column walking
    // its list in a loop was tried and cost 2 us more), then the columns are merged across lanes, the four waves' results
    // meet in LDS once, and thread c finishes column c.
    Acc2 tot[kFusedMax];
    uint64_t tc[kFusedMax];
#pragma unroll
    for (int c = 0; c < kFusedMax; ++c) {
        tot[c].init();
        tc[c] = 0;
    }
    for (unsigned i = tid; i < G; i += kBlock) {
#pragma unroll
        for (int c = 0; c < kFusedMax; ++c) {
            if (c >= a.n_cols) continue;
            const uint64_t* q = (const uint64_t*)&a.partials[(size_t)c * G + i];
            Acc2 o;
            o.a = load_agent(q);
            o.b = load_agent(q + 1);
            tot[c].merge(o, a.col[c].is_float != 0);
            tc[c] += load_agent(q + 2);
        }
    }
    __syncthreads();  // lds[][] is reused
#pragma unroll
    for (int c = 0; c < kFusedMax; ++c) {
        if (c >= a.n_cols) continue;
        const bool is_float = a.col[c].is_float != 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            tot[c].merge(tot[c].shfl_down(off), is_float);
            tc[c] += (uint64_t)__shfl_down((unsigned long long)tc[c], off, 64);
        }
        if (lane == 0) {
            lds[c][wave].a = tot[c].a;
            lds[c][wave].b = tot[c].b;
            lds[c][wave].cnt = tc[c];
        }
    }
    __syncthreads();
    if ((int)tid < a.n_cols) {
        const int c = (int)tid;
        const bool is_float = a.col[c].is_float != 0;
        Acc2 s;
        s.a = lds[c][0].a;
        s.b = lds[c][0].b;
        uint64_t n = lds[c][0].cnt;
#pragma unroll
        for (int w = 1; w < kWaves; ++w) {
            Acc2 o;
            o.a = lds[c][w].a;
            o.b = lds[c][w].b;
            s.merge(o, is_float);
            n += lds[c][w].cnt;
        }
        uint64_t* out = a.col[c].out;
        if (is_float) {
            DDAcc d;
            d.from_words(s.a, s.b);
            d.normalise();
            out[0] = (uint64_t)__double_as_longlong(d.hi);
            out[1] = (uint64_t)__double_as_longlong(d.lo);
            out[2] = n;
        } else if (a.col[c].as_partial) {
            out[0] = s.a;
            out[1] = 0;
            out[2] = n;
        } else {
            out[0] = s.a;
            out[1] = n;
        }
        if (a.col[c].as_partial) out[3] = 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the results have left before the stamp below
        __builtin_amdgcn_wave_barrier();
        if (tid == 0) {
            __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch on this stream
            if (a.done_word) __hip_atomic_store(a.done_word, a.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

template <int UNROLL, bool ANY_MASKED>
static void launch_fused(ma_ctx* ctx, const FusedArgs& a, int grid, int pace) {
    if constexpr (ANY_MASKED) {
        hipLaunchKernelGGL((sum_fused_kernel<UNROLL, true, 0>), dim3(grid), dim3(kBlock), 0, ctx->stream, a);
    } else {
#if MA_TUNING
        switch (pace) {
            case 0: hipLaunchKernelGGL((sum_fused_kernel<UNROLL, false, 0>), dim3(grid), dim3(kBlock), 0, ctx->stream, a); return;
            case 24: hipLaunchKernelGGL((sum_fused_kernel<UNROLL, false, 24>), dim3(grid), dim3(kBlock), 0, ctx->stream, a); return;
            default: break;
        }
#endif
        (void)pace;  // the shipped build: the swept pacing (20 idle cycles between a wave's loads)
        hipLaunchKernelGGL((sum_fused_kernel<UNROLL, false, 20>), dim3(grid), dim3(kBlock), 0, ctx->stream, a);
    }
}

}  // namespace ma

using namespace ma;

namespace ma {
ma_status sum_fused_impl(ma_ctx* ctx, size_t n_cols, const ma_fused_column* cols, uint64_t* stamp, uint64_t stamp_value,
                         bool as_partials = false, uint64_t* early_stamp = nullptr);
}

extern "C" ma_status ma_sum_fused(ma_ctx* ctx, size_t n_cols, const ma_fused_column* cols) {
    return sum_fused_impl(ctx, n_cols, cols, nullptr, 0);
}

extern "C" ma_status ma_sum_fused_stamped(ma_ctx* ctx, size_t n_cols, const ma_fused_column* cols, uint64_t* stamp,
                                          uint64_t stamp_value) {
    MA_REQUIRE(stamp != nullptr && ((uintptr_t)stamp & 7) == 0, MA_ERR_INVALID_ARGUMENT, "stamp is NULL or misaligned");
    MA_REQUIRE(pointer_kind(stamp) != kPageable, MA_ERR_INVALID_ARGUMENT, "stamp must be device-reachable memory");
    return sum_fused_impl(ctx, n_cols, cols, stamp, stamp_value);
}

extern "C" ma_status ma_sum_fused_stamped_early(ma_ctx* ctx, size_t n_cols, const ma_fused_column* cols, uint64_t* stamp,
                                                uint64_t stamp_value, uint64_t* early_stamp) {
    MA_REQUIRE(stamp != nullptr && ((uintptr_t)stamp & 7) == 0, MA_ERR_INVALID_ARGUMENT, "stamp is NULL or misaligned");
    MA_REQUIRE(early_stamp != nullptr && ((uintptr_t)early_stamp & 7) == 0, MA_ERR_INVALID_ARGUMENT, "early_stamp is NULL or misaligned");
    MA_REQUIRE(pointer_kind(stamp) != kPageable && pointer_kind(early_stamp) != kPageable, MA_ERR_INVALID_ARGUMENT,
               "stamps must be device-reachable memory");
    return sum_fused_impl(ctx, n_cols, cols, stamp, stamp_value, false, early_stamp);
}

// as_partials: every column's `out` receives a 32-byte ma::Partial {sum or hi, 0 or lo, valid count, 0} — the input of
// ma_reduce_batch.hip's folds — instead of the record words of the public entry points.
ma_status ma::sum_fused_impl(ma_ctx* ctx, size_t n_cols, const ma_fused_column* cols, uint64_t* stamp, uint64_t stamp_value,
                             bool as_partials, uint64_t* early_stamp) {
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    MA_REQUIRE(n_cols >= 1 && n_cols <= (size_t)kFusedMax && cols != nullptr, MA_ERR_INVALID_ARGUMENT,
               "ma_sum_fused takes 1..%d columns", kFusedMax);
    MA_ENTER(ctx);
    MA_HIP(hipSetDevice(ctx->device));
    CallScope scope(ctx);
    FusedArgs a{};
    a.n_cols = (int)n_cols;
    bool any_masked = false;
    for (size_t c = 0; c < n_cols; ++c) {
        const ma_fused_column& in = cols[c];
        MA_REQUIRE(in.format_code == 'l' || in.format_code == 'L' || in.format_code == 'g', MA_ERR_UNSUPPORTED,
                   "column %zu: format '%c' (ma_sum_fused takes the 8-byte formats l, L and g)", c, (char)in.format_code);
        MA_REQUIRE(in.n == 0 || in.data != nullptr, MA_ERR_INVALID_ARGUMENT, "column %zu: data is NULL", c);
        MA_REQUIRE(((uintptr_t)in.data & 7) == 0, MA_ERR_INVALID_ARGUMENT, "column %zu: data pointer %p is not 8-byte aligned", c, in.data);
        MA_REQUIRE(in.out != nullptr && ((uintptr_t)in.out & 7) == 0, MA_ERR_INVALID_ARGUMENT, "column %zu: out is NULL or misaligned", c);
        MA_REQUIRE(in.n == 0 || pointer_kind(in.data) != kPageable, MA_ERR_INVALID_ARGUMENT,
                   "column %zu: ma_sum_fused scans device-resident (or pinned) columns in place; sum a pageable host column with "
                   "ma_i64_sum / ma_f64_sum_dd", c);
        MA_REQUIRE(pointer_kind(in.out) != kPageable, MA_ERR_INVALID_ARGUMENT,
                   "column %zu: out must be device-reachable (device or ma_alloc64_pinned memory)", c);
        FusedCol& col = a.col[c];
        col.data = in.data;
        col.n = in.n;
        col.is_float = in.format_code == 'g';
        col.as_partial = as_partials ? 1 : 0;
        col.out = in.out;
        const bool masked = in.mask_bits != nullptr && in.null_count != 0 && in.n != 0;  // the all_true / null_count gate
        if (masked) {
            MA_REQUIRE(pointer_kind(in.mask_bits) != kPageable, MA_ERR_INVALID_ARGUMENT,
                       "column %zu: the validity bitmap must be device-reachable", c);
            MA_TRY(scope.in_mask(in.mask_bits, in.mask_bit_offset, in.n, &col.words, &col.bit_off));
            col.last_word = (col.bit_off + in.n - 1) >> 6;
            any_masked = true;
        }
    }
    // as_partials = the nested use by ma_sum_columns (its few-long-columns route): this call's scope ends before the outer
    // call's launches have run, so nothing may have been staged into it — the outer call only hands over device operands.
    MA_REQUIRE(!as_partials || !scope.staged(), MA_ERR_INVALID_ARGUMENT,
               "internal: the nested fused scan was handed an operand that had to be staged");
    // Launch shape: ma_reduce.hip's — dense 8-byte scans one workgroup per CU with 8 paced loads per lane, anything with
    // validity work between the loads two per CU with 4; mid-size jobs (<= 24 tiles per CU) three per CU.
    const int unroll = any_masked ? 4 : 8;
    const size_t tile_rows = (size_t)64 * 2 * (size_t)unroll * kWaves;
    size_t total_tiles = 0;
    for (size_t c = 0; c < n_cols; ++c) {
        FusedCol& col = a.col[c];
        col.head = (col.n && ((uintptr_t)col.data & 15)) ? 1 : 0;
        col.n_tiles = (col.n - col.head) / tile_rows;
        col.tile0 = total_tiles;
        total_tiles += col.n_tiles;
    }
    a.total_tiles = total_tiles;
    // (with validity: the kernel keeps a tile requested ahead, so one workgroup per CU is enough for long columns — i64 + f64
    // sharing a bitmap, 125 M rows each: 302.5 us at one, 306 at two, 314 at three; short jobs keep two)
    int bpc = ctx->blocks_per_cu > 0 ? ctx->blocks_per_cu : (any_masked ? (total_tiles > (size_t)24 * (size_t)ctx->num_cus ? 1 : 2) : 1);
    if (!any_masked && ctx->blocks_per_cu <= 0 && total_tiles <= (size_t)24 * (size_t)ctx->num_cus) bpc = 3;
    int grid = grid_for(ctx, total_tiles, bpc);
    if ((size_t)grid * n_cols > (size_t)kMaxGrid) grid = (int)((size_t)kMaxGrid / n_cols);
    a.partials = ctx->partials;
    a.ticket = ctx->ticket;
    static const int kPace[8] = {-1, 0, 16, 20, 24, 32, 0, 0};
    const int sel = kPace[(tuning_variant(ctx) >> 5) & 7];
    const int pace = sel >= 0 ? sel : 20;
    a.done_word = stamp;  // stored (system-scope release) by the launch's final thread behind its results
    a.done_seq = stamp_value;
    a.early_word = early_stamp;
    {   // FusedArgs::early_mode; ctx variant bits 19-21 override (tuning): 0 = the default, v = mode v - 1
        const unsigned sel = ((unsigned)tuning_variant(ctx) >> 19) & 7u;
        a.early_mode = grid > (int)kFShardFrom ? (sel ? sel - 1 : 5u) : 0u;
    }
    if (any_masked) launch_fused<4, true>(ctx, a, grid, 0);
    else launch_fused<8, false>(ctx, a, grid, pace);
    MA_HIP(hipGetLastError());
    return end_call(ctx, scope);
}
