// Sum / valid-count / mean reductions for gfx950 (MI355X).
//
// Replaces the reference's only sum implementations, which live in its bench binaries:
//   simd_sum_i64 / simd_sum_f64      benches/benchmark_parallel_simd.rs:44-59, 63-78
//   rayon_simd_sum_{i64,f64}         benches/benchmark_parallel_simd.rs:81-98   (par_chunks(1<<20) + tree sum)
//   4x-unrolled accumulators         benches/hotloop_benchmark_simd.rs:56-174
//   scalar `for &v in slice`         benches/hotloop_benchmark_std.rs:49-57
//
// Design:
//   * HBM-bound scan, 8 B/row (4 B/row for 32-bit types) + 1/8 B/row of validity. No MFMA.
//   * A workgroup = 4 wave64s. A wave owns a contiguous run of UNROLL KiB: lane k issues UNROLL
//     16-byte loads (global_load_dwordx4), 1 KiB per wave-instruction, before it consumes any; the dense scans leave
//     16-24 idle cycles between consecutive loads (pace_loads: +1-2 % read rate, tools/ubench_pace.hip).
//   * Tiles are dealt round-robin to workgroups (tile t -> workgroup t mod grid), grid = CUs x blocks/CU,
//     so at any instant all CUs stream neighbouring DRAM pages.
//   * Validity: one u64 word covers 64 rows = one wave64. Lanes 0..W load the W+1 words that cover the
//     wave's run, funnel-shift them onto the run's first bit (any bit offset works), popcount them for the
//     valid-count, and hand each lane its R bits per load through a cross-lane read.
//   * Integers accumulate in wrapping u64. Floats accumulate in double-double (Knuth two-sum): the
//     kernel is bandwidth-bound with >6x VALU headroom, so the result is within 1 ULP of the exactly
//     rounded sum regardless of order, and bit-reproducible for a fixed grid.
//   * Wave reduce by shuffles -> 4 LDS slots -> one 32-byte partial per workgroup -> the workgroup that
//     draws the last ticket folds all partials in index order (agent-scope release/acquire).
#include <atomic>
#include <chrono>
#include <cmath>
#include <limits>

#include "ma_acc.hpp"
#include "ma_device.hpp"

namespace ma {

struct SumArgs {
    const void* data;       // element pointer of the window
    size_t n;               // rows in the window
    size_t head;            // leading rows handled by the ragged path so that data+head is 16-byte aligned
    size_t n_tiles;         // full workgroup tiles after `head`
    const uint64_t* words;  // validity words (8-byte aligned base) or nullptr
    size_t bit_off;         // bit index of row 0 relative to `words`
    size_t last_word;       // index of the last word that holds a window bit
    Partial* partials;
    unsigned int* ticket;
    uint64_t* out_a;        // integer sum / double bits (f64 result) — may be nullptr
    uint64_t* out_b;        // double-double low part — may be nullptr
    uint64_t* out_cnt;      // valid count — may be nullptr
    double* out_mean;       // sum / count — may be nullptr
    int pace;               // host side only: idle cycles between a wave's consecutive loads (selects the instantiation)
    int interleave;         // dense only: 1 = n_tiles counts 1-KiB pieces dealt to ALL waves of the grid in turn
    int mode;               // 0: out_a = final value (int, or rounded double); 1: out_a/out_b = double-double
    int is_signed;          // integer mean: interpret the 64-bit sum as signed
    int fenced;             // 1: round-1 publish (release / acquire fences) instead of sc1 stores (A/B only)
    uint64_t* done_word;    // pinned host word the final thread stamps with done_seq after the results (or nullptr):
    uint64_t done_seq;      //   a synchronous call polls it instead of paying hipStreamSynchronize's wake-up
                            //   (ma_scan_lanes_sum: the lane's stamp line instead — the hand-off to the scan after the next)
    uint64_t* early_word;   // ma_scan_lanes_sum: stamped with done_seq while the launch DRAINS — two of the eight ticket shards
                            //   have arrived (grids of up to 96 workgroups: the first workgroup that has scanned its rows) —, what
                            //   the next scan, on the other stream, waits for (FusedArgs::early_word, ma_reduce_fused.hip)
};

// Arrival counters inside the context's zeroed scratch block (ma_ctx.hip): word 0 is the top ticket; the shards sit on
// their own 64-byte lines further on.
constexpr unsigned kTicketShards = 8;
constexpr unsigned kTicketShardWord0 = 64;   // 256 bytes in
constexpr unsigned kTicketShardStride = 16;  // 64 bytes
constexpr unsigned kShardFrom = 96;          // grids up to this size arrive on the one ticket

template <typename T, int UNROLL, bool MASKED, bool NT, bool IL = false, int PACE = 0>
__global__ __launch_bounds__(kBlock) void sum_kernel(SumArgs a) {
    // 1- and 2-byte columns keep their 16 bytes as four dwords: narrow_vec_sum works on dwords anyway, and a vector of 1-byte
    // elements lost the loads' non-temporal hint on the way through the optimiser (u8 / i8 read at 6.2 TB/s, u16 at 6.9)
    typedef typename std::conditional<(sizeof(T) <= 2), MaU4, typename Vec16<T>::type>::type V;
    typedef typename AccOf<T>::type Acc;
    constexpr int R = 16 / (int)sizeof(T);           // rows per lane per load
    constexpr int WPT = R * UNROLL;                  // validity words per wave run
    constexpr size_t WAVE_ROWS = (size_t)64 * R * UNROLL;
    constexpr size_t TILE_ROWS = WAVE_ROWS * kWaves;
    // (1-byte rows at 8 loads per lane: 128 run words, two per lane — the kDeepBytes branch below)
    constexpr bool kDeepBytes = MASKED && sizeof(T) == 1 && UNROLL == 8;
    static_assert(!MASKED || WPT < 64 || kDeepBytes, "a wave must be able to load its validity words in one instruction");
    constexpr bool kNarrow = sizeof(T) <= 2;  // 8 / 16 rows per load: summed inside 32-bit registers (narrow_vec_sum)

    const unsigned tid = threadIdx.x;
    const unsigned lane = tid & 63;
    const unsigned wave = tid >> 6;
    const T* __restrict__ data = (const T*)a.data;

    Acc acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r].init();
    uint64_t cnt = 0;

    // ---- dense, piece-interleaved: wave instruction u of round k reads piece (k*U + u) * n_waves + wave_id --------
    // All waves of the grid sweep memory as one front, 1 KiB per wave instruction: at any moment the chip reads
    // U contiguous spans of n_waves KiB instead of n_waves scattered runs. Optional (ctx variant bit 4).
    if constexpr (!MASKED && IL) {
        const size_t n_waves = (size_t)gridDim.x * kWaves, wave_id = (size_t)blockIdx.x * kWaves + wave;
        const size_t n_pieces = a.n_tiles, round = n_waves * UNROLL;
        const V* __restrict__ base = (const V*)(data + a.head) + lane;
        size_t k = 0;
        for (; k + round <= n_pieces; k += round) {  // full rounds: no bounds checks
            V v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = load16<V, NT>(base + (k + (size_t)u * n_waves + wave_id) * 64);
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
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {  // last, partial round
            const size_t piece = k + (size_t)u * n_waves + wave_id;
            if (piece < n_pieces) {
                V v = load16<V, NT>(base + piece * 64);
                if constexpr (kNarrow) {
                    acc[0].add(narrow_vec_sum<T>(v, ~0u));
                } else {
#pragma unroll
                    for (int r = 0; r < R; ++r) acc[r].add((T)v[r]);
                }
            }
        }
    }
    // ---- full tiles: 16-byte loads, no bounds checks --------------------------------------------
    if constexpr (kDeepBytes) {
        // 1-byte rows with validity in the shape of the dense scans (round 5): EIGHT loads per lane, one workgroup per CU, the
        // next tile requested before this one is consumed. A wave's run is 8 KiB of rows = 128 validity words: lane l holds run
        // words l and 64 + l (two wave loads; a third fetches word 128, the funnel partner of the last one, on every lane).
        // With 2 loads per tile — all that fits a one-word-per-lane run — the per-tile work (funnel shift, count, addresses,
        // loop) was spread over 2 loads and three workgroups per CU had to cover it: 35-43 vector instructions per load and a
        // quarter of the wave cycles in issue stalls (profiles/r05_subfamily_counters.md), 6.88 TB/s against 7.25 dense.
        const size_t G = gridDim.x, first = blockIdx.x;
        const size_t n_mine = first < a.n_tiles ? (a.n_tiles - first + G - 1) / G : 0;
        auto issue = [&](size_t k, V (&v)[UNROLL], uint64_t (&raw)[3], size_t& row0) {
            const bool real = k < n_mine;
            row0 = a.head + (real ? first + k * G : 0) * TILE_ROWS + (size_t)wave * WAVE_ROWS;
            const V* __restrict__ p = real ? (const V*)(data + row0) + lane : (const V*)a.partials;
            const size_t stride = real ? 64 : 0;
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = load16<V, NT>(p + (size_t)u * stride);
            const size_t w0 = (a.bit_off + row0) >> 6;
            const auto gw = as_global(real ? a.words : (const uint64_t*)a.partials);
#pragma unroll
            for (int j = 0; j < 3; ++j) {  // clamped, not skipped, past the window's last word: a full tile never needs a word behind it
                size_t idx = w0 + (size_t)(64 * j) + (j < 2 ? lane : 0u);
                idx = idx < a.last_word ? idx : a.last_word;
                raw[j] = gw[real ? idx : 0];  // (streamed or plain: the same rate, profiles/r05_sweep_gated_bytes.jsonl)
            }
        };
        auto use = [&](const V (&v)[UNROLL], const uint64_t (&raw)[3], size_t row0) {
            const unsigned sh = (unsigned)((a.bit_off + row0) & 63);
            uint64_t w[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                uint64_t nx = (uint64_t)__shfl_down((unsigned long long)raw[j], 1, 64);
                const uint64_t first_of_next = (uint64_t)__shfl((unsigned long long)raw[j + 1], 0, 64);
                if (lane == 63) nx = first_of_next;
                w[j] = sh ? ((raw[j] >> sh) | (nx << (64 - sh))) : raw[j];
                cnt += (uint64_t)__popcll(w[j]);  // every lane holds two distinct run words
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                // rows of load u on lane l: bits [(64 u + l) 16, +16) of the run = word 16 u + l / 4, quarter l % 4
                const uint64_t word = (uint64_t)__shfl((unsigned long long)w[u / 4], (u * 16 + (int)(lane >> 2)) & 63, 64);
                const unsigned bits = (unsigned)(word >> ((lane & 3u) * 16u)) & 0xFFFFu;
                acc[0].add(narrow_vec_sum<T>(v[u], bits));
            }
        };
        V va[UNROLL], vb[UNROLL];
        uint64_t ra[3], rb[3];
        size_t row_a, row_b;
        issue(0, va, ra, row_a);
        for (size_t k = 0; k < n_mine; k += 2) {
            issue(k + 1, vb, rb, row_b);
            use(va, ra, row_a);
            issue(k + 2, va, ra, row_a);
            if (k + 1 < n_mine) use(vb, rb, row_b);
        }
    } else if constexpr (MASKED) {
        // With validity a tile is few loads per lane and the validity work sits between one tile's loads and the next's: the
        // wave's bytes in flight drop to zero once per tile, which the second wave per SIMD only partly covers (i8 / u8 with
        // nulls 6.7 TB/s, f32 and i16 7.0, against 7.2-7.3 dense). So the NEXT tile's rows and raw validity words are
        // requested before this tile is consumed — the scheme of ma_reduce_batch.hip's wave kernel and of the fused scan, and
        // their rules: a request is ALWAYS UNROLL + 1 loads (with nothing left to request every lane reads the first bytes of
        // `partials` instead), because loads issued on some paths only make the compiler's in-order wait counts assume the
        // shortest queue; two register sets swap roles (a copy would wait for what it copies); the funnel shift of the words
        // waits until they are wanted (finish_run_words).
        const size_t G = gridDim.x, first = blockIdx.x;
        const size_t n_mine = first < a.n_tiles ? (a.n_tiles - first + G - 1) / G : 0;  // this workgroup's tiles
        auto issue = [&](size_t k, V (&v)[UNROLL], uint64_t& raw, size_t& row0) {
            const bool real = k < n_mine;
            row0 = a.head + (real ? first + k * G : 0) * TILE_ROWS + (size_t)wave * WAVE_ROWS;
            const V* __restrict__ p = real ? (const V*)(data + row0) + lane : (const V*)a.partials;
            const size_t stride = real ? 64 : 0;
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = load16<V, NT>(p + (size_t)u * stride);
            // lane l <= WPT holds run word l (the last one only feeds the funnel shift); clamped, not skipped, past the
            // window's last word — a full tile never needs a word behind it
            size_t idx = ((a.bit_off + row0) >> 6) + (lane < (unsigned)WPT ? lane : (unsigned)WPT);
            idx = idx < a.last_word ? idx : a.last_word;
            raw = as_global(real ? a.words : (const uint64_t*)a.partials)[real ? idx : 0];
        };
        auto use = [&](const V (&v)[UNROLL], uint64_t raw, size_t row0) {
            const uint64_t aw = finish_run_words(raw, a.bit_off + row0);
            if (lane < (unsigned)WPT) cnt += (uint64_t)__popcll(aw);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const unsigned bits = lane_bits<R>(aw, u, lane);
                if constexpr (kNarrow) {
                    acc[0].add(narrow_vec_sum<T>(v[u], bits));
                } else {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        T x = ((bits >> r) & 1u) ? v[u][r] : (T)0;
                        acc[r].add(x);
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
    }
    for (size_t t = blockIdx.x; t < ((MASKED || IL) ? 0 : a.n_tiles); t += gridDim.x) {
        const size_t row0 = a.head + t * TILE_ROWS + (size_t)wave * WAVE_ROWS;
        const V* __restrict__ p = (const V*)(data + row0) + lane;
        V v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            v[u] = load16<V, NT>(p + (size_t)u * 64);
            if (u + 1 < UNROLL) pace_loads<PACE>();
        }
        // the paced loads are all issued before anything is consumed (the asm statements split the scheduling region,
        // and the scheduler would otherwise start on v[0] after three loads)
        if constexpr (PACE > 0) __builtin_amdgcn_sched_barrier(0);
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

    // ---- ragged rows: the unaligned head and whatever follows the last full tile — on the first workgroup that has a
    // tile less than the others (tiles are dealt round-robin, so the last full round ends at workgroup n_tiles mod grid - 1):
    // its extra microsecond of row-at-a-time loads then overlaps the others' last tile instead of following it
    if (blockIdx.x == (unsigned)(a.n_tiles % gridDim.x)) {
        const size_t tail_start = a.head + a.n_tiles * ((!MASKED && IL) ? (size_t)64 * R : TILE_ROWS);
        const size_t n_ragged = a.head + (a.n - tail_start);
        for (size_t i = tid; i < n_ragged; i += kBlock) {
            size_t row = i < a.head ? i : tail_start + (i - a.head);
            T x = data[row];
            if constexpr (MASKED) {
                unsigned valid = row_bit(a.words, a.bit_off + row);
                cnt += valid;
                x = valid ? x : (T)0;
            }
            acc[0].add(x);
        }
    }
    if constexpr (!MASKED) {
        // Dense: every row is valid; credit the count once.
        if (blockIdx.x == 0 && tid == 0) cnt = a.n;
    }

    // ---- workgroup reduce -----------------------------------------------------------------------
#pragma unroll
    for (int r = 1; r < R; ++r) acc[0].merge(acc[r]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        acc[0].shfl_down_merge(off);
        cnt += (uint64_t)__shfl_down((unsigned long long)cnt, off, 64);
    }
    __shared__ Partial lds[kWaves];
    __shared__ int is_last;
    if (lane == 0) {
        acc[0].to_partial(lds[wave]);
        lds[wave].cnt = cnt;
    }
    __syncthreads();
    if (tid == 0) {
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
        Partial p;
        s.to_partial(p);
        p.cnt = c;
        int last;
        if (a.fenced) {
            // Round-1 form, kept for A/B (ctx variant bit 8): plain partial -> agent-scope release -> ticket -> acquire.
            p.pad = 0;
            a.partials[blockIdx.x] = p;
            if (a.early_word) __hip_atomic_store(a.early_word, a.done_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            unsigned int ticket = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = ticket == gridDim.x - 1;
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        } else {
            // Publish with write-through (sc1) stores, drain them, THEN arrive: one lane of the workgroup signals for all
            // of its stores, the consumer is the workgroup whose add came last and reads every partial with sc1 loads
            // (MI355X guide, workgroup hand-off table, first row). No release / acquire fence: each costs ~1.7 us, and
            // both sat on the last workgroup's critical path — most of a mid-size column's fixed overhead.
            uint64_t* q = (uint64_t*)&a.partials[blockIdx.x];
            store_agent(q, p.a);
            store_agent(q + 1, p.b);
            store_agent(q + 2, p.cnt);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (gridDim.x <= kShardFrom) {
                if (a.early_word) __hip_atomic_store(a.early_word, a.done_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                last = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
            } else {
                // Hundreds of arrivals on one word serialise at ~12 ns each: eight counters (workgroups b and b + 8 share
                // an XCD, so a shard's arrivals stay on one L2), the last arrival of each shard arrives at the top.
                const unsigned sh = blockIdx.x & (kTicketShards - 1);
                const unsigned members = (gridDim.x - sh + kTicketShards - 1) / kTicketShards;
                unsigned int* shard = a.ticket + kTicketShardWord0 + sh * kTicketShardStride;
                last = 0;
                if (__hip_atomic_fetch_add(shard, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1) {
                    __hip_atomic_store(shard, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch
                    const unsigned top = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    last = top == kTicketShards - 1;
                    if (a.early_word && top == 1)  // the second whole shard (XCD) has arrived
                        __hip_atomic_store(a.early_word, a.done_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
        is_last = last;
    }
    __syncthreads();
    if (!is_last) return;

    // ---- the last workgroup folds every partial, in index order per thread, then across threads. All 256 threads load (one
    // to three partials each for the shapes the host picks: independent loads, one round trip) — a single wave walking the
    // list in a loop was tried in round 4 and cost 2 us at 768 workgroups (twelve dependent-looking trips per lane).
    Acc tot;
    tot.init();
    uint64_t tc = 0;
    for (unsigned i = tid; i < gridDim.x; i += kBlock) {
        const uint64_t* q = (const uint64_t*)&a.partials[i];
        Acc o;
        o.from_words(load_agent(q), load_agent(q + 1));
        tot.merge(o);
        tc += load_agent(q + 2);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        tot.shfl_down_merge(off);
        tc += (uint64_t)__shfl_down((unsigned long long)tc, off, 64);
    }
    __syncthreads();  // lds[] is reused
    if (lane == 0) {
        tot.to_partial(lds[wave]);
        lds[wave].cnt = tc;
    }
    __syncthreads();
    if (tid == 0) {
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
        double as_double;
        if constexpr (std::is_same<Acc, DDAcc>::value) {
            DDAcc& d = reinterpret_cast<DDAcc&>(s);
            d.normalise();
            as_double = d.hi;
            if (a.mode == 1) {
                if (a.out_a) *a.out_a = (uint64_t)__double_as_longlong(d.hi);
                if (a.out_b) *a.out_b = (uint64_t)__double_as_longlong(d.lo);
            } else {
                if (a.out_a) *a.out_a = (uint64_t)__double_as_longlong(d.hi);
            }
        } else {
            IntAcc& q = reinterpret_cast<IntAcc&>(s);
            as_double = a.is_signed ? (double)(int64_t)q.s : (double)q.s;
            if (a.out_a) *a.out_a = q.s;
        }
        if (a.out_cnt) *a.out_cnt = c;
        if (a.out_mean) *a.out_mean = c ? as_double / (double)c : __longlong_as_double(0x7ff8000000000000ll);
        __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch on this stream
        // Last of all: tell a polling host that every result above has landed (system-scope release: the results may sit
        // in host memory or in device memory the host will read next).
        if (a.done_word) __hip_atomic_store(a.done_word, a.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ------------------------------------------------------------------------------------------------
// Host side
// ------------------------------------------------------------------------------------------------

// Idle cycles between a wave's consecutive loads in the dense scans, swept per type at 10^9 rows (enqueue_sum).
template <typename T>
constexpr int dense_pace() {
    return std::is_same<T, float>::value ? 24 : std::is_same<T, double>::value ? 20 : (sizeof(T) == 8 ? 24 : 16);
}

template <typename T, int UNROLL, bool MASKED, bool NT>
static void launch_sum(ma_ctx* ctx, const SumArgs& a, int grid) {
#if MA_TUNING
    if constexpr (!MASKED && NT && UNROLL == 8) {
        if (a.interleave) {  // tuning variant only (ctx variant bit 4)
            hipLaunchKernelGGL((sum_kernel<T, UNROLL, MASKED, NT, true>), dim3(grid), dim3(kBlock), 0, ctx->stream, a);
            return;
        }
    }
    // Load pacing (pace_loads) is instantiated for the non-temporal scans of 4- and 8-byte types: a.pace picks the cycle count.
    if constexpr (NT && sizeof(T) > 2) {
        switch (a.pace) {
            case 16: hipLaunchKernelGGL((sum_kernel<T, UNROLL, MASKED, NT, false, 16>), dim3(grid), dim3(kBlock), 0, ctx->stream, a); return;
            case 20: hipLaunchKernelGGL((sum_kernel<T, UNROLL, MASKED, NT, false, 20>), dim3(grid), dim3(kBlock), 0, ctx->stream, a); return;
            case 24: hipLaunchKernelGGL((sum_kernel<T, UNROLL, MASKED, NT, false, 24>), dim3(grid), dim3(kBlock), 0, ctx->stream, a); return;
            case 32: hipLaunchKernelGGL((sum_kernel<T, UNROLL, MASKED, NT, false, 32>), dim3(grid), dim3(kBlock), 0, ctx->stream, a); return;
            default: break;
        }
    }
#else
    // the shipped build: ONE pacing per type (the swept default of enqueue_sum), dense scans of 4- and 8-byte types only
    if constexpr (NT && !MASKED && sizeof(T) > 2) {
        constexpr int kPace = dense_pace<T>();
        if (a.pace == kPace) {
            hipLaunchKernelGGL((sum_kernel<T, UNROLL, MASKED, NT, false, kPace>), dim3(grid), dim3(kBlock), 0, ctx->stream, a);
            return;
        }
    }
#endif
    hipLaunchKernelGGL((sum_kernel<T, UNROLL, MASKED, NT, false>), dim3(grid), dim3(kBlock), 0, ctx->stream, a);
}

// Enqueues one sum launch on ctx->stream. `a` carries device-reachable data / validity (data, n, words, bit_off,
// last_word) and output addresses; the launch shape is chosen here.
template <typename T>
static ma_status enqueue_sum(ma_ctx* ctx, SumArgs a, bool masked) {
    const size_t n = a.n;
    a.partials = ctx->partials;
    a.ticket = ctx->ticket;
    constexpr int R = 16 / (int)sizeof(T);
    // Launch shape. Measured on MI355X at 10^9 rows (profiles/r01_sweep_sum_v2.txt): what matters is ~8 KiB of
    // loads in flight per SIMD (unroll x workgroups/CU = 8); more only queues, less starves. Dense 8-byte
    // scans are fastest with ONE wave per SIMD and 8 loads each; anything with per-row work between the
    // loads (validity bits, 4 rows per lane) wants two waves per SIMD to overlap it.
    // ctx->variant: bit 0 = plain (temporal) loads instead of non-temporal; bits 1-3 = unroll
    // {0: auto, 1: 2, 2: 4, 3: 8, 4: 16}. ctx->blocks_per_cu: 0 = auto.
    const int variant = tuning_variant(ctx);  // every bit this launch reads is a tuning bit: 0 in the shipped build
    const bool nt = (variant & 1) == 0;
    // 32-bit columns (profiles/r01_sweep_sum_32bit.txt, 2 x 10^9 rows): dense integers behave like the 8-byte scans
    // (7.15 vs 6.98 TB/s); f32 — widened to f64 and accumulated in double-double, 4 rows per load — wants two waves per
    // SIMD when dense (7.06 vs 6.62 TB/s) and the deeper unroll when masked (6.93 vs 6.42 TB/s).
    const bool lean = !masked && (R == 2 || std::is_integral<T>::value);
    int unroll = lean ? 8 : 4;
    if (masked && std::is_same<T, float>::value) unroll = 8;
    switch ((variant >> 1) & 7) {
        case 1: unroll = 2; break;
        case 2: unroll = 4; break;
        case 3: unroll = 8; break;
        case 4: unroll = (R * 16 < 64 && !masked) ? 16 : 8; break;
        default: break;
    }
    // (with validity the kernel keeps a tile requested ahead — ONE wave per SIMD then carries two tiles in flight, and a
    // second workgroup per CU only queues: i64 / i32 / i16 with 10 % nulls 7.29 / 7.25 / 7.15 TB/s at one per CU, 6.7 / 6.7 /
    // 6.5 at two, tools/sweep_masked_sum.py; short columns keep two for their ramp; dense f32 keeps its two)
    int bpc = ctx->blocks_per_cu > 0 ? ctx->blocks_per_cu : (lean || masked ? 1 : 2);
    if constexpr (R >= 8) {
        // narrow types, masked: a wave's validity run must fit one word per lane (R x unroll < 64): 4 loads of 8 rows, 2
        // loads of 16 rows; the bytes in flight per SIMD then come from more waves (2 / 3 workgroups per CU). Dense scans
        // have no such limit and take the 8-deep shape of the wider integers.
        if (masked) {
            // 2-byte rows: 4 loads of 8 rows (a 32-word run). 1-byte rows: 8 loads of 16 rows, two run words per lane (round 5);
            // ctx variant unroll = 2 keeps round 4's shape — 2 loads, three workgroups per CU — for A/B.
            const bool old_bytes = R == 16 && ((variant >> 1) & 7) == 1;
            unroll = R == 8 ? 4 : (old_bytes ? 2 : 8);
            if (ctx->blocks_per_cu <= 0) bpc = (R == 8 || !old_bytes) ? 1 : 3;  // swept: tools/sweep_masked_sum.py
        } else if (unroll != 2 && unroll != 4) {
            unroll = 8;
        }
    }
    const size_t tile_rows = (size_t)64 * R * unroll * kWaves;
    // Rows in front of the first 16-byte boundary.
    size_t head = 0;
    if (n) {
        uintptr_t mis = (uintptr_t)a.data & 15;
        head = mis ? (16 - mis) / sizeof(T) : 0;
        if (head > n) head = n;
    }
    a.head = head;
    a.n_tiles = (n - head) / tile_rows;
    // variant bit 4: piece-interleaved mapping for dense scans. A/B in one process: +2 % on one MI355X, -1 % on
    // another (profiles/r01_ubench_sum_v2.txt vs r01_sweep_sum_v3.txt) — within the device-to-device spread, so the
    // tiled mapping stays the default.
    a.interleave = (!masked && nt && unroll == 8 && R < 8 && (variant & 16) != 0) ? 1 : 0;
    a.fenced = ((variant & 256) || ctx->fenced_reduce) ? 1 : 0;
    // Load pacing (pace_loads): idle cycles between a wave's consecutive loads. Swept per type at 10^9 rows
    // (profiles/r01_sweep_sum_pace.txt): dense i64 1.117 -> 1.099 ms at 24 cycles, f64 1.106 -> 1.095 at 20, i32 0.586 ->
    // 0.583 at 16, f32 0.591 -> 0.566 at 16-32; the masked kernels, which already spend cycles on validity words between
    // their loads, only lose by it (i64 1.124 -> 1.135 at 16). ctx->variant bits 5-7 override for tuning: 0 = these
    // defaults, 1 = none, 2..5 = 16 / 20 / 24 / 32 cycles.
    {
        static const int kPace[8] = {-1, 0, 16, 20, 24, 32, 0, 0};
        const int sel = kPace[(variant >> 5) & 7];
        const int dense_default = dense_pace<T>();
        a.pace = sel >= 0 ? sel : ((masked || R >= 8) ? 0 : dense_default);
    }
    // Mid-size columns (up to ~24 tiles per CU: 2^24 8-byte rows — the chunk sizes the reference actually runs at,
    // src/structs/chunked/super_array.rs:51-59): the lean shape's one workgroup per CU leaves the memory pipeline
    // half empty during the ramp and the tail of so short a scan; three per CU are 3-6 % faster there, and lose 3 %
    // from 2^26 rows on (profiles/r02_sweep_mid.jsonl).
    if (lean && R < 8 && ctx->blocks_per_cu <= 0 && a.n_tiles <= (size_t)24 * (size_t)ctx->num_cus) bpc = 3;
    if (masked && R < 16 && ctx->blocks_per_cu <= 0 && a.n_tiles <= (size_t)24 * (size_t)ctx->num_cus) bpc = 2;
    size_t work = a.n_tiles;
    if (a.interleave) {
        a.n_tiles = (n - head) / ((size_t)64 * R);  // 1-KiB pieces
        work = (a.n_tiles + (size_t)unroll * kWaves - 1) / ((size_t)unroll * kWaves);
    }
    int grid = grid_for(ctx, work, bpc);

#if MA_TUNING
#define MA_LAUNCH_U(U, M)                                     \
    do {                                                      \
        if (nt) launch_sum<T, U, M, true>(ctx, a, grid);      \
        else launch_sum<T, U, M, false>(ctx, a, grid);        \
    } while (0)
#else  // non-temporal loads only (plain loads are a tuning form)
#define MA_LAUNCH_U(U, M) launch_sum<T, U, M, true>(ctx, a, grid)
#endif
    if constexpr (R >= 8) {  // narrow types: masked = the one shape per width chosen above (deeper overflows the validity run)
        constexpr int U = R == 8 ? 4 : 2;
        (void)U;
        bool deep = false;
        if constexpr (R == 16) {
            if (masked && unroll == 8) {
                MA_LAUNCH_U(8, true);
                deep = true;
            }
        }
        if (deep) {
        }
#if MA_TUNING
        else if (masked) MA_LAUNCH_U(U, true);
        else if (unroll == 2) MA_LAUNCH_U(2, false);
        else if (unroll == 4) MA_LAUNCH_U(4, false);
#else  // the shapes the shipped build chooses: masked 2-byte rows 4 loads (1-byte rows took `deep`), dense 8
        else if (masked) {
            if constexpr (R == 8) MA_LAUNCH_U(4, true);
        }
#endif
        else MA_LAUNCH_U(8, false);
    } else if (masked) {
        switch (unroll) {
#if MA_TUNING
            case 2: MA_LAUNCH_U(2, true); break;
#endif
            case 8: MA_LAUNCH_U(8, true); break;
            default: MA_LAUNCH_U(4, true); break;
        }
    } else {
        switch (unroll) {
#if MA_TUNING
            case 2: MA_LAUNCH_U(2, false); break;
            case 16:
                if constexpr (R * 16 < 64) { MA_LAUNCH_U(16, false); } else { MA_LAUNCH_U(8, false); }
                break;
#endif
            case 8: MA_LAUNCH_U(8, false); break;
            default: MA_LAUNCH_U(4, false); break;
        }
    }
#undef MA_LAUNCH_U
    MA_HIP(hipGetLastError());
    return MA_OK;
}

// Double-double / integer fold of per-tile records on the host (same operations as DDAcc::merge / normalise; the
// translation unit is built with -ffp-contract=off, so host and device round identically).
struct HostFold {
    uint64_t isum = 0, cnt = 0;
    double hi = 0.0, lo = 0.0;
    void add(uint64_t a, uint64_t b, uint64_t c, bool is_float) {
        cnt += c;
        if (!is_float) {
            isum += a;
            return;
        }
        double oh, ol;
        memcpy(&oh, &a, 8);
        memcpy(&ol, &b, 8);
        const double t = hi + oh;
        const double bp = t - hi;
        const double e = (hi - (t - bp)) + (oh - bp);
        hi = t;
        lo += e + ol;
    }
    void normalise() {
        if (std::isfinite(hi) && std::isfinite(lo)) {
            const double t = hi + lo;
            lo = lo - (t - hi);
            hi = t;
        } else {
            lo = 0.0;
        }
    }
};

// One tile of a host-resident column (run_tiled, ma_pipeline.hip): its {sum | hi, lo, count} lands in record k.
template <typename T>
struct SumTile {
    ma_ctx* ctx;
    SumArgs base;       // the whole call's validity (words, bit_off) and mode
    bool masked;
    uint64_t* records;  // device, 4 words per tile
    size_t tile_rows;
    static ma_status run(void* user, size_t row0, size_t rows, void* const* ptrs) {
        const SumTile& t = *(const SumTile*)user;
        SumArgs a = t.base;
        a.data = ptrs[0];
        a.n = rows;
        if (t.masked) {
            const size_t bit = t.base.bit_off + row0;
            a.words = t.base.words + (bit >> 6);
            a.bit_off = bit & 63;
            a.last_word = (a.bit_off + rows - 1) >> 6;
        }
        uint64_t* rec = t.records + 4 * (row0 / t.tile_rows);
        a.out_a = rec;
        a.out_b = rec + 1;
        a.out_cnt = rec + 2;
        a.out_mean = nullptr;
        return enqueue_sum<T>(t.ctx, a, t.masked);
    }
};

template <typename T>
static ma_status sum_impl(ma_ctx* ctx, const T* data, size_t n, const uint8_t* mask_bits, size_t mask_bit_offset,
                          int64_t null_count, int mode, bool is_signed, void* out_a, void* out_b,
                          uint64_t* out_cnt, double* out_mean, uint64_t* stamp = nullptr, uint64_t stamp_value = 0,
                          uint64_t* early_stamp = nullptr) {
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    // a stamped launch (ma_scan_lanes_sum) is ONE launch on resident data whose stamps somebody waits for
    MA_REQUIRE(stamp == nullptr || n == 0 || pointer_kind(data) != kPageable, MA_ERR_INVALID_ARGUMENT,
               "a pipelined sum scans device-resident (or pinned) columns in place");
    MA_REQUIRE(stamp == nullptr || mask_bits == nullptr || pointer_kind(mask_bits) != kPageable, MA_ERR_INVALID_ARGUMENT,
               "a pipelined sum needs a device-reachable validity bitmap");
    MA_REQUIRE(n == 0 || data != nullptr, MA_ERR_INVALID_ARGUMENT, "data is NULL");
    MA_REQUIRE(((uintptr_t)data % sizeof(T)) == 0, MA_ERR_INVALID_ARGUMENT, "data pointer %p is not aligned to its element size",
               (const void*)data);
    MA_ENTER(ctx);
    MA_HIP(hipSetDevice(ctx->device));

    // The reference gates on the cached null count / all_true_mask before touching a mask
    // (src/kernels/arithmetic/simd.rs:144,454): no mask, or a mask known to be all-valid => dense kernel.
    const bool masked = mask_bits != nullptr && null_count != 0 && n != 0;

    // A host-resident (pageable) column crosses PCIe in tiles through the context's staging ring instead of a
    // temporary device copy of the whole column (ma_pipeline.hip): tile k's scan writes record k, the records are
    // folded in tile order — error-free for the double-double pairs, so the result keeps its 1-ULP bound.
    constexpr bool kFloat = std::is_floating_point<T>::value;
    const size_t stage_rows = (ctx->staging_tile_bytes / sizeof(T)) & ~(size_t)32767;
    if (stage_rows && n >= 2 * stage_rows && !ctx->capturing && pointer_kind(data) == kPageable) {
        const size_t n_tiles = (n + stage_rows - 1) / stage_rows;
        CallScope tscope(ctx);
        SumArgs base{};
        base.mode = 1;  // floats: keep (hi, lo) per tile
        base.is_signed = is_signed ? 1 : 0;
        if (masked) MA_TRY(tscope.in_mask(mask_bits, mask_bit_offset, n, &base.words, &base.bit_off));
        void* rec = nullptr;
        MA_TRY(ctx_scratch(ctx, n_tiles * 32, &rec));
        SumTile<T> call{ctx, base, masked, (uint64_t*)rec, stage_rows};
        PipeOperand op{data, nullptr, sizeof(T), true};
        MA_TRY(run_tiled(ctx, n, stage_rows, &op, 1, &SumTile<T>::run, &call));
        std::vector<uint64_t> host(n_tiles * 4);
        MA_HIP(hipMemcpy(host.data(), rec, n_tiles * 32, hipMemcpyDeviceToHost));
        HostFold f;
        for (size_t k = 0; k < n_tiles; ++k) f.add(host[4 * k], host[4 * k + 1], host[4 * k + 2], kFloat);
        f.normalise();
        uint64_t word_a = f.isum, word_b = 0;
        double as_double = is_signed ? (double)(int64_t)f.isum : (double)f.isum;
        if (kFloat) {
            as_double = f.hi;
            memcpy(&word_a, &f.hi, 8);
            memcpy(&word_b, &f.lo, 8);
        }
        const double mean = f.cnt ? as_double / (double)f.cnt : std::numeric_limits<double>::quiet_NaN();
        auto put = [&](void* user, const void* value) -> ma_status {
            if (!user) return MA_OK;
            if (pointer_kind(user) == kDevice) MA_HIP(hipMemcpy(user, value, 8, hipMemcpyHostToDevice));
            else memcpy(user, value, 8);
            return MA_OK;
        };
        MA_TRY(put(out_a, &word_a));
        if (mode == 1) MA_TRY(put(out_b, &word_b));
        MA_TRY(put(out_cnt, &f.cnt));
        MA_TRY(put(out_mean, &mean));
        return MA_OK;
    }

    CallScope scope(ctx);
    SumArgs a{};
    const void* d = nullptr;
    MA_TRY(scope.in(data, n * sizeof(T), &d));
    a.data = d;
    a.n = n;
    if (masked) {
        MA_TRY(scope.in_mask(mask_bits, mask_bit_offset, n, &a.words, &a.bit_off));
        a.last_word = (a.bit_off + n - 1) >> 6;
    }
    // Scalar outputs: directly into caller memory when it is device-reachable, else via the pinned slot.
    ResultSlot* slot = ctx->result;
    auto route = [&](void* user) -> bool { return user != nullptr && pointer_kind(user) != kPageable; };
    const bool direct_a = route(out_a), direct_b = route(out_b), direct_c = route(out_cnt), direct_m = route(out_mean);
    const bool any_slot = (out_a && !direct_a) || (out_b && !direct_b) || (out_cnt && !direct_c) || (out_mean && !direct_m);
    MA_REQUIRE(!(is_async(ctx) && any_slot), MA_ERR_INVALID_ARGUMENT,
               "async mode needs device-reachable (pinned or device) output pointers");
    a.out_a = out_a ? (direct_a ? (uint64_t*)out_a : &slot->a) : nullptr;
    a.out_b = out_b ? (direct_b ? (uint64_t*)out_b : &slot->b) : nullptr;
    a.out_cnt = out_cnt ? (direct_c ? out_cnt : &slot->cnt) : nullptr;
    double* mean_slot = (double*)&slot[1].a;
    a.out_mean = out_mean ? (direct_m ? out_mean : mean_slot) : nullptr;
    a.mode = mode;
    a.is_signed = is_signed ? 1 : 0;
    // A synchronous call on resident data does not go through hipStreamSynchronize (whose wake-up costs ~10 us, most
    // of a small call): the kernel's final thread stamps a pinned word after its results and the host polls that word —
    // for at most kPollUs, then it falls back to the blocking wait (a multi-millisecond scan must not burn a core).
    const bool poll = !is_async(ctx) && !scope.staged() && !ctx->capturing && ctx->poll_us > 0;
    volatile uint64_t* done = (volatile uint64_t*)&slot[2].a;
    if (poll) {
        a.done_word = (uint64_t*)done;
        a.done_seq = ++ctx->result_seq;
    }
    if (stamp) {  // enqueue-only by construction (the pipeline holds a NoSync scope): never together with the polled word
        MA_REQUIRE(!poll && !scope.staged(), MA_ERR_INVALID_ARGUMENT, "internal: a stamped sum must be enqueue-only on resident operands");
        a.done_word = stamp;
        a.done_seq = stamp_value;
        a.early_word = early_stamp;
    }
    MA_TRY(enqueue_sum<T>(ctx, a, masked));
    bool landed = false;
    if (poll) {
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned spins = 0;; ++spins) {
            if (*done == a.done_seq) {
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
    if (!landed) MA_TRY(end_call(ctx, scope));
    if (!is_async(ctx)) {
        if (out_a && !direct_a) memcpy(out_a, &slot->a, 8);
        if (out_b && !direct_b) memcpy(out_b, &slot->b, 8);
        if (out_cnt && !direct_c) *out_cnt = slot->cnt;
        if (out_mean && !direct_m) *out_mean = *mean_slot;
    }
    return MA_OK;
}

// ma_scan_lanes_sum (ma_scan_lanes.hip): one column of any numeric type, the launch stamped for the pipeline.
ma_status sum_stamped_any(ma_ctx* ctx, int32_t format_code, const void* data, size_t n, const uint8_t* mask_bits,
                          size_t mask_bit_offset, int64_t null_count, void* out_sum, double* out_lo, uint64_t* out_cnt,
                          uint64_t* stamp, uint64_t stamp_value, uint64_t* early_stamp) {
    const int mode = out_lo ? 1 : 0;
#define MA_STAMPED(T, SIGNED)                                                                                              \
    return sum_impl<T>(ctx, (const T*)data, n, mask_bits, mask_bit_offset, null_count, mode, SIGNED, out_sum, out_lo, out_cnt, \
                       nullptr, stamp, stamp_value, early_stamp)
    switch (format_code) {
        case 'c': MA_STAMPED(int8_t, true);
        case 'C': MA_STAMPED(uint8_t, false);
        case 's': MA_STAMPED(int16_t, true);
        case 'S': MA_STAMPED(uint16_t, false);
        case 'i': MA_STAMPED(int32_t, true);
        case 'I': MA_STAMPED(uint32_t, false);
        case 'l': MA_STAMPED(int64_t, true);
        case 'L': MA_STAMPED(uint64_t, false);
        case 'f': MA_STAMPED(float, true);
        case 'g': MA_STAMPED(double, true);
        default: break;
    }
#undef MA_STAMPED
    set_error("unsupported element format '%c' (numeric primitives only)", (char)format_code);
    return MA_ERR_UNSUPPORTED;
}

}  // namespace ma

using namespace ma;

extern "C" {

#define MA_DEFINE_INT_SUM(NAME, T, OUT_T, SIGNED)                                                                   \
    ma_status ma_##NAME##_sum(ma_ctx* ctx, const T* data, size_t n, const uint8_t* mask_bits, size_t mask_bit_offset, \
                              int64_t null_count, OUT_T* out_sum, uint64_t* out_valid_count) {                       \
        return sum_impl<T>(ctx, data, n, mask_bits, mask_bit_offset, null_count, 0, SIGNED, out_sum, nullptr,       \
                           out_valid_count, nullptr);                                                               \
    }                                                                                                               \
    ma_status ma_##NAME##_mean(ma_ctx* ctx, const T* data, size_t n, const uint8_t* mask_bits,                       \
                               size_t mask_bit_offset, int64_t null_count, double* out_mean,                        \
                               uint64_t* out_valid_count) {                                                         \
        return sum_impl<T>(ctx, data, n, mask_bits, mask_bit_offset, null_count, 0, SIGNED, nullptr, nullptr,       \
                           out_valid_count, out_mean);                                                              \
    }

MA_DEFINE_INT_SUM(i64, int64_t, int64_t, true)
MA_DEFINE_INT_SUM(u64, uint64_t, uint64_t, false)
MA_DEFINE_INT_SUM(i32, int32_t, int64_t, true)
MA_DEFINE_INT_SUM(u32, uint32_t, uint64_t, false)
// the reference's extended_numeric_types (src/enums/collections/numeric_array.rs:81-99; dispatch.rs:380-387)
MA_DEFINE_INT_SUM(i16, int16_t, int64_t, true)
MA_DEFINE_INT_SUM(u16, uint16_t, uint64_t, false)
MA_DEFINE_INT_SUM(i8, int8_t, int64_t, true)
MA_DEFINE_INT_SUM(u8, uint8_t, uint64_t, false)

#define MA_DEFINE_FLOAT_SUM(NAME, T)                                                                                \
    ma_status ma_##NAME##_sum(ma_ctx* ctx, const T* data, size_t n, const uint8_t* mask_bits, size_t mask_bit_offset, \
                              int64_t null_count, double* out_sum, uint64_t* out_valid_count) {                     \
        return sum_impl<T>(ctx, data, n, mask_bits, mask_bit_offset, null_count, 0, true, out_sum, nullptr,         \
                           out_valid_count, nullptr);                                                               \
    }                                                                                                               \
    ma_status ma_##NAME##_sum_dd(ma_ctx* ctx, const T* data, size_t n, const uint8_t* mask_bits,                     \
                                 size_t mask_bit_offset, int64_t null_count, double* out_hi, double* out_lo,        \
                                 uint64_t* out_valid_count) {                                                       \
        return sum_impl<T>(ctx, data, n, mask_bits, mask_bit_offset, null_count, 1, true, out_hi, out_lo,           \
                           out_valid_count, nullptr);                                                               \
    }                                                                                                               \
    ma_status ma_##NAME##_mean(ma_ctx* ctx, const T* data, size_t n, const uint8_t* mask_bits,                       \
                               size_t mask_bit_offset, int64_t null_count, double* out_mean,                        \
                               uint64_t* out_valid_count) {                                                         \
        return sum_impl<T>(ctx, data, n, mask_bits, mask_bit_offset, null_count, 0, true, nullptr, nullptr,         \
                           out_valid_count, out_mean);                                                              \
    }

MA_DEFINE_FLOAT_SUM(f64, double)
MA_DEFINE_FLOAT_SUM(f32, float)

}  // extern "C"
