/*
 * minarrow_oracle.c — CPU restatement of the reference's hot path (pbower/minarrow v0.10.1).
 *
 * TEST INFRASTRUCTURE ONLY. Nothing under minarrow_amd/ may import, link or call this file; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker / reported baseline.
 *
 * The reference is Rust (nightly, std::simd) and cannot be built in this environment (no rustc/cargo), so
 * its algorithms are restated here in plain C, function by function, each citing the reference file:line
 * (relative to the reference repository root) it follows. Facts about std::simd / Rust semantics that are
 * not visible in the reference's own sources are marked [ext].
 *
 * Parity pinning:
 *   - elementwise arithmetic, FMA, bitmask kernels, mask<->lane helpers: pinned against the reference's own
 *     known-answer tests (tests/golden/ JSON files, transcribed from src/kernels/arithmetic/mod.rs:117-537,
 *     src/kernels/bitmask/{mod,std,simd}.rs tests) by tests/test_oracle_golden.py.
 *   - sums: the reference never asserts a sum (its benches only print). Pinned against the closed forms of
 *     its bench inputs (sum(0..n)); otherwise PARITY UNPINNED.
 *   - Bitmask-gated sum / valid-count / mean: do not exist in the reference. Build-defined. PARITY UNPINNED.
 *
 * Build: `make -C oracle` (gcc, -O3 -march=native, -ffp-contract=off so that nothing is fused that the
 * reference does not fuse).
 */
#include <math.h>
#include <pthread.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MO_API __attribute__((visibility("default")))

/* =================================================================================================
 * Bitmask primitives — src/structs/bitmask.rs
 * ============================================================================================== */

/* Bitmask::get_unchecked — src/structs/bitmask.rs:745-748 */
static inline int mo_get_bit(const uint8_t* bits, size_t idx) { return (bits[idx >> 3] >> (idx & 7)) & 1; }

/* Bitmask::set_unchecked — src/structs/bitmask.rs:248-258 */
static inline void mo_set_bit(uint8_t* bits, size_t idx, int value) {
    uint8_t bit = (uint8_t)(1u << (idx & 7));
    if (value)
        bits[idx >> 3] |= bit;
    else
        bits[idx >> 3] &= (uint8_t)~bit;
}

/* =================================================================================================
 * Sums — the reference's only reductions live in its bench binaries
 * ============================================================================================== */

/* `for &v in slice { acc += v }` — benches/hotloop_benchmark_std.rs:49-57 (i64), :148-156 (f64).
 * Release-mode `+=` on i64 wraps; done in u64 here to keep C's overflow defined. */
MO_API int64_t mo_sum_scalar_i64(const int64_t* data, size_t n) {
    uint64_t acc = 0;
    for (size_t i = 0; i < n; ++i) acc += (uint64_t)data[i];
    return (int64_t)acc;
}

MO_API double mo_sum_scalar_f64(const double* data, size_t n) {
    double acc = 0.0;
    for (size_t i = 0; i < n; ++i) acc += data[i];
    return acc;
}

/* simd_sum_i64::<LANES> — benches/benchmark_parallel_simd.rs:44-59.
 * One LANES-wide accumulator, `reduce_sum`, scalar tail. Integer lanes wrap [ext].
 * The body is stamped out for the lane counts build.rs can emit (build.rs:67-110) so the compiler sees a
 * fixed-width inner loop, exactly like the monomorphised Rust. */
#define MO_SIMD_SUM_I64(L)                                                                   \
    static int64_t mo_simd_sum_i64_##L(const int64_t* data, size_t n) {                      \
        uint64_t acc[L];                                                                     \
        for (int l = 0; l < L; ++l) acc[l] = 0;                                              \
        size_t chunks = n / L;                                                               \
        for (size_t i = 0; i < chunks; ++i)                                                  \
            for (int l = 0; l < L; ++l) acc[l] += (uint64_t)data[i * L + (size_t)l];         \
        uint64_t result = 0;                                                                 \
        for (int l = 0; l < L; ++l) result += acc[l];                                        \
        for (size_t i = chunks * L; i < n; ++i) result += (uint64_t)data[i];                 \
        return (int64_t)result;                                                              \
    }
MO_SIMD_SUM_I64(1)
MO_SIMD_SUM_I64(2)
MO_SIMD_SUM_I64(4)
MO_SIMD_SUM_I64(8)
MO_SIMD_SUM_I64(16)

MO_API int64_t mo_simd_sum_i64(const int64_t* data, size_t n, int lanes) {
    switch (lanes) {
        case 1: return mo_simd_sum_i64_1(data, n);
        case 2: return mo_simd_sum_i64_2(data, n);
        case 4: return mo_simd_sum_i64_4(data, n);
        case 8: return mo_simd_sum_i64_8(data, n);
        case 16: return mo_simd_sum_i64_16(data, n);
        default: return mo_simd_sum_i64_4(data, n); /* benches use SIMD_LANES = 4 (:40) */
    }
}

/* simd_sum_f64::<LANES> — benches/benchmark_parallel_simd.rs:63-78.
 * `reduce_sum` on float vectors is an ordered left-to-right add starting from -0.0 [ext]. */
#define MO_SIMD_SUM_F64(L)                                                                   \
    static double mo_simd_sum_f64_##L(const double* data, size_t n) {                        \
        double acc[L];                                                                       \
        for (int l = 0; l < L; ++l) acc[l] = 0.0;                                            \
        size_t chunks = n / L;                                                               \
        for (size_t i = 0; i < chunks; ++i)                                                  \
            for (int l = 0; l < L; ++l) acc[l] += data[i * L + (size_t)l];                   \
        double result = -0.0;                                                                \
        for (int l = 0; l < L; ++l) result += acc[l];                                        \
        for (size_t i = chunks * L; i < n; ++i) result += data[i];                           \
        return result;                                                                       \
    }
MO_SIMD_SUM_F64(1)
MO_SIMD_SUM_F64(2)
MO_SIMD_SUM_F64(4)
MO_SIMD_SUM_F64(8)
MO_SIMD_SUM_F64(16)

MO_API double mo_simd_sum_f64(const double* data, size_t n, int lanes) {
    switch (lanes) {
        case 1: return mo_simd_sum_f64_1(data, n);
        case 2: return mo_simd_sum_f64_2(data, n);
        case 4: return mo_simd_sum_f64_4(data, n);
        case 8: return mo_simd_sum_f64_8(data, n);
        case 16: return mo_simd_sum_f64_16(data, n);
        default: return mo_simd_sum_f64_4(data, n);
    }
}

/* 4x-unrolled simd_sum_f64 — benches/hotloop_benchmark_simd.rs:117-174:
 * four LANES-wide accumulators over groups of 4 vectors, acc = acc1+acc2+acc3+acc4 (left to right),
 * leftover whole vectors added to acc, `result = 0.0; for i in 0..LANES { result += acc[i] }`, scalar tail. */
MO_API double mo_simd_sum_f64_unrolled4(const double* data, size_t n, int lanes) {
    double a1[64], a2[64], a3[64], a4[64], acc[64];
    if (lanes < 1) lanes = 1;
    if (lanes > 64) lanes = 64;
    size_t L = (size_t)lanes;
    for (size_t l = 0; l < L; ++l) a1[l] = a2[l] = a3[l] = a4[l] = 0.0;
    size_t simd_chunks = n / L;
    size_t unrolled = simd_chunks / 4;
    for (size_t i = 0; i < unrolled; ++i) {
        size_t base = i * 4 * L;
        for (size_t l = 0; l < L; ++l) {
            a1[l] += data[base + l];
            a2[l] += data[base + L + l];
            a3[l] += data[base + 2 * L + l];
            a4[l] += data[base + 3 * L + l];
        }
    }
    for (size_t l = 0; l < L; ++l) acc[l] = ((a1[l] + a2[l]) + a3[l]) + a4[l];
    for (size_t i = unrolled * 4; i < simd_chunks; ++i)
        for (size_t l = 0; l < L; ++l) acc[l] += data[i * L + l];
    double result = 0.0;
    for (size_t l = 0; l < L; ++l) result += acc[l];
    for (size_t i = simd_chunks * L; i < n; ++i) result += data[i];
    return result;
}

/* 4x-unrolled simd_sum_i64 — benches/hotloop_benchmark_simd.rs:56-114 (same shape; wrapping). */
MO_API int64_t mo_simd_sum_i64_unrolled4(const int64_t* data, size_t n, int lanes) {
    /* integer addition is associative and commutative modulo 2^64: any order gives the same bits */
    (void)lanes;
    return mo_sum_scalar_i64(data, n);
}

/* rayon_simd_sum_{i64,f64} — benches/benchmark_parallel_simd.rs:81-98:
 * `slice.par_chunks(1 << 20).map(simd_sum::<4>).sum()`. Rayon combines the per-chunk partials in an
 * unspecified tree order; this restatement combines them in chunk order (deterministic), single thread. */
MO_API int64_t mo_chunked_sum_i64(const int64_t* data, size_t n, size_t chunk, int lanes) {
    uint64_t total = 0;
    if (chunk == 0) chunk = (size_t)1 << 20;
    for (size_t off = 0; off < n; off += chunk) {
        size_t len = n - off < chunk ? n - off : chunk;
        total += (uint64_t)mo_simd_sum_i64(data + off, len, lanes);
    }
    return (int64_t)total;
}

MO_API double mo_chunked_sum_f64(const double* data, size_t n, size_t chunk, int lanes) {
    /* Iterator::sum::<f64>() starts from 0.0 [ext: std's float Sum impl folds from 0.0; newer std uses -0.0,
     * which differs only when every partial is -0.0] */
    double total = 0.0;
    if (chunk == 0) chunk = (size_t)1 << 20;
    for (size_t off = 0; off < n; off += chunk) {
        size_t len = n - off < chunk ? n - off : chunk;
        total += mo_simd_sum_f64(data + off, len, lanes);
    }
    return total;
}

/* ---- threaded form of the same thing: the CPU baseline bench.py times --------------------------- */

typedef struct {
    const void* data;
    size_t n, chunk;
    int lanes, is_f64;
    size_t n_chunks;
    size_t next; /* shared work counter (atomic) */
    void* partials; /* per-chunk results */
} mo_par_job;

static void* mo_par_worker(void* arg) {
    mo_par_job* job = (mo_par_job*)arg;
    for (;;) {
        size_t c = __atomic_fetch_add(&job->next, 1, __ATOMIC_RELAXED);
        if (c >= job->n_chunks) break;
        size_t off = c * job->chunk;
        size_t len = job->n - off < job->chunk ? job->n - off : job->chunk;
        if (job->is_f64)
            ((double*)job->partials)[c] = mo_simd_sum_f64((const double*)job->data + off, len, job->lanes);
        else
            ((int64_t*)job->partials)[c] = mo_simd_sum_i64((const int64_t*)job->data + off, len, job->lanes);
    }
    return NULL;
}

static int mo_par_run(mo_par_job* job, int n_threads) {
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 1024) n_threads = 1024;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)n_threads);
    if (!th) return -1;
    int started = 0;
    for (int t = 0; t < n_threads - 1; ++t) {
        if (pthread_create(&th[started], NULL, mo_par_worker, job) != 0) break;
        ++started;
    }
    mo_par_worker(job); /* the calling thread works too */
    for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
    free(th);
    return 0;
}

/* Work-sharing pool over chunks of `chunk` elements, partials combined in chunk order. */
MO_API int64_t mo_par_sum_i64(const int64_t* data, size_t n, size_t chunk, int lanes, int n_threads) {
    if (chunk == 0) chunk = (size_t)1 << 20;
    mo_par_job job;
    memset(&job, 0, sizeof(job));
    job.data = data;
    job.n = n;
    job.chunk = chunk;
    job.lanes = lanes;
    job.is_f64 = 0;
    job.n_chunks = (n + chunk - 1) / chunk;
    job.partials = calloc(job.n_chunks ? job.n_chunks : 1, sizeof(int64_t));
    if (!job.partials) return 0;
    mo_par_run(&job, n_threads);
    uint64_t total = 0;
    for (size_t c = 0; c < job.n_chunks; ++c) total += (uint64_t)((int64_t*)job.partials)[c];
    free(job.partials);
    return (int64_t)total;
}

MO_API double mo_par_sum_f64(const double* data, size_t n, size_t chunk, int lanes, int n_threads) {
    if (chunk == 0) chunk = (size_t)1 << 20;
    mo_par_job job;
    memset(&job, 0, sizeof(job));
    job.data = data;
    job.n = n;
    job.chunk = chunk;
    job.lanes = lanes;
    job.is_f64 = 1;
    job.n_chunks = (n + chunk - 1) / chunk;
    job.partials = calloc(job.n_chunks ? job.n_chunks : 1, sizeof(double));
    if (!job.partials) return 0.0;
    mo_par_run(&job, n_threads);
    double total = 0.0;
    for (size_t c = 0; c < job.n_chunks; ++c) total += ((double*)job.partials)[c];
    free(job.partials);
    return total;
}

/* Parallel fill used only to build the cpu_baseline inputs quickly: data[i] = start + i
 * (Vec64<i64> = (0..N).collect(), benches/benchmark_parallel_simd.rs:103,115). */
typedef struct {
    void* data;
    size_t n;
    int64_t start;
    int is_f64, tid, nth;
} mo_fill_job;

static void* mo_fill_worker(void* arg) {
    mo_fill_job* j = (mo_fill_job*)arg;
    size_t lo = j->n * (size_t)j->tid / (size_t)j->nth, hi = j->n * (size_t)(j->tid + 1) / (size_t)j->nth;
    if (j->is_f64)
        for (size_t i = lo; i < hi; ++i) ((double*)j->data)[i] = (double)(j->start + (int64_t)i);
    else
        for (size_t i = lo; i < hi; ++i) ((int64_t*)j->data)[i] = j->start + (int64_t)i;
    return NULL;
}

MO_API void mo_par_fill_iota(void* data, size_t n, int64_t start, int is_f64, int n_threads) {
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    pthread_t th[256];
    mo_fill_job jobs[256];
    int started = 0;
    for (int t = 0; t < n_threads; ++t) {
        jobs[t] = (mo_fill_job){data, n, start, is_f64, t, n_threads};
        if (t == n_threads - 1) {
            mo_fill_worker(&jobs[t]);
        } else if (pthread_create(&th[started], NULL, mo_fill_worker, &jobs[t]) == 0) {
            ++started;
        } else {
            mo_fill_worker(&jobs[t]);
        }
    }
    for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
}

/* ---- Bitmask-gated sums (BUILD-DEFINED: no reference implementation) -------------
 * sum_valid = sum of data[i] over rows whose validity bit (bit_offset + i) is set; count = popcount. */
MO_API void mo_masked_sum_i64(const int64_t* data, size_t n, const uint8_t* bits, size_t bit_offset, int64_t* out_sum,
                              uint64_t* out_count) {
    uint64_t acc = 0, cnt = 0;
    for (size_t i = 0; i < n; ++i) {
        if (mo_get_bit(bits, bit_offset + i)) {
            acc += (uint64_t)data[i];
            ++cnt;
        }
    }
    *out_sum = (int64_t)acc;
    *out_count = cnt;
}

MO_API void mo_masked_sum_i32(const int32_t* data, size_t n, const uint8_t* bits, size_t bit_offset, int64_t* out_sum,
                              uint64_t* out_count) {
    uint64_t acc = 0, cnt = 0;
    for (size_t i = 0; i < n; ++i) {
        if (mo_get_bit(bits, bit_offset + i)) {
            acc += (uint64_t)(int64_t)data[i];
            ++cnt;
        }
    }
    *out_sum = (int64_t)acc;
    *out_count = cnt;
}

/* Plain left-to-right masked f64 sum. The exactly rounded value the GPU is held to comes from
 * Python's math.fsum in the tests; this is the "what a scalar CPU loop gives" companion. */
MO_API void mo_masked_sum_f64(const double* data, size_t n, const uint8_t* bits, size_t bit_offset, double* out_sum,
                              uint64_t* out_count) {
    double acc = 0.0;
    uint64_t cnt = 0;
    for (size_t i = 0; i < n; ++i) {
        if (mo_get_bit(bits, bit_offset + i)) {
            acc += data[i];
            ++cnt;
        }
    }
    *out_sum = acc;
    *out_count = cnt;
}
