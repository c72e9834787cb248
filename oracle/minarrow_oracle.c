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
#ifndef _GNU_SOURCE
#define _GNU_SOURCE /* pthread_setaffinity_np, CPU_SET: pinning of the bench pool */
#endif
#include <math.h>
#include <pthread.h>
#include <sched.h>
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

/* A persistent worker pool (Rayon keeps its threads alive between calls too): workers sleep on a condition
 * variable, a call publishes one job, everybody — the caller included — claims chunks from the shared counter. */
static struct {
    pthread_mutex_t mu;
    pthread_cond_t cv_start, cv_done;
    pthread_t* threads;
    int n_workers;
    void* (*fn)(void*);
    void* arg;
    unsigned long generation;
    int remaining;
    int active;      /* workers with index < active take part in the current job */
} mo_pool = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, NULL, 0, NULL, NULL, 0, 0, 0};

/* bench.py's cpu_baseline pins the pool: worker k stays on the k-th CPU (mod their number) of the set the process may
 * run on — its cgroup's cpuset as sched_getaffinity reports it — so that the timed passes are not at the mercy of
 * migrations between the box's many idle cores (best-of-N 11 ms against a 16 ms median was the symptom). */
static int mo_pool_pinning = 0;
MO_API void mo_pool_set_pinning(int on) { mo_pool_pinning = on; }
static void mo_pin_self(int index) {
    cpu_set_t allowed;
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return;
    const int n = CPU_COUNT(&allowed);
    if (n <= 0) return;
    int want = index % n, seen = 0;
    for (int c = 0; c < CPU_SETSIZE; ++c)
        if (CPU_ISSET(c, &allowed) && seen++ == want) {
            cpu_set_t one;
            CPU_ZERO(&one);
            CPU_SET(c, &one);
            (void)pthread_setaffinity_np(pthread_self(), sizeof(one), &one);
            return;
        }
}

static void* mo_pool_worker(void* index_ptr) {
    const int index = (int)(intptr_t)index_ptr;
    unsigned long seen = 0;
    if (mo_pool_pinning) mo_pin_self(index + 1); /* the caller is thread 0 */
    for (;;) {
        pthread_mutex_lock(&mo_pool.mu);
        while (mo_pool.generation == seen || index >= mo_pool.active) {
            seen = mo_pool.generation;
            pthread_cond_wait(&mo_pool.cv_start, &mo_pool.mu);
        }
        seen = mo_pool.generation;
        void* (*fn)(void*) = mo_pool.fn;
        void* arg = mo_pool.arg;
        pthread_mutex_unlock(&mo_pool.mu);
        fn(arg);
        pthread_mutex_lock(&mo_pool.mu);
        if (--mo_pool.remaining == 0) pthread_cond_signal(&mo_pool.cv_done);
        pthread_mutex_unlock(&mo_pool.mu);
    }
    return NULL;
}

/* Runs fn(arg) on `n_threads` threads (n_threads - 1 pool workers + the caller) and waits for all of them. */
static int mo_pool_run(void* (*fn)(void*), void* arg, int n_threads) {
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 1024) n_threads = 1024;
    int want = n_threads - 1;
    pthread_mutex_lock(&mo_pool.mu);
    if (mo_pool.n_workers < want) {
        pthread_t* grown = (pthread_t*)realloc(mo_pool.threads, sizeof(pthread_t) * (size_t)want);
        if (grown) {
            mo_pool.threads = grown;
            while (mo_pool.n_workers < want) {
                if (pthread_create(&mo_pool.threads[mo_pool.n_workers], NULL, mo_pool_worker,
                                   (void*)(intptr_t)mo_pool.n_workers) != 0)
                    break;
                pthread_detach(mo_pool.threads[mo_pool.n_workers]);
                ++mo_pool.n_workers;
            }
        }
    }
    /* exactly `want` workers (or as many as exist) take part; the others go back to sleep */
    int workers = mo_pool.n_workers < want ? mo_pool.n_workers : want;
    mo_pool.fn = fn;
    mo_pool.arg = arg;
    mo_pool.active = workers;
    mo_pool.remaining = workers;
    ++mo_pool.generation;
    pthread_cond_broadcast(&mo_pool.cv_start);
    pthread_mutex_unlock(&mo_pool.mu);
    fn(arg);
    pthread_mutex_lock(&mo_pool.mu);
    while (mo_pool.remaining) pthread_cond_wait(&mo_pool.cv_done, &mo_pool.mu);
    pthread_mutex_unlock(&mo_pool.mu);
    return 0;
}

static int mo_par_run(mo_par_job* job, int n_threads) { return mo_pool_run(mo_par_worker, job, n_threads); }

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
    size_t n, chunk, n_chunks, next;
    int64_t start;
    int is_f64;
} mo_fill_job;

static void* mo_fill_worker(void* arg) {
    mo_fill_job* j = (mo_fill_job*)arg;
    for (;;) {
        size_t c = __atomic_fetch_add(&j->next, 1, __ATOMIC_RELAXED);
        if (c >= j->n_chunks) break;
        size_t lo = c * j->chunk, hi = lo + j->chunk < j->n ? lo + j->chunk : j->n;
        if (j->is_f64)
            for (size_t i = lo; i < hi; ++i) ((double*)j->data)[i] = (double)(j->start + (int64_t)i);
        else
            for (size_t i = lo; i < hi; ++i) ((int64_t*)j->data)[i] = j->start + (int64_t)i;
    }
    return NULL;
}

MO_API void mo_par_fill_iota(void* data, size_t n, int64_t start, int is_f64, int n_threads) {
    mo_fill_job job = {data, n, (size_t)1 << 20, (n + ((size_t)1 << 20) - 1) >> 20, 0, start, is_f64};
    mo_pool_run(mo_fill_worker, &job, n_threads);
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

/* The reference's extended_numeric_types (src/enums/collections/numeric_array.rs:81-99): the same build-defined scalar loop,
 * every element widened to 64 bits (sign- or zero-extended) before the wrapping add. `is_signed` picks the extension. */
#define MO_DEFINE_NARROW_MASKED_SUM(NAME, T, WIDEN)                                                              \
    MO_API void mo_masked_sum_##NAME(const T* data, size_t n, const uint8_t* bits, size_t bit_offset, int64_t* out_sum, \
                                     uint64_t* out_count) {                                                      \
        uint64_t acc = 0, cnt = 0;                                                                               \
        for (size_t i = 0; i < n; ++i) {                                                                         \
            if (bits == NULL || mo_get_bit(bits, bit_offset + i)) {                                              \
                acc += (uint64_t)(WIDEN)data[i];                                                                 \
                ++cnt;                                                                                           \
            }                                                                                                    \
        }                                                                                                        \
        *out_sum = (int64_t)acc;                                                                                 \
        *out_count = cnt;                                                                                        \
    }
MO_DEFINE_NARROW_MASKED_SUM(i8, int8_t, int64_t)
MO_DEFINE_NARROW_MASKED_SUM(u8, uint8_t, uint64_t)
MO_DEFINE_NARROW_MASKED_SUM(i16, int16_t, int64_t)
MO_DEFINE_NARROW_MASKED_SUM(u16, uint16_t, uint64_t)
MO_DEFINE_NARROW_MASKED_SUM(u32, uint32_t, uint64_t)

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

/* =================================================================================================
 * Elementwise arithmetic — src/kernels/arithmetic/{std,simd,dispatch}.rs
 *
 * Status codes returned by the restated kernels (the reference signals these by panicking / Err):
 *   MO_OK                0
 *   MO_LENGTH_MISMATCH   1   KernelError::LengthMismatch (src/utils.rs:163-171)
 *   MO_PANIC_DIV_ZERO    2   "Division by zero" / "Remainder by zero" / "Floor division by zero"
 *                            (std.rs:53-77; SIMD `/` `%` with a zero lane panics [ext])
 *   MO_PANIC_OVERFLOW    4   scalar `MIN / -1` or `MIN % -1` ("attempt to divide with overflow" [ext]).
 *                            The value written is the wrapping one (MIN, resp. 0), which is also what the
 *                            reference's SIMD lanes produce [ext]; bit 4 is OR-ed into the status so tests
 *                            can tell the two apart.
 * After a div-zero panic the reference's output is unobservable; this restatement stops at the first
 * offending element like the reference's loop does.
 * ============================================================================================== */

#define MO_OK 0
#define MO_LENGTH_MISMATCH 1
#define MO_PANIC_DIV_ZERO 2
#define MO_PANIC_OVERFLOW 4

/* ArithmeticOperator discriminants — src/enums/operators.rs:18-48 */
enum { MO_ADD = 0, MO_SUB = 1, MO_MUL = 2, MO_DIV = 3, MO_REM = 4, MO_POW = 5, MO_FLOORDIV = 6 };

/* is_simd_aligned — src/utils.rs:185-191 */
static inline int mo_is_simd_aligned(const void* p, size_t len) { return len == 0 || ((uintptr_t)p % 64) == 0; }

/* Bitmask::word_unchecked / set_word_unchecked — src/structs/bitmask.rs:266-277 (u64 view of the bytes) */
static inline uint64_t mo_word(const uint8_t* bits, size_t w) {
    uint64_t v;
    memcpy(&v, bits + 8 * w, 8);
    return v;
}
static inline void mo_set_word(uint8_t* bits, size_t w, uint64_t v) { memcpy(bits + 8 * w, &v, 8); }

/* Bitmask::mask_trailing_bits — src/structs/bitmask.rs:83-90 (bits has ceil(len/8) bytes) */
static inline void mo_mask_trailing_bits(uint8_t* bits, size_t len) {
    if (len == 0 || (len & 7) == 0) return;
    size_t last = (len + 7) / 8 - 1;
    bits[last] &= (uint8_t)((1u << (len & 7)) - 1);
}

/* Bitmask::new_set_all — src/structs/bitmask.rs:94-105 */
MO_API void mo_bitmask_new_set_all(uint8_t* bits, size_t len, int set) {
    size_t n_bytes = (len + 7) / 8;
    memset(bits, set ? 0xFF : 0, n_bytes);
    mo_mask_trailing_bits(bits, len);
}

/* Bitmask::fill — src/structs/bitmask.rs:728-734 */
static inline void mo_bitmask_fill(uint8_t* bits, size_t len, int value) { mo_bitmask_new_set_all(bits, len, value); }

/* Bitmask::count_ones — src/structs/bitmask.rs:393-405 */
MO_API size_t mo_bitmask_count_ones(const uint8_t* bits, size_t len) {
    size_t full = len / 8, count = 0;
    for (size_t i = 0; i < full; ++i) count += (size_t)__builtin_popcount(bits[i]);
    size_t rem = len & 7;
    if (rem) count += (size_t)__builtin_popcount(bits[full] & ((1u << rem) - 1));
    return count;
}

/* simd_mask::<_, N> — src/utils.rs:221-250: N validity bits starting at `offset`, lanes >= len cleared.
 * `mask_len` is the Bitmask's own `len` field (bounds the straddle read). Returns the lane bitmask. */
static inline uint64_t mo_simd_mask(const uint8_t* bits, size_t mask_len, size_t offset, size_t len, int n_lanes) {
    size_t word_idx = offset / 64;
    unsigned bit_shift = (unsigned)(offset % 64);
    uint64_t raw = mo_word(bits, word_idx) >> bit_shift;
    if (bit_shift > 0 && word_idx + 1 < (mask_len + 63) / 64) raw |= mo_word(bits, word_idx + 1) << (64 - bit_shift);
    size_t remaining = offset < len ? len - offset : 0;
    if (remaining < (size_t)n_lanes && remaining < 64) raw &= (((uint64_t)1) << remaining) - 1;
    /* Mask::from_bitmask keeps the low N bits [ext] */
    if (n_lanes < 64) raw &= (((uint64_t)1) << n_lanes) - 1;
    return raw;
}

/* write_simd_mask_bits — src/utils.rs:255-283: RMW of N bits at `offset` */
static inline void mo_write_simd_mask_bits(uint8_t* out_bits, size_t offset, uint64_t mbits, int n_lanes) {
    size_t word_idx = offset / 64;
    unsigned bit_shift = (unsigned)(offset % 64);
    uint64_t existing = mo_word(out_bits, word_idx);
    uint64_t lane_mask = n_lanes >= 64 ? ~(uint64_t)0 : ((((uint64_t)1) << n_lanes) - 1);
    uint64_t cleared = existing & ~(lane_mask << bit_shift);
    mo_set_word(out_bits, word_idx, cleared | (mbits << bit_shift));
    if (bit_shift > 0 && bit_shift + (unsigned)n_lanes > 64) {
        unsigned overflow_bits = (unsigned)n_lanes - (64 - bit_shift);
        uint64_t next_existing = mo_word(out_bits, word_idx + 1);
        uint64_t overflow_mask = (((uint64_t)1) << overflow_bits) - 1;
        mo_set_word(out_bits, word_idx + 1, (next_existing & ~overflow_mask) | (mbits >> (64 - bit_shift)));
    }
}

/* exported for the helper tests (src/utils.rs has no tests of its own for these two) */
MO_API uint64_t mo_simd_mask_bits(const uint8_t* bits, size_t mask_len, size_t offset, size_t len, int n_lanes) {
    return mo_simd_mask(bits, mask_len, offset, len, n_lanes);
}
MO_API void mo_write_mask_bits(uint8_t* out_bits, size_t offset, uint64_t mbits, int n_lanes) {
    mo_write_simd_mask_bits(out_bits, offset, mbits, n_lanes);
}

/* all_true_mask_simd::<LANES> — src/kernels/bitmask/simd.rs:648-692.
 * Note the quirk restated here: when n_words % LANES == 0 the last (partial) word is compared against
 * all-ones inside the SIMD loop, so a fully valid mask whose len is not a multiple of 64 reports false
 * (trailing bits are zero). Callers then take the masked path; results are identical. */
MO_API int mo_all_true_mask_simd(const uint8_t* bits, size_t len, int lanes) {
    if (len == 0) return 1;
    if (len < 64) {
        uint64_t w = mo_word(bits, 0), valid = (((uint64_t)1) << len) - 1;
        return (w & valid) == valid;
    }
    size_t n_words = (len + 63) / 64, simd_chunks = n_words / (size_t)lanes;
    for (size_t c = 0; c < simd_chunks; ++c)
        for (int l = 0; l < lanes; ++l)
            if (mo_word(bits, c * (size_t)lanes + (size_t)l) != ~(uint64_t)0) return 0;
    size_t tail_words = n_words % (size_t)lanes, base = simd_chunks * (size_t)lanes;
    for (size_t k = 0; k < tail_words; ++k) {
        if (base + k == n_words - 1 && len % 64 != 0) {
            uint64_t slack = (((uint64_t)1) << (len % 64)) - 1;
            if (mo_word(bits, base + k) != slack) return 0;
        } else if (mo_word(bits, base + k) != ~(uint64_t)0) {
            return 0;
        }
    }
    return 1;
}

/* all_false_mask_simd::<LANES> — src/kernels/bitmask/simd.rs:695-736 */
MO_API int mo_all_false_mask_simd(const uint8_t* bits, size_t len, int lanes) {
    if (len == 0) return 1;
    if (len < 64) {
        uint64_t w = mo_word(bits, 0), valid = (((uint64_t)1) << len) - 1;
        return (w & valid) == 0;
    }
    size_t n_words = (len + 63) / 64, simd_chunks = n_words / (size_t)lanes;
    for (size_t c = 0; c < simd_chunks; ++c)
        for (int l = 0; l < lanes; ++l)
            if (mo_word(bits, c * (size_t)lanes + (size_t)l) != 0) return 0;
    size_t tail_words = n_words % (size_t)lanes, base = simd_chunks * (size_t)lanes;
    for (size_t k = 0; k < tail_words; ++k) {
        if (base + k == n_words - 1 && len % 64 != 0) {
            uint64_t slack = (((uint64_t)1) << (len % 64)) - 1;
            if (mo_word(bits, base + k) & slack) return 0;
        } else if (mo_word(bits, base + k) != 0) {
            return 0;
        }
    }
    return 1;
}

/* all_true_mask / all_false_mask (scalar) — src/kernels/bitmask/std.rs:300-366 */
MO_API int mo_all_true_mask_std(const uint8_t* bits, size_t len) {
    if (len == 0) return 1;
    if (len < 64) {
        for (size_t i = 0; i < len; ++i)
            if (!mo_get_bit(bits, i)) return 0;
        return 1;
    }
    size_t n_words = (len + 63) >> 6, trailing = len & 63;
    for (size_t i = 0; i < n_words; ++i) {
        uint64_t w = mo_word(bits, i);
        if (i == n_words - 1 && trailing != 0) {
            uint64_t m = (((uint64_t)1) << trailing) - 1;
            if ((w & m) != m) return 0;
        } else if (w != ~(uint64_t)0) {
            return 0;
        }
    }
    return 1;
}

MO_API int mo_all_false_mask_std(const uint8_t* bits, size_t len) {
    if (len < 64) {
        for (size_t i = 0; i < len; ++i)
            if (mo_get_bit(bits, i)) return 0;
        return 1;
    }
    size_t n_words = (len + 63) >> 6, trailing = len & 63;
    for (size_t i = 0; i < n_words; ++i) {
        uint64_t w = mo_word(bits, i);
        if (i == n_words - 1 && trailing != 0) {
            uint64_t m = (((uint64_t)1) << trailing) - 1;
            if (w & m) return 0;
        } else if (w != 0) {
            return 0;
        }
    }
    return 1;
}

/* ---- integer element functions ------------------------------------------------------------------- */

/* `x.pow(exp)` (num-traits PrimInt::pow -> core pow; release build wraps [ext]) and the dense SIMD tail's
 * repeated `wrapping_mul` (simd.rs:94-101) agree modulo 2^bits; computed here by squaring. */
#define MO_DEFINE_INT(NAME, T, UT, IS_SIGNED, TMIN)                                                              \
    static inline T mo_pow_##NAME(T base, uint32_t exp) {                                                       \
        UT acc = 1, b = (UT)base;                                                                               \
        while (exp) {                                                                                           \
            if (exp & 1) acc = (UT)(acc * b);                                                                   \
            b = (UT)(b * b);                                                                                    \
            exp >>= 1;                                                                                          \
        }                                                                                                       \
        return (T)acc;                                                                                          \
    }                                                                                                           \
    /* rhs.to_u32().unwrap_or(0) — std.rs:67, simd.rs:96,166,296 */                                              \
    static inline uint32_t mo_exp_##NAME(T e) {                                                                 \
        if (IS_SIGNED && e < (T)0) return 0;                                                                    \
        if ((uint64_t)e > 0xFFFFFFFFull) return 0;                                                              \
        return (uint32_t)e;                                                                                     \
    }                                                                                                           \
    /* one element, divisor known non-zero; *st |= MO_PANIC_OVERFLOW on MIN / -1 */                              \
    static inline T mo_div_##NAME(T a, T b, int* st) {                                                          \
        if (IS_SIGNED && a == (T)(TMIN) && b == (T)-1) {                                                        \
            *st |= MO_PANIC_OVERFLOW;                                                                           \
            return a;                                                                                           \
        }                                                                                                       \
        return (T)(a / b);                                                                                      \
    }                                                                                                           \
    static inline T mo_rem_##NAME(T a, T b, int* st) {                                                          \
        if (IS_SIGNED && a == (T)(TMIN) && b == (T)-1) {                                                        \
            *st |= MO_PANIC_OVERFLOW;                                                                           \
            return 0;                                                                                           \
        }                                                                                                       \
        return (T)(a % b);                                                                                      \
    }                                                                                                           \
    /* FloorDiv — std.rs:68-77: trunc quotient, minus one when the remainder is non-zero and signs differ */    \
    static inline T mo_floordiv_##NAME(T a, T b, int* st) {                                                     \
        T d = mo_div_##NAME(a, b, st);                                                                          \
        T r = mo_rem_##NAME(a, b, st);                                                                          \
        if (r != 0 && IS_SIGNED && ((T)(a ^ b)) < (T)0) return (T)((UT)d - (UT)1);                              \
        return d;                                                                                               \
    }                                                                                                           \
                                                                                                                \
    /* int_dense_body_std — src/kernels/arithmetic/std.rs:41-80 */                                               \
    MO_API int mo_int_dense_std_##NAME(int op, const T* lhs, const T* rhs, T* out, size_t n) {                   \
        int st = MO_OK;                                                                                         \
        for (size_t i = 0; i < n; ++i) {                                                                        \
            switch (op) {                                                                                       \
                case MO_ADD: out[i] = (T)((UT)lhs[i] + (UT)rhs[i]); break;                                      \
                case MO_SUB: out[i] = (T)((UT)lhs[i] - (UT)rhs[i]); break;                                      \
                case MO_MUL: out[i] = (T)((UT)lhs[i] * (UT)rhs[i]); break;                                      \
                case MO_DIV:                                                                                    \
                    if (rhs[i] == 0) return st | MO_PANIC_DIV_ZERO;                                             \
                    out[i] = mo_div_##NAME(lhs[i], rhs[i], &st);                                                \
                    break;                                                                                      \
                case MO_REM:                                                                                    \
                    if (rhs[i] == 0) return st | MO_PANIC_DIV_ZERO;                                             \
                    out[i] = mo_rem_##NAME(lhs[i], rhs[i], &st);                                                \
                    break;                                                                                      \
                case MO_POW: out[i] = mo_pow_##NAME(lhs[i], mo_exp_##NAME(rhs[i])); break;                       \
                case MO_FLOORDIV:                                                                               \
                    if (rhs[i] == 0) return st | MO_PANIC_DIV_ZERO;                                             \
                    out[i] = mo_floordiv_##NAME(lhs[i], rhs[i], &st);                                           \
                    break;                                                                                      \
                default: return -1;                                                                             \
            }                                                                                                   \
        }                                                                                                       \
        return st;                                                                                              \
    }                                                                                                           \
                                                                                                                \
    /* int_masked_body_std — src/kernels/arithmetic/std.rs:86-138 */                                             \
    MO_API int mo_int_masked_std_##NAME(int op, const T* lhs, const T* rhs, const uint8_t* mask, T* out,         \
                                        uint8_t* out_mask, size_t n) {                                          \
        int st = MO_OK;                                                                                         \
        for (size_t i = 0; i < n; ++i) {                                                                        \
            if (mo_get_bit(mask, i)) {                                                                          \
                T result = 0;                                                                                   \
                int final_valid = 1;                                                                            \
                switch (op) {                                                                                   \
                    case MO_ADD: result = (T)((UT)lhs[i] + (UT)rhs[i]); break;                                  \
                    case MO_SUB: result = (T)((UT)lhs[i] - (UT)rhs[i]); break;                                  \
                    case MO_MUL: result = (T)((UT)lhs[i] * (UT)rhs[i]); break;                                  \
                    case MO_DIV:                                                                                \
                        if (rhs[i] == 0) { result = 0; final_valid = 0; }                                       \
                        else result = mo_div_##NAME(lhs[i], rhs[i], &st);                                       \
                        break;                                                                                  \
                    case MO_REM:                                                                                \
                        if (rhs[i] == 0) { result = 0; final_valid = 0; }                                       \
                        else result = mo_rem_##NAME(lhs[i], rhs[i], &st);                                       \
                        break;                                                                                  \
                    case MO_POW: result = mo_pow_##NAME(lhs[i], mo_exp_##NAME(rhs[i])); break;                   \
                    case MO_FLOORDIV:                                                                           \
                        if (rhs[i] == 0) { result = 0; final_valid = 0; }                                       \
                        else result = mo_floordiv_##NAME(lhs[i], rhs[i], &st);                                  \
                        break;                                                                                  \
                    default: return -1;                                                                         \
                }                                                                                               \
                out[i] = result;                                                                                \
                mo_set_bit(out_mask, i, final_valid);                                                           \
            } else {                                                                                            \
                out[i] = 0;                                                                                     \
                mo_set_bit(out_mask, i, 0);                                                                     \
            }                                                                                                   \
        }                                                                                                       \
        return st;                                                                                              \
    }                                                                                                           \
                                                                                                                \
    /* int_dense_body_simd::<T, LANES> — src/kernels/arithmetic/simd.rs:52-113 */                                \
    MO_API int mo_int_dense_simd_##NAME(int op, const T* lhs, const T* rhs, T* out, size_t n, int lanes) {       \
        int st = MO_OK;                                                                                         \
        size_t L = (size_t)lanes, vectorisable = n / L * L, i = 0;                                              \
        while (i < vectorisable) {                                                                              \
            if (op == MO_POW || op == MO_FLOORDIV) { vectorisable = 0; break; } /* simd.rs:77-80 */             \
            if (op == MO_DIV || op == MO_REM)                                                                   \
                for (size_t l = 0; l < L; ++l)                                                                  \
                    if (rhs[i + l] == 0) return st | MO_PANIC_DIV_ZERO; /* any zero lane panics [ext] */        \
            for (size_t l = 0; l < L; ++l) {                                                                    \
                T a = lhs[i + l], b = rhs[i + l];                                                               \
                int lane_st = 0;                                                                                \
                switch (op) {                                                                                   \
                    case MO_ADD: out[i + l] = (T)((UT)a + (UT)b); break;                                        \
                    case MO_SUB: out[i + l] = (T)((UT)a - (UT)b); break;                                        \
                    case MO_MUL: out[i + l] = (T)((UT)a * (UT)b); break;                                        \
                    /* SIMD lanes: MIN / -1 = MIN, MIN % -1 = 0, no panic [ext] */                              \
                    case MO_DIV: out[i + l] = mo_div_##NAME(a, b, &lane_st); break;                             \
                    case MO_REM: out[i + l] = mo_rem_##NAME(a, b, &lane_st); break;                             \
                    default: return -1;                                                                         \
                }                                                                                               \
            }                                                                                                   \
            i += L;                                                                                             \
        }                                                                                                       \
        for (size_t idx = vectorisable; idx < n; ++idx) { /* scalar tail, simd.rs:87-112 */                      \
            switch (op) {                                                                                       \
                case MO_ADD: out[idx] = (T)((UT)lhs[idx] + (UT)rhs[idx]); break; /* release: wraps */           \
                case MO_SUB: out[idx] = (T)((UT)lhs[idx] - (UT)rhs[idx]); break;                                \
                case MO_MUL: out[idx] = (T)((UT)lhs[idx] * (UT)rhs[idx]); break;                                \
                case MO_DIV:                                                                                    \
                    if (rhs[idx] == 0) return st | MO_PANIC_DIV_ZERO;                                           \
                    out[idx] = mo_div_##NAME(lhs[idx], rhs[idx], &st);                                          \
                    break;                                                                                      \
                case MO_REM:                                                                                    \
                    if (rhs[idx] == 0) return st | MO_PANIC_DIV_ZERO;                                           \
                    out[idx] = mo_rem_##NAME(lhs[idx], rhs[idx], &st);                                          \
                    break;                                                                                      \
                case MO_POW: out[idx] = mo_pow_##NAME(lhs[idx], mo_exp_##NAME(rhs[idx])); break;                 \
                case MO_FLOORDIV:                                                                               \
                    if (rhs[idx] == 0) return st | MO_PANIC_DIV_ZERO;                                           \
                    out[idx] = mo_floordiv_##NAME(lhs[idx], rhs[idx], &st);                                     \
                    break;                                                                                      \
                default: return -1;                                                                             \
            }                                                                                                   \
        }                                                                                                       \
        return st;                                                                                              \
    }                                                                                                           \
                                                                                                                \
    /* int_masked_body_simd::<T, LANES> — src/kernels/arithmetic/simd.rs:118-370.                                \
     * `mask_len` is mask.len; out_mask arrives as Bitmask::new_set_all(n, true) (dispatch.rs:92). */            \
    MO_API int mo_int_masked_simd_##NAME(int op, const T* lhs, const T* rhs, const uint8_t* mask,                \
                                         size_t mask_len, T* out, uint8_t* out_mask, size_t n, int lanes) {     \
        int st = MO_OK;                                                                                         \
        size_t L = (size_t)lanes;                                                                               \
        int dense = mo_all_true_mask_simd(mask, mask_len, lanes); /* simd.rs:144 */                              \
        if (dense) { /* simd.rs:151-266 */                                                                       \
            size_t vectorisable = n / L * L, i = 0;                                                             \
            while (i < vectorisable) {                                                                          \
                uint64_t valid = L >= 64 ? ~(uint64_t)0 : ((((uint64_t)1) << L) - 1);                            \
                for (size_t l = 0; l < L; ++l) {                                                                \
                    T a = lhs[i + l], b = rhs[i + l], r = 0;                                                    \
                    int lane_st = 0;                                                                            \
                    switch (op) {                                                                               \
                        case MO_ADD: r = (T)((UT)a + (UT)b); break;                                             \
                        case MO_SUB: r = (T)((UT)a - (UT)b); break;                                             \
                        case MO_MUL: r = (T)((UT)a * (UT)b); break;                                             \
                        case MO_POW: r = mo_pow_##NAME(a, mo_exp_##NAME(b)); break;                              \
                        case MO_DIV:                                                                            \
                        case MO_REM:                                                                            \
                            if (b == 0) { r = 0; valid &= ~(((uint64_t)1) << l); }                               \
                            else r = op == MO_DIV ? mo_div_##NAME(a, b, &lane_st) : mo_rem_##NAME(a, b, &lane_st); \
                            break;                                                                              \
                        case MO_FLOORDIV:                                                                       \
                            if (b == 0) { r = 0; valid &= ~(((uint64_t)1) << l); }                               \
                            else r = mo_floordiv_##NAME(a, b, &st); /* per-lane scalar code: can panic */       \
                            break;                                                                              \
                        default: return -1;                                                                     \
                    }                                                                                           \
                    out[i + l] = r;                                                                             \
                }                                                                                               \
                mo_write_simd_mask_bits(out_mask, i, valid, lanes);                                             \
                i += L;                                                                                         \
            }                                                                                                   \
            for (size_t idx = vectorisable; idx < n; ++idx) {                                                   \
                T a = lhs[idx], b = rhs[idx];                                                                   \
                switch (op) {                                                                                   \
                    case MO_ADD: out[idx] = (T)((UT)a + (UT)b); mo_set_bit(out_mask, idx, 1); break;            \
                    case MO_SUB: out[idx] = (T)((UT)a - (UT)b); mo_set_bit(out_mask, idx, 1); break;            \
                    case MO_MUL: out[idx] = (T)((UT)a * (UT)b); mo_set_bit(out_mask, idx, 1); break;            \
                    case MO_POW: out[idx] = mo_pow_##NAME(a, mo_exp_##NAME(b)); mo_set_bit(out_mask, idx, 1); break; \
                    case MO_DIV:                                                                                \
                    case MO_REM:                                                                                \
                        if (b == 0) { out[idx] = 0; mo_set_bit(out_mask, idx, 0); }                              \
                        else {                                                                                  \
                            out[idx] = op == MO_DIV ? mo_div_##NAME(a, b, &st) : mo_rem_##NAME(a, b, &st);      \
                            mo_set_bit(out_mask, idx, 1);                                                       \
                        }                                                                                       \
                        break;                                                                                  \
                    case MO_FLOORDIV:                                                                           \
                        if (b == 0) { out[idx] = 0; mo_set_bit(out_mask, idx, 0); }                              \
                        else { out[idx] = mo_floordiv_##NAME(a, b, &st); mo_set_bit(out_mask, idx, 1); }        \
                        break;                                                                                  \
                    default: return -1;                                                                         \
                }                                                                                               \
            }                                                                                                   \
            return st;                                                                                          \
        }                                                                                                       \
        size_t i = 0;                                                                                           \
        while (i + L <= n) { /* simd.rs:269-328 */                                                               \
            uint64_t m_src = mo_simd_mask(mask, mask_len, i, n, lanes);                                         \
            uint64_t div_zero = 0;                                                                              \
            for (size_t l = 0; l < L; ++l)                                                                      \
                if (rhs[i + l] == 0) div_zero |= ((uint64_t)1) << l;                                            \
            for (size_t l = 0; l < L; ++l) {                                                                    \
                T a = lhs[i + l], b = rhs[i + l], r = 0;                                                        \
                int lane_st = 0;                                                                                \
                switch (op) {                                                                                   \
                    case MO_ADD: r = (T)((UT)a + (UT)b); break;                                                 \
                    case MO_SUB: r = (T)((UT)a - (UT)b); break;                                                 \
                    case MO_MUL: r = (T)((UT)a * (UT)b); break;                                                 \
                    case MO_DIV: r = b == 0 ? 0 : mo_div_##NAME(a, b, &lane_st); break; /* safe_b, simd.rs:283 */ \
                    case MO_REM: r = b == 0 ? 0 : mo_rem_##NAME(a, b, &lane_st); break;                         \
                    case MO_POW: r = mo_pow_##NAME(a, mo_exp_##NAME(b)); break;                                  \
                    case MO_FLOORDIV: r = b == 0 ? 0 : mo_floordiv_##NAME(a, b, &st); break; /* all lanes, valid or not */ \
                    default: return -1;                                                                         \
                }                                                                                               \
                out[i + l] = ((m_src >> l) & 1) ? r : 0; /* simd.rs:315 */                                       \
            }                                                                                                   \
            uint64_t final_mask = (op == MO_DIV || op == MO_REM || op == MO_FLOORDIV) ? (m_src & ~div_zero) : m_src; \
            mo_write_simd_mask_bits(out_mask, i, final_mask, lanes);                                            \
            i += L;                                                                                             \
        }                                                                                                       \
        for (size_t j = i; j < n; ++j) { /* simd.rs:331-369 */                                                   \
            if (mo_get_bit(mask, j)) {                                                                          \
                T a = lhs[j], b = rhs[j], result = 0;                                                           \
                int final_valid = 1;                                                                            \
                switch (op) {                                                                                   \
                    case MO_ADD: result = (T)((UT)a + (UT)b); break;                                            \
                    case MO_SUB: result = (T)((UT)a - (UT)b); break;                                            \
                    case MO_MUL: result = (T)((UT)a * (UT)b); break;                                            \
                    case MO_DIV: if (b == 0) final_valid = 0; else result = mo_div_##NAME(a, b, &st); break;    \
                    case MO_REM: if (b == 0) final_valid = 0; else result = mo_rem_##NAME(a, b, &st); break;    \
                    case MO_POW: result = mo_pow_##NAME(a, mo_exp_##NAME(b)); break;                             \
                    case MO_FLOORDIV: if (b == 0) final_valid = 0; else result = mo_floordiv_##NAME(a, b, &st); break; \
                    default: return -1;                                                                         \
                }                                                                                               \
                out[j] = result;                                                                                \
                mo_set_bit(out_mask, j, final_valid);                                                           \
            } else {                                                                                            \
                out[j] = 0;                                                                                     \
                mo_set_bit(out_mask, j, 0);                                                                     \
            }                                                                                                   \
        }                                                                                                       \
        return st;                                                                                              \
    }                                                                                                           \
                                                                                                                \
    /* apply_int_<t> — src/kernels/arithmetic/dispatch.rs:65-133 (macro), :376-387 (instances).                  \
     * mask == NULL <=> None. out_mask (when masked) is initialised here as new_set_all(len, true) (:92).        \
     * Returns MO_LENGTH_MISMATCH or the body's status. `used_simd` reports which body ran. */                   \
    MO_API int mo_apply_int_##NAME(const T* lhs, size_t lhs_len, const T* rhs, size_t rhs_len, int op,           \
                                   const uint8_t* mask, size_t mask_len, T* out, uint8_t* out_mask, int lanes,  \
                                   int* used_simd) {                                                            \
        if (lhs_len != rhs_len) return MO_LENGTH_MISMATCH; /* confirm_equal_len, dispatch.rs:81 */               \
        size_t len = lhs_len;                                                                                   \
        int simd = mo_is_simd_aligned(lhs, len) && mo_is_simd_aligned(rhs, len); /* dispatch.rs:86 */            \
        if (used_simd) *used_simd = simd;                                                                       \
        if (mask) {                                                                                             \
            mo_bitmask_new_set_all(out_mask, len, 1);                                                           \
            return simd ? mo_int_masked_simd_##NAME(op, lhs, rhs, mask, mask_len, out, out_mask, len, lanes)    \
                        : mo_int_masked_std_##NAME(op, lhs, rhs, mask, out, out_mask, len);                     \
        }                                                                                                       \
        return simd ? mo_int_dense_simd_##NAME(op, lhs, rhs, out, len, lanes)                                   \
                    : mo_int_dense_std_##NAME(op, lhs, rhs, out, len);                                          \
    }

MO_DEFINE_INT(i8, int8_t, uint8_t, 1, INT8_MIN)
MO_DEFINE_INT(i16, int16_t, uint16_t, 1, INT16_MIN)
MO_DEFINE_INT(i32, int32_t, uint32_t, 1, INT32_MIN)
MO_DEFINE_INT(i64, int64_t, uint64_t, 1, INT64_MIN)
MO_DEFINE_INT(u8, uint8_t, uint8_t, 0, 0)
MO_DEFINE_INT(u16, uint16_t, uint16_t, 0, 0)
MO_DEFINE_INT(u32, uint32_t, uint32_t, 0, 0)
MO_DEFINE_INT(u64, uint64_t, uint64_t, 0, 0)

/* ---- floating point ------------------------------------------------------------------------------
 * float_dense_body_std / float_masked_body_std — src/kernels/arithmetic/std.rs:144-194
 * float_{dense,masked}_body_f{32,64}_simd      — src/kernels/arithmetic/simd.rs:376-589
 * Per element the SIMD and scalar bodies compute the same IEEE operations (lane-wise + - * / are correctly
 * rounded; `%` is fmod; FloorDiv is floor(a/b)); Power is exp(b * ln(a)) through libm in both
 * (std.rs:153, simd.rs:570,585) — std::simd's ln/exp call the scalar libm per lane [ext], so results depend
 * on the platform libm to the last ulp and the reference's own tests use 1e-6 / 1e-12 tolerances
 * (src/kernels/arithmetic/mod.rs:369-370). */
#define MO_DEFINE_FLOAT(NAME, T, FMOD, EXP, LOG, FLOOR, FMA)                                                    \
    static inline T mo_fop_##NAME(int op, T a, T b) {                                                           \
        switch (op) {                                                                                           \
            case MO_ADD: return a + b;                                                                          \
            case MO_SUB: return a - b;                                                                          \
            case MO_MUL: return a * b;                                                                          \
            case MO_DIV: return a / b;                                                                          \
            case MO_REM: return FMOD(a, b);                                                                     \
            case MO_POW: return EXP(b * LOG(a));                                                                \
            case MO_FLOORDIV: return FLOOR(a / b);                                                              \
            default: return (T)NAN;                                                                             \
        }                                                                                                       \
    }                                                                                                           \
    /* float_dense_body_std — std.rs:144-157; the SIMD twin (simd.rs:511-589) is element-for-element equal */   \
    MO_API int mo_float_dense_##NAME(int op, const T* lhs, const T* rhs, T* out, size_t n) {                     \
        if (op < 0 || op > MO_FLOORDIV) return -1;                                                              \
        for (size_t i = 0; i < n; ++i) out[i] = mo_fop_##NAME(op, lhs[i], rhs[i]);                              \
        return MO_OK;                                                                                           \
    }                                                                                                           \
    /* float_masked_body_std — std.rs:163-194 */                                                                 \
    MO_API int mo_float_masked_std_##NAME(int op, const T* lhs, const T* rhs, const uint8_t* mask, T* out,       \
                                          uint8_t* out_mask, size_t n) {                                        \
        if (op < 0 || op > MO_FLOORDIV) return -1;                                                              \
        for (size_t i = 0; i < n; ++i) {                                                                        \
            if (mo_get_bit(mask, i)) {                                                                          \
                out[i] = mo_fop_##NAME(op, lhs[i], rhs[i]);                                                     \
                mo_set_bit(out_mask, i, 1);                                                                     \
            } else {                                                                                            \
                out[i] = (T)0;                                                                                  \
                mo_set_bit(out_mask, i, 0);                                                                     \
            }                                                                                                   \
        }                                                                                                       \
        return MO_OK;                                                                                           \
    }                                                                                                           \
    /* float_masked_body_f{32,64}_simd — simd.rs:376-505: all-valid => dense body + out_mask.fill(true);        \
     * else per LANES: select(valid, res, 0.0), out-mask = input mask; scalar tail. */                           \
    MO_API int mo_float_masked_simd_##NAME(int op, const T* lhs, const T* rhs, const uint8_t* mask,              \
                                           size_t mask_len, T* out, uint8_t* out_mask, size_t n, int lanes) {   \
        if (op < 0 || op > MO_FLOORDIV) return -1;                                                              \
        size_t L = (size_t)lanes;                                                                               \
        if (mo_all_true_mask_simd(mask, mask_len, lanes)) {                                                     \
            mo_float_dense_##NAME(op, lhs, rhs, out, n);                                                        \
            mo_bitmask_fill(out_mask, n, 1);                                                                    \
            return MO_OK;                                                                                       \
        }                                                                                                       \
        size_t i = 0;                                                                                           \
        while (i + L <= n) {                                                                                    \
            uint64_t m = mo_simd_mask(mask, mask_len, i, n, lanes);                                             \
            for (size_t l = 0; l < L; ++l) {                                                                    \
                T res = mo_fop_##NAME(op, lhs[i + l], rhs[i + l]);                                              \
                out[i + l] = ((m >> l) & 1) ? res : (T)0;                                                       \
            }                                                                                                   \
            mo_write_simd_mask_bits(out_mask, i, m, lanes);                                                     \
            i += L;                                                                                             \
        }                                                                                                       \
        for (size_t j = i; j < n; ++j) {                                                                        \
            if (mo_get_bit(mask, j)) {                                                                          \
                out[j] = mo_fop_##NAME(op, lhs[j], rhs[j]);                                                     \
                mo_set_bit(out_mask, j, 1);                                                                     \
            } else {                                                                                            \
                out[j] = (T)0;                                                                                  \
                mo_set_bit(out_mask, j, 0);                                                                     \
            }                                                                                                   \
        }                                                                                                       \
        return MO_OK;                                                                                           \
    }                                                                                                           \
    /* apply_float_<t> — src/kernels/arithmetic/dispatch.rs:138-206, :389-402 */                                 \
    MO_API int mo_apply_float_##NAME(const T* lhs, size_t lhs_len, const T* rhs, size_t rhs_len, int op,         \
                                     const uint8_t* mask, size_t mask_len, T* out, uint8_t* out_mask,           \
                                     int lanes, int* used_simd) {                                               \
        if (lhs_len != rhs_len) return MO_LENGTH_MISMATCH;                                                      \
        size_t len = lhs_len;                                                                                   \
        int simd = mo_is_simd_aligned(lhs, len) && mo_is_simd_aligned(rhs, len);                                \
        if (used_simd) *used_simd = simd;                                                                       \
        if (mask) {                                                                                             \
            mo_bitmask_new_set_all(out_mask, len, 1);                                                           \
            return simd ? mo_float_masked_simd_##NAME(op, lhs, rhs, mask, mask_len, out, out_mask, len, lanes)  \
                        : mo_float_masked_std_##NAME(op, lhs, rhs, mask, out, out_mask, len);                   \
        }                                                                                                       \
        return mo_float_dense_##NAME(op, lhs, rhs, out, len);                                                   \
    }                                                                                                           \
    /* fma_dense_body_std / _simd — std.rs:223-230, simd.rs:696-751: fused a.mul_add(b, c) */                    \
    MO_API void mo_fma_dense_##NAME(const T* lhs, const T* rhs, const T* acc, T* out, size_t n) {                \
        for (size_t i = 0; i < n; ++i) out[i] = FMA(lhs[i], rhs[i], acc[i]);                                    \
    }                                                                                                           \
    /* fma_masked_body_std — std.rs:196-221; SIMD twin simd.rs:591-694 (all-valid => dense + fill(true)) */      \
    MO_API void mo_fma_masked_##NAME(const T* lhs, const T* rhs, const T* acc, const uint8_t* mask, T* out,      \
                                     uint8_t* out_mask, size_t n) {                                             \
        for (size_t i = 0; i < n; ++i) {                                                                        \
            if (mo_get_bit(mask, i)) {                                                                          \
                out[i] = FMA(lhs[i], rhs[i], acc[i]);                                                           \
                mo_set_bit(out_mask, i, 1);                                                                     \
            } else {                                                                                            \
                out[i] = (T)0;                                                                                  \
                mo_set_bit(out_mask, i, 0);                                                                     \
            }                                                                                                   \
        }                                                                                                       \
    }                                                                                                           \
    /* apply_fma_<t> — dispatch.rs:211-290, :404-418. Aligned inputs take the fused bodies; the unaligned        \
     * fallback inside dispatch is UNFUSED `lhs*rhs + acc` (dispatch.rs:266,280). `force_unfused` selects it. */ \
    MO_API int mo_apply_fma_##NAME(const T* lhs, size_t lhs_len, const T* rhs, size_t rhs_len, const T* acc,     \
                                   size_t acc_len, const uint8_t* mask, T* out, uint8_t* out_mask,              \
                                   int force_unfused) {                                                         \
        if (lhs_len != rhs_len) return MO_LENGTH_MISMATCH;                                                      \
        if (lhs_len != acc_len) return MO_LENGTH_MISMATCH;                                                      \
        size_t len = lhs_len;                                                                                   \
        int simd = !force_unfused && mo_is_simd_aligned(lhs, len) && mo_is_simd_aligned(rhs, len) &&            \
                   mo_is_simd_aligned(acc, len);                                                                \
        if (mask) mo_bitmask_new_set_all(out_mask, len, 1);                                                     \
        if (simd) {                                                                                             \
            if (mask) mo_fma_masked_##NAME(lhs, rhs, acc, mask, out, out_mask, len);                            \
            else mo_fma_dense_##NAME(lhs, rhs, acc, out, len);                                                  \
            return MO_OK;                                                                                       \
        }                                                                                                       \
        for (size_t i = 0; i < len; ++i) {                                                                      \
            if (!mask || mo_get_bit(mask, i)) {                                                                 \
                volatile T prod = lhs[i] * rhs[i]; /* keep the product rounded: no contraction */               \
                out[i] = prod + acc[i];                                                                         \
            } else {                                                                                            \
                out[i] = (T)0;                                                                                  \
                mo_set_bit(out_mask, i, 0);                                                                     \
            }                                                                                                   \
        }                                                                                                       \
        return MO_OK;                                                                                           \
    }

MO_DEFINE_FLOAT(f32, float, fmodf, expf, logf, floorf, fmaf)
MO_DEFINE_FLOAT(f64, double, fmod, exp, log, floor, fma)

/* =================================================================================================
 * Bitmask kernels — src/kernels/bitmask/{mod,std,simd,dispatch}.rs
 * A window is (bits, offset_bits, len_bits) = BitmaskVT (src/aliases.rs:172).
 * Output bitmaps: `out` must hold 8*ceil(len/64) bytes (the reference writes whole words into a
 * Vec64-backed Bitmask and relies on its 64-byte allocation granularity).
 * ============================================================================================== */

enum { MO_AND = 0, MO_OR = 1, MO_XOR = 2 };

/* clear_trailing_bits — src/kernels/bitmask/mod.rs:141-150 (acts on the last of ceil(len/8) bytes) */
static inline void mo_clear_trailing_bits(uint8_t* bits, size_t len) {
    if (len == 0) return;
    size_t used = len & 7;
    if (used) bits[(len + 7) / 8 - 1] &= (uint8_t)((1u << used) - 1);
}

/* bitmask_binop_std / bitmask_binop_simd — std.rs:73-93, simd.rs:95-139.
 * QUIRK restated: the window starts at BYTE offset/8 (bitmask_window_bytes, mod.rs:124-128); a sub-byte
 * offset is not shifted out. The words are read from that byte address (unaligned u64 reads). */
MO_API void mo_bitmask_binop(int op, const uint8_t* lhs, size_t lhs_off, const uint8_t* rhs, size_t rhs_off, size_t len,
                             uint8_t* out) {
    if (len == 0) return;
    size_t nw = (len + 63) / 64;
    const uint8_t* lp = lhs + lhs_off / 8;
    const uint8_t* rp = rhs + rhs_off / 8;
    memset(out, 0, nw * 8); /* Bitmask::new_set_all(len, false) */
    for (size_t k = 0; k < nw; ++k) {
        uint64_t a = mo_word(lp, k), b = mo_word(rp, k), r;
        switch (op) {
            case MO_AND: r = a & b; break;
            case MO_OR: r = a | b; break;
            default: r = a ^ b; break;
        }
        mo_set_word(out, k, r);
    }
    /* Only the last *byte* of the ceil(len/8)-byte bitmap is masked by the reference; bytes between
     * ceil(len/8) and 8*nw are outside its Bitmask (allocation slack). They are zeroed here so results
     * compare as whole words. */
    size_t n_bytes = (len + 7) / 8;
    memset(out + n_bytes, 0, nw * 8 - n_bytes);
    mo_clear_trailing_bits(out, len);
}

/* bitmask_unop_std / _simd (Not) — std.rs:96-114, simd.rs:169-203 */
MO_API void mo_bitmask_not(const uint8_t* src, size_t off, size_t len, uint8_t* out) {
    if (len == 0) return;
    size_t nw = (len + 63) / 64;
    const uint8_t* sp = src + off / 8;
    for (size_t k = 0; k < nw; ++k) mo_set_word(out, k, ~mo_word(sp, k));
    size_t n_bytes = (len + 7) / 8;
    memset(out + n_bytes, 0, nw * 8 - n_bytes);
    mo_clear_trailing_bits(out, len);
}

/* Bitmask::slice_clone(offset, len) as used by in_mask — bit-accurate copy of the window to offset 0. */
static void mo_slice_clone(const uint8_t* src, size_t off, size_t len, uint8_t* out) {
    size_t nw = (len + 63) / 64;
    memset(out, 0, nw * 8);
    for (size_t i = 0; i < len; ++i)
        if (mo_get_bit(src, off + i)) mo_set_bit(out, i, 1);
}

/* in_mask_simd — simd.rs:327-375 (scalar twin std.rs:155-181 scans bit by bit from rhs_off + i).
 * QUIRK restated (SIMD form, the default build): the rhs scan starts at WORD rhs_off/64, not at bit rhs_off. */
MO_API void mo_bitmask_in(const uint8_t* lhs, size_t lhs_off, const uint8_t* rhs, size_t rhs_off, size_t len,
                          uint8_t* out) {
    if (len == 0) return;
    size_t n_words = (len + 63) / 64, trailing = len & 63;
    uint64_t any_set = 0, any_unset = 0;
    for (size_t k = 0; k < n_words; ++k) {
        uint64_t w = mo_word(rhs, rhs_off / 64 + k);
        if (k == n_words - 1 && trailing != 0) {
            uint64_t valid = (((uint64_t)1) << trailing) - 1;
            w &= valid;
            any_set |= w;
            any_unset |= (~w) & valid;
        } else {
            any_set |= w;
            any_unset |= ~w;
        }
        if (any_set && any_unset) break;
    }
    size_t nw = n_words;
    if (any_set && any_unset) {
        memset(out, 0, nw * 8);
        mo_bitmask_new_set_all(out, len, 1);
    } else if (any_set) {
        mo_slice_clone(lhs, lhs_off, len, out);
    } else if (any_unset) {
        mo_bitmask_not(lhs, lhs_off, len, out);
    } else {
        memset(out, 0, nw * 8);
    }
}

/* not_in_mask_simd — simd.rs:392-398 */
MO_API void mo_bitmask_not_in(const uint8_t* lhs, size_t lhs_off, const uint8_t* rhs, size_t rhs_off, size_t len,
                              uint8_t* out, uint8_t* scratch) {
    if (len == 0) return;
    mo_bitmask_in(lhs, lhs_off, rhs, rhs_off, len, scratch);
    mo_bitmask_not(scratch, 0, len, out);
}

/* eq_mask_simd — simd.rs:402-450 (scalar std.rs:191-215). Returns 1 if the reference would panic
 * (offsets not multiples of 64), else 0. */
MO_API int mo_bitmask_eq(const uint8_t* a, size_t ao, const uint8_t* b, size_t bo, size_t len, uint8_t* out) {
    if (len == 0) return 0;
    if (ao % 64 != 0 || bo % 64 != 0) return 1;
    size_t n_words = (len + 63) / 64;
    for (size_t k = 0; k < n_words; ++k) mo_set_word(out, k, ~(mo_word(a, ao / 64 + k) ^ mo_word(b, bo / 64 + k)));
    size_t n_bytes = (len + 7) / 8;
    memset(out + n_bytes, 0, n_words * 8 - n_bytes);
    mo_mask_trailing_bits(out, len);
    return 0;
}

/* ne_mask_simd = !eq_mask_simd — simd.rs:468-472 (Bitmask `!` inverts and re-masks the tail) */
MO_API int mo_bitmask_ne(const uint8_t* a, size_t ao, const uint8_t* b, size_t bo, size_t len, uint8_t* out) {
    if (len == 0) return 0;
    if (ao % 64 != 0 || bo % 64 != 0) return 1;
    size_t n_words = (len + 63) / 64;
    for (size_t k = 0; k < n_words; ++k) mo_set_word(out, k, mo_word(a, ao / 64 + k) ^ mo_word(b, bo / 64 + k));
    size_t n_bytes = (len + 7) / 8;
    memset(out + n_bytes, 0, n_words * 8 - n_bytes);
    mo_mask_trailing_bits(out, len);
    return 0;
}

/* all_eq_mask_simd — simd.rs:511-581. Returns 0/1, or -1 where the reference panics (len >= 64 with an
 * offset that is not a multiple of 64). */
MO_API int mo_bitmask_all_eq(const uint8_t* a, size_t ao, const uint8_t* b, size_t bo, size_t len) {
    if (len == 0) return 1;
    if (len < 64) {
        uint64_t wa = mo_word(a, ao / 64), wb = mo_word(b, bo / 64), valid = (((uint64_t)1) << len) - 1;
        return (wa & valid) == (wb & valid);
    }
    if (ao % 64 != 0 || bo % 64 != 0) return -1;
    size_t n_words = (len + 63) / 64, trailing = len & 63;
    for (size_t k = 0; k < n_words; ++k) {
        uint64_t wa = mo_word(a, ao / 64 + k), wb = mo_word(b, bo / 64 + k);
        if (k == n_words - 1 && trailing != 0) {
            uint64_t m = (((uint64_t)1) << trailing) - 1;
            if ((wa & m) != (wb & m)) return 0;
        } else if (wa != wb) {
            return 0;
        }
    }
    return 1;
}

/* all_ne_mask_simd = !all_eq_mask_simd — simd.rs:490-494 ("not all equal", NOT "all different") */
MO_API int mo_bitmask_all_ne(const uint8_t* a, size_t ao, const uint8_t* b, size_t bo, size_t len) {
    int r = mo_bitmask_all_eq(a, ao, b, bo, len);
    return r < 0 ? r : !r;
}

/* popcount_mask / popcount_mask_simd — std.rs:279-296, simd.rs:596-644.
 * QUIRK restated: counting starts at WORD offset/64 (bit offset inside the word is ignored). */
MO_API size_t mo_bitmask_popcount(const uint8_t* bits, size_t off, size_t len) {
    if (len == 0) return 0;
    size_t n_words = (len + 63) / 64, word_start = off / 64, acc = 0;
    for (size_t k = 0; k < n_words; ++k) {
        uint64_t w = mo_word(bits, word_start + k);
        if (k == n_words - 1 && len % 64 != 0) w &= (((uint64_t)1) << (len % 64)) - 1;
        acc += (size_t)__builtin_popcountll(w);
    }
    return acc;
}

/* merge_bitmasks_to_new — src/kernels/bitmask/mod.rs:171-196: per-row AND of optional masks (bit 0 based).
 * Returns 0 when both are NULL (=> None), else 1 and fills out (8*ceil(len/64) bytes). */
MO_API int mo_merge_bitmasks(const uint8_t* l, const uint8_t* r, size_t len, uint8_t* out) {
    if (!l && !r) return 0;
    size_t nw = (len + 63) / 64;
    memset(out, 0, nw * 8);
    for (size_t i = 0; i < len; ++i) {
        int v = (l ? mo_get_bit(l, i) : 1) && (r ? mo_get_bit(r, i) : 1);
        if (v) mo_set_bit(out, i, 1);
    }
    return 1;
}

/* Bitmask::union — src/structs/bitmask.rs:661 (bitwise OR), used by route_super_array_broadcast
 * (src/kernels/broadcast/super_array.rs:224) */
MO_API void mo_bitmask_union(const uint8_t* l, const uint8_t* r, size_t len, uint8_t* out) {
    size_t nw = (len + 63) / 64;
    memset(out, 0, nw * 8);
    for (size_t i = 0; i < len; ++i)
        if (mo_get_bit(l, i) || mo_get_bit(r, i)) mo_set_bit(out, i, 1);
}

/* simd_eq_mask_u{8,16,32,64}(data, field_mask, target) — simd.rs:741-788: bit j = ((data[j] & field_mask) == target) */
#define MO_DEFINE_EQ_MASK(NAME, T)                                                                              \
    MO_API void mo_simd_eq_mask_##NAME(const T* data, size_t n, T field_mask, T target, uint8_t* out) {          \
        size_t nw = (n + 63) / 64;                                                                              \
        memset(out, 0, nw * 8);                                                                                 \
        for (size_t j = 0; j < n; ++j)                                                                          \
            if ((T)(data[j] & field_mask) == target) out[j / 8] |= (uint8_t)(1u << (j % 8));                    \
    }
MO_DEFINE_EQ_MASK(u8, uint8_t)
MO_DEFINE_EQ_MASK(u16, uint16_t)
MO_DEFINE_EQ_MASK(u32, uint32_t)
MO_DEFINE_EQ_MASK(u64, uint64_t)

/* =================================================================================================
 * Scalar broadcast — src/kernels/routing/broadcast.rs:25-112: a length-1 side is MATERIALISED as
 * vec64![x; n] and the ordinary two-array kernel runs. (The GPU path fuses this; results are equal.)
 * ============================================================================================== */
MO_API void mo_broadcast_len1(const void* one, size_t elem_size, size_t n, void* out) {
    for (size_t i = 0; i < n; ++i) memcpy((uint8_t*)out + i * elem_size, one, elem_size);
}

/* =================================================================================================
 * Consolidation of a chunked numeric column — src/traits/consolidate.rs:80-207 (consolidate_{int,float}_variant!,
 * extend_null_mask), reached from Consolidate::consolidate (src/structs/chunked/super_table.rs:657-722).
 * Values: extend_from_slice per chunk. Validity: the result has a mask iff any chunk has one (`has_nulls`,
 * consolidate.rs:118-124); chunks with a mask extend it bit by bit from [offset, offset+len), chunks without
 * set their rows valid (consolidate.rs:86-103). Returns has_nulls.
 * ============================================================================================== */
MO_API int mo_consolidate_column(size_t elem_size, size_t n_chunks, const void* const* chunk_data, const size_t* chunk_lens,
                                 const uint8_t* const* chunk_masks, const size_t* chunk_mask_offsets, void* out_data,
                                 uint8_t* out_mask) {
    int has_nulls = 0;
    size_t total = 0;
    for (size_t i = 0; i < n_chunks; ++i) {
        if (chunk_masks && chunk_masks[i]) has_nulls = 1;
        total += chunk_lens[i];
    }
    if (has_nulls) memset(out_mask, 0, ((total + 63) / 64) * 8);
    size_t cur = 0;
    for (size_t i = 0; i < n_chunks; ++i) {
        memcpy((uint8_t*)out_data + cur * elem_size, chunk_data[i], chunk_lens[i] * elem_size);
        if (has_nulls) {
            for (size_t k = 0; k < chunk_lens[i]; ++k) {
                int v = (chunk_masks && chunk_masks[i])
                            ? mo_get_bit(chunk_masks[i], (chunk_mask_offsets ? chunk_mask_offsets[i] : 0) + k)
                            : 1;
                if (v) mo_set_bit(out_mask, cur + k, 1);
            }
        }
        cur += chunk_lens[i];
    }
    return has_nulls;
}

/* =================================================================================================
 * Bit-packed column consolidation — Bitmask::extend_from_bitmask_range (src/structs/bitmask.rs:520-545) on top of
 * Bitmask::extend_from_slice (:547-586), driven per chunk as BooleanArray::append_range does
 * (src/structs/variants/boolean.rs:627-653) / Arena::write_boolean_slices (src/structs/arena.rs:391-430).
 *  - source window not byte aligned: the bytes are first shifted down by (offset & 7) into a scratch run (:530-544);
 *  - destination length byte aligned: whole source bytes are stored, then the tail bits are merged (:557-571);
 *  - otherwise: one bit at a time (:573-584).
 * `dst` must have room for ceil((dst_len + len) / 8) + 1 bytes; returns the new length in bits.
 * ============================================================================================== */
static size_t mo_extend_from_slice(uint8_t* dst, size_t dst_len, const uint8_t* src, size_t len) {
    if ((dst_len & 7) == 0) {
        size_t at = dst_len >> 3, full = len >> 3, tail = len & 7;
        for (size_t i = 0; i < full; ++i) dst[at + i] = src[i];
        if (tail) {
            uint8_t keep = (uint8_t)((1u << tail) - 1u);
            dst[at + full] = (uint8_t)((dst[at + full] & (uint8_t)~keep) | (src[full] & keep));
        }
    } else {
        for (size_t i = 0; i < len; ++i) mo_set_bit(dst, dst_len + i, mo_get_bit(src, i));
    }
    /* mask_trailing_bits (bitmask.rs:83-90): bits past the new length in the last byte are zero */
    size_t new_len = dst_len + len;
    if (new_len & 7) dst[new_len >> 3] &= (uint8_t)((1u << (new_len & 7)) - 1u);
    return new_len;
}

MO_API size_t mo_bitmask_extend_from_range(uint8_t* dst, size_t dst_len, const uint8_t* src, size_t src_bytes,
                                           size_t offset, size_t len) {
    if (len == 0) return dst_len;
    if ((offset & 7) == 0) return mo_extend_from_slice(dst, dst_len, src + (offset >> 3), len);
    size_t first = offset >> 3, shift = offset & 7;
    size_t want = ((len + 7) >> 3) + 1;
    size_t end = first + want < src_bytes ? first + want : src_bytes;
    uint8_t* run = (uint8_t*)calloc(want + 1, 1);
    for (size_t i = first; i < end; ++i) {
        uint8_t lo = (uint8_t)(src[i] >> shift);
        uint8_t hi = (i + 1 < src_bytes) ? (uint8_t)(src[i + 1] << (8 - shift)) : 0;
        run[i - first] = (uint8_t)(lo | hi);
    }
    size_t r = mo_extend_from_slice(dst, dst_len, run, len);
    free(run);
    return r;
}

/* has_nulls rule as mo_consolidate_column. out_bits / out_mask: 8*ceil(total/64) + 8 zeroed bytes. */
MO_API int mo_consolidate_boolean_column(size_t n_chunks, const uint8_t* const* chunk_bits, const size_t* chunk_bytes,
                                         const size_t* chunk_bit_offsets, const size_t* chunk_lens,
                                         const uint8_t* const* chunk_masks, const size_t* chunk_mask_bytes,
                                         const size_t* chunk_mask_offsets, uint8_t* out_bits, uint8_t* out_mask) {
    int has_nulls = 0;
    for (size_t i = 0; i < n_chunks; ++i)
        if (chunk_masks && chunk_masks[i]) has_nulls = 1;
    size_t len = 0, mlen = 0;
    for (size_t i = 0; i < n_chunks; ++i) {
        len = mo_bitmask_extend_from_range(out_bits, len, chunk_bits[i], chunk_bytes[i],
                                           chunk_bit_offsets ? chunk_bit_offsets[i] : 0, chunk_lens[i]);
        if (!has_nulls) continue;
        if (chunk_masks[i]) {
            mlen = mo_bitmask_extend_from_range(out_mask, mlen, chunk_masks[i], chunk_mask_bytes[i],
                                                chunk_mask_offsets ? chunk_mask_offsets[i] : 0, chunk_lens[i]);
        } else { /* mask.resize(len + n, true) — boolean.rs:642-644, arena.rs:419 */
            for (size_t k = 0; k < chunk_lens[i]; ++k) mo_set_bit(out_mask, mlen + k, 1);
            mlen += chunk_lens[i];
        }
    }
    return has_nulls;
}
