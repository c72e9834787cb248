/* The reference's hot loop — N passes of `sum` over the same IntegerArray<i64> and FloatArray<f64>
 * (benches/hotloop_benchmark_avg_std.rs:48-62: ITERATIONS passes, an i64 and an f64 sum each; the pass itself: hotloop_benchmark_std.rs:109-127, benches/hotloop_benchmark_simd.rs: one call per pass, results kept) — on one GPU
 * from a C99 host through the C ABI alone, two ways:
 *   1. every pass one ma_sum_fused launch on the context's stream (asynchronous context, one record per pass);
 *   2. the same launches through ma_scan_lanes_sum_fused: consecutive passes on two streams of the GPU, each started when the pass
 *      in front of it has begun to drain — no launch ramp and no spread of finish times between the passes;
 *   3. every pass two single-column scans (the kernels of ma_i64_sum and ma_f64_sum_dd) through ma_scan_lanes_sum — the form that
 *      takes any numeric type.
 * Every pass's record is checked against the closed forms of the bench's own input (v[i] = i).
 * Build:  gcc -std=c99 -Iinclude examples/hot_loop_sums.c -Lminarrow_amd/lib -lminarrow_hip -Wl,-rpath,$PWD/minarrow_amd/lib
 * Run:    ./a.out [rows per column = 2^24] [passes = 64]
 * Exit code 0 and a line starting with "ok" when every pass matched; 2 when no GPU is visible. */
#define _POSIX_C_SOURCE 199309L
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "minarrow_hip.h"

#define CHECK(call)                                                                                  \
    do {                                                                                             \
        ma_status st_ = (call);                                                                      \
        if (st_ != MA_OK) {                                                                          \
            fprintf(stderr, "%s: %s: %s\n", #call, ma_status_name(st_), ma_last_error_string());     \
            return 1;                                                                                \
        }                                                                                            \
    } while (0)

static double now_ms(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec * 1e3 + (double)t.tv_nsec * 1e-6;
}

/* record k of `records`: {i64 sum, count, f64 hi, f64 lo, count, -, -, -} */
static int records_match(ma_ctx* ctx, const void* records, int passes, size_t rows) {
    const size_t bytes = (size_t)passes * 64;
    uint64_t* w = (uint64_t*)malloc(bytes);
    if (!w || ma_dev_download(ctx, w, records, bytes) != MA_OK) return 0;
    const uint64_t tri = (uint64_t)rows * (uint64_t)(rows - 1) / 2;
    int ok = 1;
    for (int k = 0; k < passes && ok; ++k) {
        double hi, lo;
        memcpy(&hi, &w[8 * k + 2], 8);
        memcpy(&lo, &w[8 * k + 3], 8);
        ok = w[8 * k] == tri && w[8 * k + 1] == rows && w[8 * k + 4] == rows && hi + lo == (double)tri; /* < 2^53: exact */
    }
    free(w);
    return ok;
}

int main(int argc, char** argv) {
    const size_t rows = argc > 1 ? (size_t)strtoull(argv[1], NULL, 10) : ((size_t)1 << 24);
    const int passes = argc > 2 ? atoi(argv[2]) : 64;
    if (ma_device_count() <= 0) {
        printf("no HIP device is visible\n");
        return 2;
    }
    if (rows < 2 || rows > ((size_t)1 << 26) || passes < 2) { /* 2^26: the f64 closed form stays exact */
        fprintf(stderr, "rows in [2, 2^26], passes >= 2\n");
        return 1;
    }
    ma_ctx* ctx = NULL;
    CHECK(ma_ctx_create(0, &ctx));
    void *ints = NULL, *floats = NULL, *records = NULL;
    CHECK(ma_dev_alloc(ctx, rows * 8 + 64, &ints));
    CHECK(ma_dev_alloc(ctx, rows * 8 + 64, &floats));
    CHECK(ma_dev_alloc(ctx, (size_t)passes * 64, &records));
    CHECK(ma_synth_iota_i64(ctx, (int64_t*)ints, rows, 0));
    CHECK(ma_synth_iota_f64(ctx, (double*)floats, rows, 0));
    CHECK(ma_ctx_set_async(ctx, 1)); /* enqueue-only calls: the host runs ahead of the GPU, as a stepping host does */

    ma_fused_column cols[2];
    memset(cols, 0, sizeof(cols));
    cols[0].data = ints, cols[0].n = rows, cols[0].null_count = -1, cols[0].format_code = 'l';
    cols[1].data = floats, cols[1].n = rows, cols[1].null_count = -1, cols[1].format_code = 'g';

    double ms[3] = {0, 0, 0};
    ma_scan_lanes* lanes = NULL;
    CHECK(ma_scan_lanes_create(ctx, &lanes));
    for (int form = 0; form < 3; ++form) {
        for (int round = 0; round < 2; ++round) { /* the first round warms the clocks up */
            CHECK(ma_dev_memset(ctx, records, 0, (size_t)passes * 64));
            CHECK(ma_ctx_synchronize(ctx));
            const double t0 = now_ms();
            for (int k = 0; k < passes; ++k) {
                cols[0].out = (uint64_t*)records + 8 * k;
                cols[1].out = (uint64_t*)records + 8 * k + 2;
                uint64_t* rec = (uint64_t*)records + 8 * k;
                if (form == 0) {
                    CHECK(ma_sum_fused(ctx, 2, cols));
                } else if (form == 1) {
                    CHECK(ma_scan_lanes_sum_fused(lanes, 2, cols));
                } else {
                    CHECK(ma_scan_lanes_sum(lanes, 'l', ints, rows, NULL, 0, 0, rec, NULL, rec + 1));
                    CHECK(ma_scan_lanes_sum(lanes, 'g', floats, rows, NULL, 0, 0, rec + 2, (double*)(rec + 3), rec + 4));
                }
            }
            if (form == 0) CHECK(ma_ctx_synchronize(ctx));
            else CHECK(ma_scan_lanes_synchronize_for(lanes, 20000.0));  /* the bounded form: a gate nobody opens is an error, not a hung host */
            ms[form] = (now_ms() - t0) / passes;
            if (!records_match(ctx, records, passes, rows)) {
                fprintf(stderr, "form %d: a pass's record does not match the closed forms\n", form);
                return 1;
            }
        }
    }
    ma_scan_lanes_destroy(lanes);
    const double gb = (double)rows * 16e-9;
    printf("ok: %d passes over 2 x %zu rows; one stream %.2f us per pass (%.2f TB/s), two scan lanes %.2f us (%.2f TB/s), "
           "two scan lanes with one column per scan %.2f us (%.2f TB/s)\n",
           passes, rows, ms[0] * 1e3, gb / ms[0], ms[1] * 1e3, gb / ms[1], ms[2] * 1e3, gb / ms[2]);
    CHECK(ma_ctx_set_async(ctx, 0));
    CHECK(ma_dev_free(ctx, ints));
    CHECK(ma_dev_free(ctx, floats));
    CHECK(ma_dev_free(ctx, records));
    ma_ctx_destroy(ctx);
    return 0;
}
