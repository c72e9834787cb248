/* Is the 6.9 TB/s state the kernel driver wiping freed VRAM in the background? One torch-free process keeps summing ONE
 * resident 8-GB column (10 launches per sample, ~11 ms) while, at marked moments, it really frees a ballast block of
 * `ballast_gb` (ma_dev_free + ma_dev_pool_trim(0) = hipFree): KFD clears released VRAM asynchronously, so a dip in the
 * read rate right after each free that lasts in proportion to the freed bytes — with nothing else changing — is that wipe.
 * (Back-to-back short processes see the same thing from their predecessor's exit.) Round 4.
 * Usage: probe_wipe [ballast_gb ...]   e.g. probe_wipe 16 64 0 32
 * Build: gcc -std=gnu99 -O2 -Iinclude tools/probe_wipe.c -Lminarrow_amd/lib -lminarrow_hip -Wl,-rpath,$PWD/minarrow_amd/lib */
#define _GNU_SOURCE
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "minarrow_hip.h"

#define CHECK(call)                                                                              \
    do {                                                                                         \
        ma_status st_ = (call);                                                                  \
        if (st_ != MA_OK) {                                                                      \
            fprintf(stderr, "%s: %s: %s\n", #call, ma_status_name(st_), ma_last_error_string()); \
            return 1;                                                                            \
        }                                                                                        \
    } while (0)

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static ma_ctx* ctx;
static void *col, *slot;
static const size_t n = 1000000000;

static double sample(void) {
    float ms = 0;
    ma_ctx_timer_start(ctx);
    for (int r = 0; r < 10; ++r) ma_i64_sum(ctx, (const int64_t*)col, n, NULL, 0, 0, (int64_t*)slot, (uint64_t*)slot + 1);
    ma_ctx_timer_stop(ctx);
    ma_ctx_timer_elapsed_ms(ctx, &ms);
    return 8.0 * (double)n * 10 / ms / 1e9;
}

int main(int argc, char** argv) {
    const char* tag = getenv("PROBE_TAG") ? getenv("PROBE_TAG") : "";
    if (ma_device_count() <= 0) return printf("{\"error\": \"no device\"}\n"), 2;
    CHECK(ma_ctx_create(0, &ctx));
    CHECK(ma_ctx_set_async(ctx, 1));
    CHECK(ma_dev_alloc(ctx, 256, &slot));
    CHECK(ma_dev_alloc(ctx, n * 8, &col));
    CHECK(ma_synth_iota_i64(ctx, (int64_t*)col, n, 0));
    CHECK(ma_ctx_synchronize(ctx));
    for (int i = 0; i < 60; ++i) sample(); /* clocks up, whatever the previous process left behind is over (0.7 s) */
    const double t0 = now_s();
    for (int a = 1; a < argc; ++a) {
        const double gb = atof(argv[a]);
        void* ballast = NULL;
        if (gb > 0) {
            CHECK(ma_dev_alloc(ctx, (size_t)(gb * 1e9), &ballast));
            CHECK(ma_dev_memset(ctx, ballast, 1, (size_t)(gb * 1e9))); /* touched: really backed */
            CHECK(ma_ctx_synchronize(ctx));
        }
        double before = 0;
        for (int i = 0; i < 30; ++i) before += sample() / 30;
        const double t_free0 = now_s();
        if (ballast) {
            CHECK(ma_dev_free(ctx, ballast));
            CHECK(ma_dev_pool_trim(ctx, 0));
        }
        const double t_free1 = now_s();
        printf("{\"tag\": \"%s\", \"ballast_gb\": %.0f, \"before_tbps\": %.3f, \"free_call_ms\": %.1f, \"t_free\": %.3f, \"after\": [", tag, gb, before,
               (t_free1 - t_free0) * 1e3, t_free1 - t0);
        double lo = 1e9;
        int slow = 0, total = 0;
        const double t_end = now_s() + 2.5;
        while (now_s() < t_end) {
            const double r = sample();
            if (r < lo) lo = r;
            if (r < before - 0.15) ++slow;
            if (total < 230) printf("%s[%.0f, %.2f]", total ? ", " : "", (now_s() - t_free1) * 1e3, r);
            ++total;
        }
        printf("], \"min_after_tbps\": %.3f, \"samples_0.15_below\": %d, \"samples\": %d}\n", lo, slow, total);
        fflush(stdout);
    }
    ma_dev_free(ctx, col);
    ma_dev_free(ctx, slot);
    ma_ctx_destroy(ctx);
    return 0;
}
