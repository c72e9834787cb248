/* Does the read rate of a block depend on WHICH physical memory the driver hands out? One torch-free process:
 *  A: allocate 8 GB, fill, time ma_i64_sum, really free it (pool trimmed to 0) — eight times over (recycled memory);
 *  B: six 8-GB blocks alive at once, each timed (fresh regions next to recycled ones);
 *  C: everything freed, one more block.
 * Round 4 (the 7.3-vs-6.9 TB/s states of tools/probe_proc.c that begin at process boundaries).
 * Build: gcc -std=gnu99 -O2 -Iinclude tools/probe_churn.c -Lminarrow_amd/lib -lminarrow_hip -Wl,-rpath,$PWD/minarrow_amd/lib */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>

#include "minarrow_hip.h"

#define CHECK(call)                                                                              \
    do {                                                                                         \
        ma_status st_ = (call);                                                                  \
        if (st_ != MA_OK) {                                                                      \
            fprintf(stderr, "%s: %s: %s\n", #call, ma_status_name(st_), ma_last_error_string()); \
            return 1;                                                                            \
        }                                                                                        \
    } while (0)

static double rate(ma_ctx* ctx, const void* col, size_t n, void* slot) {
    float best = 1e9f;
    for (int w = 0; w < 2; ++w) ma_i64_sum(ctx, (const int64_t*)col, n, NULL, 0, 0, (int64_t*)slot, (uint64_t*)slot + 1);
    for (int trial = 0; trial < 3; ++trial) {
        float ms = 0;
        ma_ctx_synchronize(ctx);
        ma_ctx_timer_start(ctx);
        for (int r = 0; r < 10; ++r) ma_i64_sum(ctx, (const int64_t*)col, n, NULL, 0, 0, (int64_t*)slot, (uint64_t*)slot + 1);
        ma_ctx_timer_stop(ctx);
        ma_ctx_timer_elapsed_ms(ctx, &ms);
        if (ms / 10 < best) best = ms / 10;
    }
    return 8.0 * (double)n / best / 1e9;
}

int main(int argc, char** argv) {
    const size_t n = argc > 1 ? (size_t)strtoull(argv[1], NULL, 10) : (size_t)1000000000;
    const char* tag = getenv("PROBE_TAG") ? getenv("PROBE_TAG") : "";
    if (ma_device_count() <= 0) return printf("{\"error\": \"no device\"}\n"), 2;
    ma_ctx* ctx = NULL;
    void* slot = NULL;
    CHECK(ma_ctx_create(0, &ctx));
    CHECK(ma_ctx_set_async(ctx, 1));
    CHECK(ma_dev_alloc(ctx, 256, &slot));
    printf("{\"tag\": \"%s\", \"A_recycled\": [", tag);
    for (int i = 0; i < 8; ++i) {
        void* b = NULL;
        CHECK(ma_dev_alloc(ctx, n * 8, &b));
        CHECK(ma_synth_iota_i64(ctx, (int64_t*)b, n, i));
        printf("%s[\"%p\", %.3f]", i ? ", " : "", b, rate(ctx, b, n, slot));
        fflush(stdout);
        CHECK(ma_ctx_synchronize(ctx));
        CHECK(ma_dev_free(ctx, b));
        CHECK(ma_dev_pool_trim(ctx, 0));
    }
    printf("], \"B_six_alive\": [");
    void* blk[6] = {0};
    for (int i = 0; i < 6; ++i) {
        CHECK(ma_dev_alloc(ctx, n * 8, &blk[i]));
        CHECK(ma_synth_iota_i64(ctx, (int64_t*)blk[i], n, i));
    }
    for (int round = 0; round < 2; ++round)
        for (int i = 0; i < 6; ++i) printf("%s[\"%p\", %.3f]", (i || round) ? ", " : "", blk[i], rate(ctx, blk[i], n, slot));
    CHECK(ma_ctx_synchronize(ctx));
    for (int i = 0; i < 6; ++i) CHECK(ma_dev_free(ctx, blk[i]));
    CHECK(ma_dev_pool_trim(ctx, 0));
    void* b = NULL;
    CHECK(ma_dev_alloc(ctx, n * 8, &b));
    CHECK(ma_synth_iota_i64(ctx, (int64_t*)b, n, 0));
    printf("], \"C_after\": [\"%p\", %.3f]}\n", b, rate(ctx, b, n, slot));
    CHECK(ma_ctx_synchronize(ctx));
    ma_dev_free(ctx, b);
    ma_dev_free(ctx, slot);
    ma_ctx_destroy(ctx);
    return 0;
}
