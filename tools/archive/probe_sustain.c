/* Read rate of ma_i64_sum over time in ONE torch-free process, next to the GPU's own clock / power / temperature sensors:
 * phases of back-to-back launches separated by idle gaps. Round 4: the "per-process" 7.3-vs-6.9 TB/s bimodality of
 * profiles/r03_read_rate_by_allocation.txt moves INSIDE a process (tools/probe_proc.c), so what does it follow?
 * Usage: probe_sustain [rows] [variant] [blocks_per_cu] [phases: "busy_s,idle_s,busy_s,..."]
 * Output: one JSON line per sample (10 launches each) + a summary line per phase.
 * Build: gcc -std=gnu99 -O2 -Iinclude tools/probe_sustain.c -Lminarrow_amd/lib -lminarrow_hip -Wl,-rpath,$PWD/minarrow_amd/lib */
#define _GNU_SOURCE
#include <dirent.h>
#include <fcntl.h>
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

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

#include "probe_sensors.h"

int main(int argc, char** argv) {
    const size_t n = argc > 1 ? (size_t)strtoull(argv[1], NULL, 10) : (size_t)1000000000;
    const int variant = argc > 2 ? atoi(argv[2]) : 0;
    const int bpc = argc > 3 ? atoi(argv[3]) : 0;
    const char* phases = argc > 4 ? argv[4] : "8,3,4,10,4";
    const char* tag = getenv("PROBE_TAG") ? getenv("PROBE_TAG") : "";
    if (ma_device_count() <= 0) return printf("{\"error\": \"no device\"}\n"), 2;
    find_hwmon();
    ma_ctx* ctx = NULL;
    void *blk = NULL, *slot = NULL;
    CHECK(ma_ctx_create(0, &ctx));
    CHECK(ma_ctx_set_async(ctx, 1));
    CHECK(ma_ctx_set_variant(ctx, variant));
    CHECK(ma_ctx_set_blocks_per_cu(ctx, bpc));
    CHECK(ma_dev_alloc(ctx, 256, &slot));
    CHECK(ma_dev_alloc(ctx, n * 8, &blk));
    CHECK(ma_synth_iota_i64(ctx, (int64_t*)blk, n, 0));
    CHECK(ma_ctx_synchronize(ctx));
    {
        struct timespec rt;
        clock_gettime(CLOCK_REALTIME, &rt);
        printf("{\"epoch_at_t0\": %.3f}\n", (double)rt.tv_sec + 1e-9 * (double)rt.tv_nsec);
    }
    printf("{\"tag\": \"%s\", \"rows\": %zu, \"variant\": %d, \"blocks_per_cu\": %d, \"hwmon\": %d, \"phases\": \"%s\"", tag, n, variant, bpc,
           g_n_hwmon, phases);
    print_sensors();
    printf("}\n");
    const double t_origin = now_s();
    char* spec = strdup(phases);
    int phase = 0;
    for (char* tok = strtok(spec, ","); tok; tok = strtok(NULL, ","), ++phase) {
        const double dur = atof(tok);
        if (phase & 1) {  /* idle gap */
            usleep((useconds_t)(dur * 1e6));
            printf("{\"tag\": \"%s\", \"phase\": %d, \"idle_s\": %.1f, \"t\": %.3f", tag, phase, dur, now_s() - t_origin);
            print_sensors();
            printf("}\n");
            continue;
        }
        const double t_end = now_s() + dur;
        double sum = 0, lo = 1e9, hi = 0, first = 0;
        int samples = 0;
        while (now_s() < t_end) {
            float ms = 0;
            CHECK(ma_ctx_timer_start(ctx));
            for (int r = 0; r < 10; ++r) ma_i64_sum(ctx, (const int64_t*)blk, n, NULL, 0, 0, (int64_t*)slot, (uint64_t*)slot + 1);
            CHECK(ma_ctx_timer_stop(ctx));
            /* keep the queue fed: the next batch is enqueued by the next loop trip right after this wait returns */
            CHECK(ma_ctx_timer_elapsed_ms(ctx, &ms));
            const double tbps = 8.0 * (double)n * 10 / ms / 1e9;
            if (!samples) first = tbps;
            sum += tbps;
            if (tbps < lo) lo = tbps;
            if (tbps > hi) hi = tbps;
            if (samples % 8 == 0) {
                printf("{\"tag\": \"%s\", \"phase\": %d, \"t\": %.3f, \"tbps\": %.3f", tag, phase, now_s() - t_origin, tbps);
                print_sensors();
                printf("}\n");
            }
            ++samples;
        }
        printf("{\"tag\": \"%s\", \"phase\": %d, \"busy_s\": %.1f, \"samples\": %d, \"first_tbps\": %.3f, \"mean_tbps\": %.3f, \"min_tbps\": %.3f, "
               "\"max_tbps\": %.3f}\n", tag, phase, dur, samples, first, sum / samples, lo, hi);
    }
    free(spec);
    ma_dev_free(ctx, blk);
    ma_dev_free(ctx, slot);
    ma_ctx_destroy(ctx);
    return 0;
}
