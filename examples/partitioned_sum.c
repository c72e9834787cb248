/* The reference's parallel sums — `slice.par_chunks(..).map(simd_sum).sum()` for an i64 and an f64 column
 * (benches/benchmark_parallel_simd.rs:81-125) — partitioned over every visible GPU from ONE C99 host through the C ABI alone,
 * written the way a host that must not hang writes it:
 *   1. a group over the GPUs with the best exchange (RCCL all-gather + device fold, overlapped with the next step's scans,
 *      consecutive steps on two scan lanes gated on each other's early stamp);
 *   2. ma_group_selftest before the machinery is trusted with a job (rank-tagged records through the exchange, every peer link,
 *      the stamp hand-off — each step under a deadline);
 *   3. a stepping loop whose waits are bounded (ma_group_synchronize_for); when one runs out — injected here with the library's
 *      own fault hook on the third step — the group is rebuilt ONE NOTCH DOWN (same members, same columns) and the job goes on:
 *      two scan lanes -> one -> in-stream -> the calling thread issuing grouped collectives -> the host fold;
 *   4. every step's totals against their closed forms.
 * Build:  gcc -std=c99 -Iinclude examples/partitioned_sum.c -Lminarrow_amd/lib -lminarrow_hip -Wl,-rpath,$PWD/minarrow_amd/lib
 * Run:    ./a.out [rows per column = 2^26] [steps = 8] [deadline in ms = 500]
 * Exit code 0 and a line starting with "ok" when every step matched; 2 when no GPU is visible. */
#include <inttypes.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "minarrow_hip.h"
#include "minarrow_hip_testing.h" /* the fault hook below: live only when MINARROW_HIP_TEST_HOOKS=1 is in the environment */

#define CHECK(call)                                                                                  \
    do {                                                                                             \
        ma_status st_ = (call);                                                                      \
        if (st_ != MA_OK) {                                                                          \
            fprintf(stderr, "%s: %s: %s\n", #call, ma_status_name(st_), ma_last_error_string());     \
            return 1;                                                                                \
        }                                                                                            \
    } while (0)

#define MAX_GPUS 8

/* one notch down from the flags in effect */
static uint32_t next_notch(uint32_t flags) {
    if (flags & MA_GROUP_SCAN_LANES) return flags & ~(uint32_t)MA_GROUP_SCAN_LANES;                            /* one scan stream */
    if (flags & MA_GROUP_EXCHANGE_OVERLAP) return flags & ~(uint32_t)MA_GROUP_EXCHANGE_OVERLAP;              /* in-stream */
    if ((flags & MA_GROUP_EXCHANGE_RCCL) && !(flags & MA_GROUP_ISSUE_CALLER)) return flags | MA_GROUP_ISSUE_CALLER; /* grouped calls */
    return 0;                                                                                                  /* host fold */
}

static const char* form_name(uint32_t flags) {
    if (!(flags & MA_GROUP_EXCHANGE_RCCL)) return "host fold of pinned records";
    if (flags & MA_GROUP_SCAN_LANES) return "RCCL, overlapped on side streams, consecutive steps on two scan lanes";
    if (flags & MA_GROUP_EXCHANGE_OVERLAP) return "RCCL, overlapped on side streams";
    return (flags & MA_GROUP_ISSUE_CALLER) ? "RCCL, in-stream, grouped on the calling thread" : "RCCL, in-stream, one issue thread per GPU";
}

int main(int argc, char** argv) {
    const size_t rows = argc > 1 ? (size_t)strtoull(argv[1], NULL, 10) : ((size_t)1 << 26);
    const int steps = argc > 2 ? atoi(argv[2]) : 8;
    const double deadline_ms = argc > 3 ? atof(argv[3]) : 500.0;
    const int visible = ma_device_count();
    int n = visible;
    if (n <= 0) {
        printf("no HIP device is visible\n");
        return 2;
    }
    if (n > MAX_GPUS) n = MAX_GPUS;
    /* argv[4] = members: more members than GPUs makes them share devices (member i on GPU i mod visible) — what only the loopback
     * collective double takes (MINARROW_HIP_RCCL_PATH, a rehearsal on a one-GPU box); with RCCL itself the group falls back to the host fold */
    if (argc > 4 && atoi(argv[4]) > 0 && atoi(argv[4]) <= MAX_GPUS) n = atoi(argv[4]);
    int32_t devices[MAX_GPUS];
    for (int i = 0; i < n; ++i) devices[i] = i % visible;

    ma_group* g = NULL;
    CHECK(ma_group_create_ex(devices, n,
                             MA_GROUP_EXCHANGE_RCCL | MA_GROUP_EXCHANGE_OVERLAP | MA_GROUP_SCAN_LANES | MA_GROUP_EXCHANGE_FALLBACK_HOST, &g));
    printf("group of %d GPU(s): %s [%s]\n", n, form_name(ma_group_flags(g)), ma_group_exchange_note(g));

    /* 64-row-aligned row chunks (a chunk's validity window then starts on a word); chunk i lives on GPU i */
    const void* col_i[MAX_GPUS];
    const void* col_f[MAX_GPUS];
    size_t len[MAX_GPUS];
    const size_t units = (rows + 63) / 64;
    size_t lo = 0;
    for (int i = 0; i < n; ++i) {
        size_t hi = i + 1 == n ? rows : (units * (size_t)(i + 1) / (size_t)n) * 64;
        if (hi > rows) hi = rows;
        len[i] = hi - lo;
        ma_ctx* c = ma_group_ctx(g, i);
        void *pi = NULL, *pf = NULL;
        CHECK(ma_dev_alloc(c, len[i] * 8 + 64, &pi));
        CHECK(ma_dev_alloc(c, len[i] * 8 + 64, &pf));
        CHECK(ma_synth_iota_i64(c, (int64_t*)pi, len[i], (int64_t)lo)); /* v[row] = row, as in the reference's bench (:103, :115) */
        CHECK(ma_synth_iota_f64(c, (double*)pf, len[i], (int64_t)lo));
        col_i[i] = pi;
        col_f[i] = pf;
        lo = hi;
    }

    /* prove the machinery first */
    ma_selftest_report rep;
    ma_status st = ma_group_selftest(g, 0, 20000.0, &rep);
    printf("self-test: %s\n", rep.text);
    if (st != MA_OK && ma_group_is_broken(g) == 1) CHECK(ma_group_rebuild_exchange(g, next_notch(ma_group_flags(g))));
    else if (st != MA_OK) return 1;

    /* both columns of a step in ONE launch per GPU, their totals in record slot 0; then one exchange */
    const int32_t slots[2] = {0, 0}, formats[2] = {'l', 'g'};
    const void* const* const data[2] = {(const void* const*)col_i, (const void* const*)col_f};
    const size_t* const lens[2] = {len, len};
    const int64_t want_i = (int64_t)(rows * (rows - 1) / 2);
    const double want_f = (double)want_i; /* exactly rounded: the f64 total must be within 1 ULP of it */
    int rebuilt = 0, stalled = 0;
    for (int step = 0; step < steps; ++step) {
        const int stall_now = step == 2 && !stalled && ma_test_hooks_enabled();
        if (stall_now) { /* a peer that never arrives, as the waiting host sees it (once) */
            CHECK(ma_group_test_stall_next_exchange(g, n - 1));
            stalled = 1;
        }
        CHECK(ma_group_enqueue_sum_table(g, 2, slots, formats, data, lens, NULL, NULL));
        CHECK(ma_group_exchange(g));
        st = ma_group_synchronize_for(g, stall_now ? deadline_ms : 20000.0);
        if (st != MA_OK) {
            printf("step %d: %s\n", step, ma_last_error_string());
            if (ma_group_is_broken(g) != 1) return 1; /* a stream never ran empty: nothing to rebuild on */
            const uint32_t down = next_notch(ma_group_flags(g));
            CHECK(ma_group_rebuild_exchange(g, down));
            printf("step %d: going on one notch down: %s\n", step, form_name(ma_group_flags(g)));
            ++rebuilt;
            --step; /* the step is run again */
            continue;
        }
        for (int m = 0; m < n; ++m) { /* every GPU holds the job's finals, bit for bit */
            int64_t si = 0;
            uint64_t ci = 0, cf = 0;
            double sf = 0;
            CHECK(ma_group_member_result(g, m, 0, &si, &ci, &sf, &cf));
            if (si != want_i || ci != rows || cf != rows || fabs(sf - want_f) > want_f * ldexp(1.0, -52))
                return fprintf(stderr, "step %d, GPU %d: %" PRId64 " / %" PRIu64 " / %.17g / %" PRIu64 "\n", step, m, si, ci, sf, cf), 1;
        }
    }
    if (rebuilt != 1) return fprintf(stderr, "expected exactly one rebuild, saw %d\n", rebuilt), 1;
    printf("ok: %d steps of 2 x %zu rows over %d GPU(s), one deadline met and survived; running as: %s\n", steps, rows, n,
           form_name(ma_group_flags(g)));
    for (int i = 0; i < n; ++i) {
        CHECK(ma_dev_free(ma_group_ctx(g, i), (void*)col_i[i]));
        CHECK(ma_dev_free(ma_group_ctx(g, i), (void*)col_f[i]));
    }
    ma_group_destroy(g);
    return 0;
}
