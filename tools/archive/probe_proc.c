/* One torch-free host process, as a Rust / C host would be: what read rate does ma_i64_sum see on each block it allocates,
 * on each of several contexts (each owns one HIP stream = one HSA queue), and what did the runtime set this process up
 * with (loaded runtime libraries, KFD queue properties, clocks)? One JSON line per process; tools/run_probe_proc.sh runs it
 * many times under different runtimes / environments (round 4: the 7.3-vs-6.9 TB/s per-process bimodality,
 * profiles/r03_read_rate_by_allocation.txt).
 * Build: gcc -std=gnu99 -O2 -Iinclude tools/probe_proc.c -Lminarrow_amd/lib -lminarrow_hip -Wl,-rpath,$PWD/minarrow_amd/lib -o build/probe_proc */
#include <dirent.h>
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "minarrow_hip.h"
#include "probe_sensors.h"

#define CHECK(call)                                                                              \
    do {                                                                                         \
        ma_status st_ = (call);                                                                  \
        if (st_ != MA_OK) {                                                                      \
            fprintf(stderr, "%s: %s: %s\n", #call, ma_status_name(st_), ma_last_error_string()); \
            return 1;                                                                            \
        }                                                                                        \
    } while (0)

static double time_sum(ma_ctx* ctx, const int64_t* col, size_t n, void* slot, int reps) {
    float best = 1e9f;
    for (int w = 0; w < 2; ++w) ma_i64_sum(ctx, col, n, NULL, 0, 0, (int64_t*)slot, (uint64_t*)slot + 1);
    for (int trial = 0; trial < 3; ++trial) {
        float ms = 0;
        ma_ctx_synchronize(ctx);
        ma_ctx_timer_start(ctx);
        for (int r = 0; r < reps; ++r) ma_i64_sum(ctx, col, n, NULL, 0, 0, (int64_t*)slot, (uint64_t*)slot + 1);
        ma_ctx_timer_stop(ctx);
        ma_ctx_timer_elapsed_ms(ctx, &ms);
        if (ms / reps < best) best = ms / reps;
    }
    return best;
}

static void cat_first_line(const char* path, char* out, size_t cap) {
    out[0] = 0;
    FILE* f = fopen(path, "r");
    if (!f) return;
    if (fgets(out, (int)cap, f)) out[strcspn(out, "\n")] = 0;
    fclose(f);
}

int main(int argc, char** argv) {
    const size_t n = argc > 1 ? (size_t)strtoull(argv[1], NULL, 10) : (size_t)1000000000;
    const int n_blocks = argc > 2 ? atoi(argv[2]) : 2;
    const int n_ctx = argc > 3 ? atoi(argv[3]) : 3;
    const int reps = argc > 4 ? atoi(argv[4]) : 10;
    const char* tag = getenv("PROBE_TAG") ? getenv("PROBE_TAG") : "";
    if (ma_device_count() <= 0) return printf("{\"error\": \"no device\"}\n"), 2;
    ma_ctx* ctx[8] = {0};
    void* blk[8] = {0};
    void* slot[8] = {0};
    for (int c = 0; c < n_ctx && c < 8; ++c) {
        CHECK(ma_ctx_create(0, &ctx[c]));
        CHECK(ma_ctx_set_async(ctx[c], 1));
        CHECK(ma_dev_alloc(ctx[c], 256, &slot[c]));
    }
    for (int b = 0; b < n_blocks && b < 8; ++b) {
        CHECK(ma_dev_alloc(ctx[0], n * 8, &blk[b]));
        CHECK(ma_synth_iota_i64(ctx[0], (int64_t*)blk[b], n, b));
    }
    CHECK(ma_ctx_synchronize(ctx[0]));
    find_hwmon();
    printf("{\"tag\": \"%s\", \"pid\": %d, \"rows\": %zu, \"rates_tbps\": [", tag, (int)getpid(), n);
    char sens[4096] = "";
    size_t sl = 0;
    double lo = 1e9, hi = 0;
    for (int c = 0; c < n_ctx; ++c) {
        printf("%s[", c ? ", " : "");
        for (int b = 0; b < n_blocks; ++b) {
            const double ms = time_sum(ctx[c], (const int64_t*)blk[b], n, slot[c], reps);
            const double tbps = 8.0 * (double)n / ms / 1e9;
            if (tbps < lo) lo = tbps;
            if (tbps > hi) hi = tbps;
            printf("%s%.3f", b ? ", " : "", tbps);
            if (g_n_hwmon && sl + 96 < sizeof sens) {
                char q[400];
                snprintf(q, sizeof q, "%s/freq1_input", g_hwmon[0]);
                const long sclk = read_long(q);
                snprintf(q, sizeof q, "%s/power1_input", g_hwmon[0]);
                const long pw = read_long(q);
                snprintf(q, sizeof q, "%s/temp3_input", g_hwmon[0]);
                const long tm = read_long(q);
                sl += (size_t)snprintf(sens + sl, sizeof sens - sl, "%s[%ld, %ld, %ld]", sl ? ", " : "", sclk / 1000000, pw / 1000000, tm / 1000);
            }
        }
        printf("]");
    }
    printf("], \"sclk_mhz_power_w_hbm_c\": [%s], \"min\": %.3f, \"max\": %.3f, \"blocks\": [", sens, lo, hi);
    for (int b = 0; b < n_blocks; ++b) printf("%s\"%p\"", b ? ", " : "", blk[b]);
    printf("]");
    /* which runtime libraries this process really runs on */
    {
        FILE* f = fopen("/proc/self/maps", "r");
        char line[1024], hip[512] = "", hsa[512] = "", drm[512] = "";
        while (f && fgets(line, sizeof line, f)) {
            char* p = strchr(line, '/');
            if (!p) continue;
            p[strcspn(p, "\n")] = 0;
            if (strstr(p, "libamdhip64") && !hip[0]) snprintf(hip, sizeof hip, "%s", p);
            if (strstr(p, "libhsa-runtime64") && !hsa[0]) snprintf(hsa, sizeof hsa, "%s", p);
            if (strstr(p, "libdrm_amdgpu") && !drm[0]) snprintf(drm, sizeof drm, "%s", p);
        }
        if (f) fclose(f);
        printf(", \"libamdhip64\": \"%s\", \"libhsa\": \"%s\", \"libdrm_amdgpu\": \"%s\"", hip, hsa, drm);
    }
    /* KFD's view of this process's queues (the directory is named after the pid KFD knows; try ours and every readable one) */
    {
        DIR* d = opendir("/sys/class/kfd/kfd/proc");
        struct dirent* e;
        printf(", \"kfd_procs\": [");
        int first = 1;
        while (d && (e = readdir(d))) {
            if (e->d_name[0] == '.') continue;
            char qdir[512];
            snprintf(qdir, sizeof qdir, "/sys/class/kfd/kfd/proc/%s/queues", e->d_name);
            DIR* q = opendir(qdir);
            if (!q) continue;
            printf("%s{\"pid\": \"%s\", \"queues\": [", first ? "" : ", ", e->d_name);
            first = 0;
            struct dirent* qe;
            int qfirst = 1;
            while ((qe = readdir(q))) {
                if (qe->d_name[0] == '.') continue;
                char p[1024], type[64], size[64], gpuid[64];
                snprintf(p, sizeof p, "%s/%s/type", qdir, qe->d_name);
                cat_first_line(p, type, sizeof type);
                snprintf(p, sizeof p, "%s/%s/size", qdir, qe->d_name);
                cat_first_line(p, size, sizeof size);
                snprintf(p, sizeof p, "%s/%s/gpuid", qdir, qe->d_name);
                cat_first_line(p, gpuid, sizeof gpuid);
                printf("%s{\"id\": \"%s\", \"type\": \"%s\", \"size\": \"%s\", \"gpuid\": \"%s\"}", qfirst ? "" : ", ", qe->d_name, type,
                       size, gpuid);
                qfirst = 0;
            }
            closedir(q);
            printf("]}");
        }
        if (d) closedir(d);
        printf("]");
    }
    /* current clock levels of every card (the busy one shows its active level) */
    {
        printf(", \"clocks\": [");
        int first = 1;
        for (int card = 0; card < 64; ++card) {
            const char* names[] = {"pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk"};
            char cur[3][160] = {"", "", ""};
            int any = 0;
            for (int k = 0; k < 3; ++k) {
                char p[256], line[128];
                snprintf(p, sizeof p, "/sys/class/drm/card%d/device/%s", card, names[k]);
                FILE* f = fopen(p, "r");
                if (!f) continue;
                while (fgets(line, sizeof line, f))
                    if (strchr(line, '*')) {
                        line[strcspn(line, "\n")] = 0;
                        snprintf(cur[k], sizeof cur[k], "%s", line);
                        any = 1;
                    }
                fclose(f);
            }
            if (any) {
                printf("%s{\"card\": %d, \"sclk\": \"%s\", \"mclk\": \"%s\", \"fclk\": \"%s\"}", first ? "" : ", ", card, cur[0], cur[1],
                       cur[2]);
                first = 0;
            }
        }
        printf("]");
    }
    printf("}\n");
    for (int b = 0; b < n_blocks; ++b) ma_dev_free(ctx[0], blk[b]);
    for (int c = 0; c < n_ctx; ++c) {
        ma_dev_free(ctx[c], slot[c]);
        ma_ctx_destroy(ctx[c]);
    }
    return 0;
}
