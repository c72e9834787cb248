/* A chunked column on the GPU through the C ABI alone (C99, no HIP headers): the calls a Rust / C host makes for
 *   - the 1B-row sum of the reference's bench (benches/benchmark_parallel_simd.rs:81-98) over one resident slice,
 *   - the same column held as 8192-row chunks (RechunkStrategy::Auto, src/structs/chunked/super_array.rs:51-59):
 *     its total in one call (ma_sum_chunks), chunk (+) scalar for every chunk in one launch
 *     (broadcast_superarray_to_scalar, src/kernels/broadcast/super_array.rs:87-116), and the consolidated copy
 *     (SuperTable::consolidate, src/structs/chunked/super_table.rs:677-743).
 * Build:  gcc -std=c99 -Iinclude examples/chunked_column.c -Lminarrow_amd/lib -lminarrow_hip -Wl,-rpath,$PWD/minarrow_amd/lib
 * Exit code 0 and "ok" when every result matches its closed form; 2 when no GPU is visible. */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>

#include "minarrow_hip.h"

#define CHECK(call)                                                                                  \
    do {                                                                                             \
        ma_status st_ = (call);                                                                      \
        if (st_ != MA_OK) {                                                                          \
            fprintf(stderr, "%s: %s: %s\n", #call, ma_status_name(st_), ma_last_error_string());     \
            return 1;                                                                                \
        }                                                                                            \
    } while (0)

int main(int argc, char** argv) {
    const size_t rows = argc > 1 ? (size_t)strtoull(argv[1], NULL, 10) : ((size_t)1 << 24);
    const size_t chunk_rows = 8192;
    if (ma_device_count() <= 0) {
        printf("no HIP device is visible\n");
        return 2;
    }
    ma_ctx* ctx = NULL;
    CHECK(ma_ctx_create(0, &ctx));
    void *col = NULL, *out = NULL, *joined = NULL;
    CHECK(ma_dev_alloc(ctx, rows * 8 + 64, &col));
    CHECK(ma_dev_alloc(ctx, rows * 8 + 64, &out));
    CHECK(ma_dev_alloc(ctx, rows * 8 + 64, &joined));
    CHECK(ma_synth_iota_i64(ctx, (int64_t*)col, rows, 0)); /* 0, 1, 2, ... : sums have closed forms */

    /* the resident slice in one call */
    int64_t sum = 0;
    uint64_t count = 0;
    CHECK(ma_i64_sum(ctx, (const int64_t*)col, rows, NULL, 0, 0, &sum, &count));
    const int64_t want = (int64_t)(rows * (rows - 1) / 2);
    if (sum != want || count != rows) return fprintf(stderr, "ma_i64_sum: %" PRId64 " / %" PRIu64 "\n", sum, count), 1;

    /* the same column as a list of 8192-row chunks: pointer tables are all the host builds */
    const size_t n_chunks = (rows + chunk_rows - 1) / chunk_rows;
    const void** chunk = malloc(n_chunks * sizeof *chunk);
    void** chunk_out = malloc(n_chunks * sizeof *chunk_out);
    size_t* len = malloc(n_chunks * sizeof *len);
    if (!chunk || !chunk_out || !len) return 1;
    for (size_t i = 0; i < n_chunks; ++i) {
        chunk[i] = (const char*)col + i * chunk_rows * 8;
        chunk_out[i] = (char*)out + i * chunk_rows * 8;
        len[i] = i + 1 < n_chunks ? chunk_rows : rows - i * chunk_rows;
    }
    double fsum = 0;
    CHECK(ma_sum_chunks(ctx, 'l', n_chunks, chunk, len, NULL, NULL, &fsum, &sum, &count));
    if (sum != want || count != rows) return fprintf(stderr, "ma_sum_chunks: %" PRId64 " / %" PRIu64 "\n", sum, count), 1;

    /* every chunk + 5 in one launch; then the chunks of the result joined into one column */
    const int64_t five = 5;
    CHECK(ma_broadcast_super_array_scalar(ctx, 'l', MA_OP_ADD, 0, &five, n_chunks, chunk, len, NULL, chunk_out, NULL, NULL));
    int32_t has_mask = 0;
    CHECK(ma_consolidate_column(ctx, 8, n_chunks, (const void* const*)chunk_out, len, NULL, NULL, joined, NULL, &has_mask));
    CHECK(ma_i64_sum(ctx, (const int64_t*)joined, rows, NULL, 0, 0, &sum, &count));
    if (sum != want + 5 * (int64_t)rows || has_mask) return fprintf(stderr, "chunk + 5: %" PRId64 "\n", sum), 1;

    printf("ok: %zu rows as %zu chunks, sum %" PRId64 "\n", rows, n_chunks, want);
    free(chunk);
    free(chunk_out);
    free(len);
    CHECK(ma_dev_free(ctx, joined));
    CHECK(ma_dev_free(ctx, out));
    CHECK(ma_dev_free(ctx, col));
    ma_ctx_destroy(ctx);
    return 0;
}
