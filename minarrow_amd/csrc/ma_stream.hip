// Arrow C Stream ingestion: sum / valid-count of one column of a stream of record batches, chunk by chunk —
// how a SuperTable travels over the reference's C Stream interface (src/ffi/arrow_c_ffi.rs:160-184 ArrowArrayStream,
// :2104-2260 export/import of chunked tables).
//
// Per batch: the column's values (and the bytes of its validity window) go host -> device slot with the runtime's
// pageable-copy path (measured 55 GB/s, i.e. PCIe Gen5 x16 line rate: profiles/r01_pcie.json — faster than a
// hand-rolled threaded memcpy into pinned memory, 24 GB/s), the batch is released, and the sum kernel is enqueued; it
// runs (microseconds) while the host pulls the next batch from the producer. Per-batch partials are double-double /
// wrapping-i64 and folded on the host in batch order, so the total obeys the same bounds as a single-array sum
// (DESIGN.md §3.1).
#include "ma_common.hpp"

#ifndef ARROW_C_STREAM_INTERFACE
#error "minarrow_hip.h must define the Arrow C Stream interface"
#endif

using namespace ma;

namespace {

// Device slots come from (and go back to) the device's block cache: a hipMalloc + hipFree pair per stream call would stall
// every stream of the device in hipFree (~160 us) — as long as a whole short stream takes.
ma_status grow_dev(void** p, size_t* cap, size_t need) {
    if (need <= *cap) return MA_OK;
    int dev = 0;
    MA_HIP(hipGetDevice(&dev));
    if (*p) {
        MA_HIP(hipDeviceSynchronize());  // the old slot may still be read by a kernel in flight
        MA_HIP(device_block_free(dev, *p));
    }
    *p = nullptr;
    size_t bytes = need + need / 2 + 4096;
    MA_HIP(device_block_alloc(dev, p, bytes));
    *cap = bytes;
    return MA_OK;
}

// Small batches (a SuperTable rechunked at RechunkStrategy::Auto travels as 8192-row record batches: 64 KiB per column)
// are gathered: their values are copied into one of two pinned 8-MiB tiles, their validity bits are appended to the tile's
// bitmap at bit granularity (batches without a bitmap contribute valid bits once any batch of the tile has one), and a
// full tile goes to the GPU as ONE copy and ONE sum — a copy, two synchronisations and a launch per 64-KiB batch cost
// 30-44 us each: 1.5-2.2 GB/s for 8192-row batches, 9-11 GB/s for 65 536-row ones; gathered 43-48 and 36-47 GB/s
// (tools/bench_stream_ingest.py 8192 20000 / 65536 2000) — the host's memcpy into pinned memory. Batches of 4 MiB and more keep
// the direct path above (52-54 GB/s).
constexpr size_t kSmallBatchBytes = (size_t)4 << 20;  // the direct path's ~35 us per batch equal the gather's memcpy here
constexpr size_t kTileBytes = (size_t)8 << 20;  // 16 MiB gathers 20 % slower (the tile falls out of the host caches)

struct GatherTile {
    char* values = nullptr;      // pinned, kTileBytes
    uint64_t* bits = nullptr;    // pinned, one bit per row of the tile (+ a word of slack)
    uint64_t* record = nullptr;  // pinned: [0] sum / hi, [1] lo, [2] count of the tile last sent from here
    hipEvent_t done = nullptr;   // behind that tile's sum kernel
    bool in_flight = false;
    bool masked = false;
    size_t rows = 0;
};

inline void two_sum_acc(double& hi, double& lo, double h, double l) {
    double t = hi + h;
    double bp = t - hi;
    double e = (hi - (t - bp)) + (h - bp);
    hi = t;
    lo += e + l;
}

}  // namespace

extern "C" ma_status ma_sum_arrow_stream(ma_ctx* ctx, struct ArrowArrayStream* stream, int64_t column, double* out_sum_f64,
                                         int64_t* out_sum_i64, uint64_t* out_valid_count, uint64_t* out_rows,
                                         uint64_t* out_batches) {
    MA_REQUIRE(ctx != nullptr && stream != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx or stream is NULL");
    MA_REQUIRE(stream->get_schema && stream->get_next, MA_ERR_INVALID_ARGUMENT, "ArrowArrayStream callbacks are NULL");
    struct ArrowSchema schema;
    memset(&schema, 0, sizeof(schema));
    if (stream->get_schema(stream, &schema) != 0) {
        set_error("ArrowArrayStream.get_schema failed: %s",
                  stream->get_last_error ? stream->get_last_error(stream) : "(no message)");
        return MA_ERR_INVALID_ARGUMENT;
    }
    // which column: a "+s" (struct = record batch) stream needs a child index; a primitive stream uses -1
    const struct ArrowSchema* col_schema = &schema;
    const bool is_struct = schema.format && schema.format[0] == '+' && schema.format[1] == 's';
    ma_status st = MA_OK;
    char code = 0;
    size_t esz = 0;
    if (is_struct) {
        if (column < 0 || column >= schema.n_children) {
            set_error("column %lld out of range for a record batch stream with %lld columns", (long long)column,
                      (long long)schema.n_children);
            st = MA_ERR_INVALID_ARGUMENT;
        } else {
            col_schema = schema.children[column];
        }
    } else if (column > 0) {
        set_error("a primitive stream has no column %lld", (long long)column);
        st = MA_ERR_INVALID_ARGUMENT;
    }
    if (st == MA_OK) {
        const char* f = col_schema->format;
        if (!f || f[0] == 0 || f[1] != 0 || !strchr("iIlLfg", f[0]) || col_schema->dictionary) {
            set_error("unsupported Arrow format \"%s\" (numeric primitives only)", f ? f : "(null)");
            st = MA_ERR_UNSUPPORTED;
        } else {
            code = f[0];
            esz = (code == 'l' || code == 'L' || code == 'g') ? 8 : 4;
        }
    }
    if (schema.release) schema.release(&schema);
    if (st != MA_OK) return st;

    const bool is_float = code == 'f' || code == 'g';
    void* d_values = nullptr;
    size_t values_cap = 0;
    void* d_mask = nullptr;
    size_t mask_cap = 0;
    uint64_t* record = nullptr;  // pinned: [0] sum / hi, [1] lo, [2] count — written by the batch's kernel
    bool pending = false;
    double hi = 0.0, lo = 0.0;
    uint64_t isum = 0, count = 0, rows = 0, batches = 0;

    auto fold_record = [&](const uint64_t* rec) {
        if (is_float) {
            double h, l;
            memcpy(&h, &rec[0], 8);
            memcpy(&l, &rec[1], 8);
            two_sum_acc(hi, lo, h, l);
        } else {
            isum += rec[0];
        }
        count += rec[2];
    };
    auto fold = [&]() {
        if (!pending) return;
        pending = false;
        fold_record(record);
    };
    GatherTile tiles[2];
    int cur = 0;
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(ctx->stream);
        if (d_values) (void)device_block_free(ctx->device, d_values);
        if (d_mask) (void)device_block_free(ctx->device, d_mask);
        if (record) (void)ma_free_pinned(record);
        for (GatherTile& t : tiles) {
            if (t.values) (void)ma_free_pinned(t.values);
            if (t.bits) (void)ma_free_pinned(t.bits);
            if (t.record) (void)ma_free_pinned(t.record);
            if (t.done) (void)hipEventDestroy(t.done);
        }
    };
    auto sum_into = [&](const void* values, size_t n, const uint8_t* m, size_t m_off, uint64_t* rec) -> ma_status {
        const int64_t nc = m ? -1 : 0;
        rec[0] = rec[1] = rec[2] = 0;
        switch (code) {
            case 'l': return ma_i64_sum(ctx, (const int64_t*)values, n, m, m_off, nc, (int64_t*)&rec[0], &rec[2]);
            case 'L': return ma_u64_sum(ctx, (const uint64_t*)values, n, m, m_off, nc, &rec[0], &rec[2]);
            case 'i': return ma_i32_sum(ctx, (const int32_t*)values, n, m, m_off, nc, (int64_t*)&rec[0], &rec[2]);
            case 'I': return ma_u32_sum(ctx, (const uint32_t*)values, n, m, m_off, nc, &rec[0], &rec[2]);
            case 'f': return ma_f32_sum_dd(ctx, (const float*)values, n, m, m_off, nc, (double*)&rec[0], (double*)&rec[1], &rec[2]);
            default: return ma_f64_sum_dd(ctx, (const double*)values, n, m, m_off, nc, (double*)&rec[0], (double*)&rec[1], &rec[2]);
        }
    };
    const size_t tile_rows_cap = kTileBytes / (esz ? esz : 8);
    const size_t tile_bits_bytes = (tile_rows_cap / 64 + 2) * 8;
    // a tile is reusable once the sum of what was last sent from it has finished (its record is then folded)
    auto wait_tile = [&](GatherTile& t) -> ma_status {
        if (!t.in_flight) return MA_OK;
        MA_HIP(hipEventSynchronize(t.done));
        t.in_flight = false;
        fold_record(t.record);
        return MA_OK;
    };
    auto prepare_tile = [&](GatherTile& t) -> ma_status {
        if (t.values) return MA_OK;
        // recycled pinned blocks (ma_alloc64_pinned): pinning 2 x 8 MiB afresh would cost every call ~2 ms
        MA_TRY(ma_alloc64_pinned(kTileBytes, (void**)&t.values));
        MA_TRY(ma_alloc64_pinned(tile_bits_bytes, (void**)&t.bits));
        MA_TRY(ma_alloc64_pinned(64, (void**)&t.record));
        MA_HIP(hipEventCreateWithFlags(&t.done, hipEventDisableTiming));
        return MA_OK;
    };
    // the gathered rows of the current tile -> device (one copy), one sum; the other tile becomes current
    auto flush_tile = [&]() -> ma_status {
        GatherTile& t = tiles[cur];
        if (t.rows == 0) return MA_OK;
        MA_TRY(grow_dev(&d_values, &values_cap, kTileBytes + 64));  // stream-ordered re-use: copies and sums share ctx->stream
        MA_HIP(hipMemcpyAsync(d_values, t.values, t.rows * esz, hipMemcpyHostToDevice, ctx->stream));
        const uint8_t* m = nullptr;
        if (t.masked) {
            MA_TRY(grow_dev(&d_mask, &mask_cap, tile_bits_bytes + 16));
            MA_HIP(hipMemcpyAsync(d_mask, t.bits, ((t.rows + 63) / 64) * 8 + 8, hipMemcpyHostToDevice, ctx->stream));
            m = (const uint8_t*)d_mask;
        }
        MA_TRY(sum_into(d_values, t.rows, m, 0, t.record));
        MA_HIP(hipEventRecord(t.done, ctx->stream));
        t.in_flight = true;
        t.rows = 0;
        t.masked = false;
        cur ^= 1;
        return wait_tile(tiles[cur]);
    };
    auto gather = [&](const void* values, size_t n, const uint8_t* validity, size_t validity_bytes, size_t off) -> ma_status {
        if (tiles[cur].rows + n > tile_rows_cap) MA_TRY(flush_tile());
        GatherTile& t = tiles[cur];
        MA_TRY(prepare_tile(t));
        memcpy(t.values + t.rows * esz, (const char*)values + off * esz, n * esz);
        if (validity && !t.masked) {  // the first batch with nulls in this tile: the rows gathered so far are all valid
            memset(t.bits, 0, tile_bits_bytes);
            append_bits(t.bits, 0, nullptr, 0, 0, t.rows);
            t.masked = true;
        }
        if (t.masked) append_bits(t.bits, t.rows, validity, validity_bytes, off, n);
        t.rows += n;
        return MA_OK;
    };

    // One lane of the context for the whole stream (the per-batch reductions re-use it); they only enqueue (ma::NoSync)
    // — the context's user-visible mode is never touched, so other threads sharing it are unaffected.
    MA_ENTER(ctx);
    NoSync enqueue_only;
    MA_NO_CAPTURE(ctx, "ma_sum_arrow_stream");
    MA_HIP(hipSetDevice(ctx->device));
    hipError_t he = hipSuccess;
    MA_TRY(ma_alloc64_pinned(64, (void**)&record));  // pooled like the tiles: a fresh pin is ~100 us of a 600-us short stream

    for (;;) {
        struct ArrowArray batch;
        memset(&batch, 0, sizeof(batch));
        if (stream->get_next(stream, &batch) != 0) {
            set_error("ArrowArrayStream.get_next failed: %s",
                      stream->get_last_error ? stream->get_last_error(stream) : "(no message)");
            st = MA_ERR_INVALID_ARGUMENT;
            break;
        }
        if (batch.release == nullptr) break;  // end of stream
        const struct ArrowArray* col = &batch;
        if (is_struct) {
            if (column >= batch.n_children) {
                set_error("batch %llu has %lld columns, column %lld requested", (unsigned long long)batches,
                          (long long)batch.n_children, (long long)column);
                st = MA_ERR_INVALID_ARGUMENT;
            } else {
                col = batch.children[column];
            }
        }
        if (st == MA_OK && (col->n_buffers != 2 || !col->buffers || (col->length > 0 && !col->buffers[1]))) {
            set_error("batch %llu: not a primitive array (n_buffers = %lld)", (unsigned long long)batches,
                      (long long)col->n_buffers);
            st = MA_ERR_INVALID_ARGUMENT;
        }
        if (st != MA_OK) {
            batch.release(&batch);
            break;
        }
        const size_t n = (size_t)col->length;
        // a struct array's own offset shifts every child (Arrow C Data Interface); PyArrow exports 0 here
        const size_t off = (size_t)col->offset + (is_struct ? (size_t)batch.offset : 0);
        const uint8_t* validity = col->null_count == 0 ? nullptr : (const uint8_t*)col->buffers[0];
        size_t mask_off = 0;
        if (n && n * esz < kSmallBatchBytes) {  // gathered: nothing is enqueued until a tile is full
            st = gather(col->buffers[1], n, validity, (off + n + 7) >> 3, off);
            batch.release(&batch);
            if (st != MA_OK) break;
            rows += n;
            ++batches;
            continue;
        }
        if (n) st = flush_tile();  // a large batch: what was gathered goes first, then the direct path
        // The previous batch's kernel reads the device slots: it must finish before they are overwritten (it has
        // had the whole get_next() of this batch to do so).
        he = hipStreamSynchronize(ctx->stream);
        if (he != hipSuccess) st = hip_fail(he, "hipStreamSynchronize", __FILE__, __LINE__);
        fold();
        if (st == MA_OK && n) st = grow_dev(&d_values, &values_cap, n * esz + 64);
        if (st == MA_OK && n && validity) {
            const size_t first = off >> 3, end = (off + n + 7) >> 3;
            st = grow_dev(&d_mask, &mask_cap, end - first + 16);
            if (st == MA_OK) {
                he = hipMemsetAsync((uint8_t*)d_mask + (end - first), 0, 16, ctx->stream);
                if (he == hipSuccess)
                    he = hipMemcpyAsync(d_mask, validity + first, end - first, hipMemcpyHostToDevice, ctx->stream);
                if (he != hipSuccess) st = hip_fail(he, "validity upload", __FILE__, __LINE__);
                mask_off = off & 7;
            }
        }
        if (st == MA_OK && n) {
            he = hipMemcpyAsync(d_values, (const char*)col->buffers[1] + off * esz, n * esz, hipMemcpyHostToDevice, ctx->stream);
            if (he == hipSuccess) he = hipStreamSynchronize(ctx->stream);  // the producer's memory is about to go away
            if (he != hipSuccess) st = hip_fail(he, "values upload", __FILE__, __LINE__);
        }
        batch.release(&batch);
        if (st != MA_OK) break;
        rows += n;
        ++batches;
        if (n == 0) continue;
        st = sum_into(d_values, n, validity ? (const uint8_t*)d_mask : nullptr, mask_off, record);
        if (st != MA_OK) break;
        pending = true;
    }
    if (st == MA_OK) st = flush_tile();
    if (st == MA_OK) {
        he = hipStreamSynchronize(ctx->stream);
        if (he != hipSuccess) {
            st = hip_fail(he, "hipStreamSynchronize", __FILE__, __LINE__);
        } else {
            fold();
            for (GatherTile& t : tiles)
                if (t.in_flight) {
                    t.in_flight = false;
                    fold_record(t.record);
                }
        }
    }
    cleanup();
    if (st != MA_OK) return st;
    if (is_float) {
        double total = (hi == hi && lo == lo && hi - hi == 0.0 && lo - lo == 0.0) ? hi + lo : hi;
        if (out_sum_f64) *out_sum_f64 = total;
        if (out_sum_i64) *out_sum_i64 = 0;
    } else {
        if (out_sum_i64) *out_sum_i64 = (int64_t)isum;
        if (out_sum_f64) *out_sum_f64 = (code == 'L' || code == 'I') ? (double)isum : (double)(int64_t)isum;
    }
    if (out_valid_count) *out_valid_count = count;
    if (out_rows) *out_rows = rows;
    if (out_batches) *out_batches = batches;
    return MA_OK;
}
