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

ma_status grow_dev(void** p, size_t* cap, size_t need) {
    if (need <= *cap) return MA_OK;
    if (*p) MA_HIP(hipFree(*p));
    *p = nullptr;
    size_t bytes = need + need / 2 + 4096;
    int dev = 0;
    MA_HIP(hipGetDevice(&dev));
    MA_HIP(device_malloc(dev, p, bytes));
    *cap = bytes;
    return MA_OK;
}

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

    auto fold = [&]() {
        if (!pending) return;
        pending = false;
        if (is_float) {
            double h, l;
            memcpy(&h, &record[0], 8);
            memcpy(&l, &record[1], 8);
            two_sum_acc(hi, lo, h, l);
        } else {
            isum += record[0];
        }
        count += record[2];
    };
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(ctx->stream);
        if (d_values) (void)hipFree(d_values);
        if (d_mask) (void)hipFree(d_mask);
        if (record) (void)hipHostFree(record);
    };

    // One lane of the context for the whole stream (the per-batch reductions re-use it); they only enqueue (ma::NoSync)
    // — the context's user-visible mode is never touched, so other threads sharing it are unaffected.
    MA_ENTER(ctx);
    NoSync enqueue_only;
    MA_NO_CAPTURE(ctx, "ma_sum_arrow_stream");
    MA_HIP(hipSetDevice(ctx->device));
    hipError_t he = hipHostMalloc((void**)&record, 64, hipHostMallocPortable | hipHostMallocMapped);
    if (he != hipSuccess) return hip_fail(he, "hipHostMalloc(record)", __FILE__, __LINE__);

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
        const uint8_t* m = validity ? (const uint8_t*)d_mask : nullptr;
        const int64_t nc = validity ? -1 : 0;
        record[0] = record[1] = record[2] = 0;
        switch (code) {
            case 'l': st = ma_i64_sum(ctx, (const int64_t*)d_values, n, m, mask_off, nc, (int64_t*)&record[0], &record[2]); break;
            case 'L': st = ma_u64_sum(ctx, (const uint64_t*)d_values, n, m, mask_off, nc, &record[0], &record[2]); break;
            case 'i': st = ma_i32_sum(ctx, (const int32_t*)d_values, n, m, mask_off, nc, (int64_t*)&record[0], &record[2]); break;
            case 'I': st = ma_u32_sum(ctx, (const uint32_t*)d_values, n, m, mask_off, nc, &record[0], &record[2]); break;
            case 'f': st = ma_f32_sum_dd(ctx, (const float*)d_values, n, m, mask_off, nc, (double*)&record[0], (double*)&record[1], &record[2]); break;
            default: st = ma_f64_sum_dd(ctx, (const double*)d_values, n, m, mask_off, nc, (double*)&record[0], (double*)&record[1], &record[2]); break;
        }
        if (st != MA_OK) break;
        pending = true;
    }
    if (st == MA_OK) {
        he = hipStreamSynchronize(ctx->stream);
        if (he != hipSuccess) st = hip_fail(he, "hipStreamSynchronize", __FILE__, __LINE__);
        else fold();
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
