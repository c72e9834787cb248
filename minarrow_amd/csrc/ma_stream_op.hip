// SuperTable (op) SuperTable as a streaming operator over the Arrow C Stream Interface.
//
// The reference computes it batch by batch — broadcast_super_table_with_operator,
// src/kernels/broadcast/super_table.rs:37-72: chunk counts must match ("SuperTable chunk count mismatch: {} vs {}"),
// batch i of the result = broadcast_table_with_operator(op, lhs_i, rhs_i) — and moves SuperTables across its FFI as
// record-batch streams (src/ffi/arrow_c_ffi.rs:2104-2260, one struct array per batch). Here the two are joined: the
// operator CONSUMES two ArrowArrayStreams and IS an ArrowArrayStream; each get_next pulls one batch from either side,
// runs every column pair on the GPU (ma_apply_arrow_batch_export) and hands back an owned struct array in pinned
// memory. Nothing is materialised beyond the batch in flight, so a table larger than HBM streams through.
#include <cerrno>
#include <new>
#include <string>

#include "ma_common.hpp"

using namespace ma;

namespace {

struct OpStream {
    ma_ctx* ctx = nullptr;
    int32_t op = 0;
    ArrowArrayStream lhs{}, rhs{};    // moved in: released with the operator
    ArrowSchema lhs_schema{}, rhs_schema{};
    bool have_schemas = false;
    uint64_t batches = 0;
    std::string last_error;
};

void release_schema_if(ArrowSchema* s) {
    if (s->release) s->release(s);
}

int fail(OpStream* o, int code, const std::string& msg) {
    o->last_error = msg;
    return code;
}

int fetch_schemas(OpStream* o) {
    if (o->have_schemas) return 0;
    if (o->lhs.get_schema(&o->lhs, &o->lhs_schema) != 0) {
        const char* e = o->lhs.get_last_error ? o->lhs.get_last_error(&o->lhs) : nullptr;
        return fail(o, EIO, std::string("lhs stream get_schema failed: ") + (e ? e : "(no message)"));
    }
    if (o->rhs.get_schema(&o->rhs, &o->rhs_schema) != 0) {
        const char* e = o->rhs.get_last_error ? o->rhs.get_last_error(&o->rhs) : nullptr;
        release_schema_if(&o->lhs_schema);
        return fail(o, EIO, std::string("rhs stream get_schema failed: ") + (e ? e : "(no message)"));
    }
    o->have_schemas = true;
    return 0;
}

int op_get_schema(ArrowArrayStream* self, ArrowSchema* out) {
    OpStream* o = (OpStream*)self->private_data;
    if (int rc = fetch_schemas(o)) return rc;
    // The result schema is what a batch export produces for two EMPTY batches of these schemas: run the routing on
    // zero rows (no kernel is launched for n == 0) and keep the schema.
    const ArrowSchema &ls = o->lhs_schema, &rs = o->rhs_schema;
    if (!ls.format || !rs.format || strcmp(ls.format, "+s") != 0 || strcmp(rs.format, "+s") != 0)
        return fail(o, EINVAL, "both inputs must be record-batch streams (struct arrays, format \"+s\")");
    if (ls.n_children != rs.n_children)
        return fail(o, EINVAL, "Table column count mismatch: " + std::to_string(ls.n_children) + " vs " + std::to_string(rs.n_children));
    const int64_t nc = ls.n_children;
    std::vector<ArrowArray> lkids(nc), rkids(nc);
    std::vector<ArrowArray*> lptr(nc), rptr(nc);
    const void* no_buffers[2] = {nullptr, nullptr};
    for (int64_t c = 0; c < nc; ++c) {
        for (ArrowArray* a : {&lkids[c], &rkids[c]}) {
            memset(a, 0, sizeof(*a));
            a->n_buffers = 2;
            a->buffers = no_buffers;
        }
        lptr[c] = &lkids[c];
        rptr[c] = &rkids[c];
    }
    ArrowArray lb{}, rb{};
    for (ArrowArray* b : {&lb, &rb}) {
        b->n_buffers = 1;
        b->buffers = no_buffers;
        b->n_children = nc;
    }
    lb.children = lptr.data();
    rb.children = rptr.data();
    ArrowArray empty{};
    ma_status st = ma_apply_arrow_batch_export(o->ctx, o->op, &lb, &ls, &rb, &rs, &empty, out);
    if (st != MA_OK) return fail(o, st == MA_ERR_UNSUPPORTED ? EINVAL : EIO, ma_last_error_string());
    if (empty.release) empty.release(&empty);
    // A result column may carry validity whenever either input column may (ARROW_FLAG_NULLABLE = 2).
    for (int64_t c = 0; c < nc && c < out->n_children; ++c)
        out->children[c]->flags = (ls.children[c]->flags | rs.children[c]->flags) & 2;
    return 0;
}

int op_get_next(ArrowArrayStream* self, ArrowArray* out) {
    OpStream* o = (OpStream*)self->private_data;
    memset(out, 0, sizeof(*out));
    if (int rc = fetch_schemas(o)) return rc;
    ArrowArray l{}, r{};
    if (o->lhs.get_next(&o->lhs, &l) != 0) {
        const char* e = o->lhs.get_last_error ? o->lhs.get_last_error(&o->lhs) : nullptr;
        return fail(o, EIO, std::string("lhs stream get_next failed: ") + (e ? e : "(no message)"));
    }
    if (o->rhs.get_next(&o->rhs, &r) != 0) {
        const char* e = o->rhs.get_last_error ? o->rhs.get_last_error(&o->rhs) : nullptr;
        if (l.release) l.release(&l);
        return fail(o, EIO, std::string("rhs stream get_next failed: ") + (e ? e : "(no message)"));
    }
    const bool l_end = l.release == nullptr, r_end = r.release == nullptr;
    if (l_end && r_end) return 0;  // end of stream: out->release == NULL
    if (l_end != r_end) {          // super_table.rs:46-55
        if (l.release) l.release(&l);
        if (r.release) r.release(&r);
        return fail(o, EINVAL, "SuperTable chunk count mismatch: " + std::to_string(o->batches + (l_end ? 0 : 1)) + " vs " +
                                   std::to_string(o->batches + (r_end ? 0 : 1)) + " (one stream ended first)");
    }
    ArrowSchema result_schema{};
    ma_status st = ma_apply_arrow_batch_export(o->ctx, o->op, &l, &o->lhs_schema, &r, &o->rhs_schema, out, &result_schema);
    std::string msg = st == MA_OK ? "" : ma_last_error_string();
    l.release(&l);  // the inputs of this batch are consumed (the result owns its own pinned buffers)
    r.release(&r);
    if (st != MA_OK) {
        memset(out, 0, sizeof(*out));
        return fail(o, (st == MA_ERR_UNSUPPORTED || st == MA_ERR_LENGTH_MISMATCH || st == MA_ERR_INVALID_ARGUMENT) ? EINVAL : EIO,
                    "batch " + std::to_string(o->batches) + ": " + msg);
    }
    release_schema_if(&result_schema);
    ++o->batches;
    return 0;
}

const char* op_get_last_error(ArrowArrayStream* self) {
    OpStream* o = (OpStream*)self->private_data;
    return o && !o->last_error.empty() ? o->last_error.c_str() : nullptr;
}

void op_release(ArrowArrayStream* self) {
    if (!self || !self->release) return;
    OpStream* o = (OpStream*)self->private_data;
    if (o) {
        if (o->have_schemas) {
            release_schema_if(&o->lhs_schema);
            release_schema_if(&o->rhs_schema);
        }
        if (o->lhs.release) o->lhs.release(&o->lhs);
        if (o->rhs.release) o->rhs.release(&o->rhs);
        delete o;
    }
    self->private_data = nullptr;
    self->release = nullptr;
}

}  // namespace

extern "C" ma_status ma_apply_arrow_stream_export(ma_ctx* ctx, int32_t op, struct ArrowArrayStream* lhs_stream,
                                                  struct ArrowArrayStream* rhs_stream, struct ArrowArrayStream* out_stream) {
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    MA_REQUIRE(lhs_stream && rhs_stream && out_stream, MA_ERR_INVALID_ARGUMENT, "a stream pointer is NULL");
    MA_REQUIRE(lhs_stream->release && rhs_stream->release, MA_ERR_INVALID_ARGUMENT, "an input stream is already released");
    MA_REQUIRE(lhs_stream->get_schema && lhs_stream->get_next && rhs_stream->get_schema && rhs_stream->get_next,
               MA_ERR_INVALID_ARGUMENT, "an input stream lacks get_schema / get_next");
    MA_REQUIRE(op >= MA_OP_ADD && op <= MA_OP_FLOORDIV, MA_ERR_INVALID_ARGUMENT, "unknown ArithmeticOperator code %d", op);
    MA_NO_CAPTURE(ctx, "ma_apply_arrow_stream_export");
    OpStream* o = new (std::nothrow) OpStream();
    MA_REQUIRE(o != nullptr, MA_ERR_DEVICE, "out of host memory");
    o->ctx = ctx;
    o->op = op;
    // Move both inputs (Arrow C Stream move semantics: bitwise copy, then mark the source released).
    o->lhs = *lhs_stream;
    o->rhs = *rhs_stream;
    lhs_stream->release = nullptr;
    rhs_stream->release = nullptr;
    out_stream->get_schema = op_get_schema;
    out_stream->get_next = op_get_next;
    out_stream->get_last_error = op_get_last_error;
    out_stream->release = op_release;
    out_stream->private_data = o;
    return MA_OK;
}
