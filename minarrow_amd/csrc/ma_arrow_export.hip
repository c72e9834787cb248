// Results handed back through the Arrow C Data Interface — the producer side of the boundary.
//
// The reference exports with create_arrow_export (src/ffi/arrow_c_ffi.rs:1742-1821): a heap ArrowArray whose
// `private_data` is a Holder keeping the buffers alive, `offset` 0, `n_buffers` 2 for primitives, `null_count` 0
// when there is no validity buffer, 64-byte aligned buffers (check_alignment, :1722-1738) and `release` callbacks
// that drop the Holder (:193-262). The structs produced here follow the same contract; the Holder owns pinned host
// allocations (ma_alloc64_pinned — page aligned, device-mapped) that the kernels wrote directly, and `release`
// returns them with ma_free_pinned. Record batches are struct arrays ("+s") with one child per column, the shape
// the reference's record-batch stream yields (arrow_c_ffi.rs:1823-1834, 2104-2260).
#include <atomic>
#include <new>
#include <string>

#include "ma_common.hpp"

using namespace ma;

namespace {

// One pinned allocation per call, shared by every column it produced (hipHostMalloc costs ~100 us, as much as
// moving a 1 M-row column over PCIe). Children may be moved out of their parent by a consumer, so each array holds
// its own reference; the last release frees the slab.
struct Slab {
    void* base = nullptr;
    std::atomic<long> refs{0};
};

struct ArrayHolder {
    Slab* slab = nullptr;      // owner of the memory `values` / `validity` point into
    void* values = nullptr;
    void* validity = nullptr;  // nullptr when the result carries no validity
    const void* buffers[2] = {nullptr, nullptr};
    std::vector<ArrowArray*> children;  // struct arrays only (each heap-allocated, released with the parent)
};

void slab_unref(Slab* s) {
    if (s && s->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) {
        (void)ma_free_pinned(s->base);
        delete s;
    }
}

struct SchemaHolder {
    std::string format;
    std::string name;
    std::vector<ArrowSchema*> children;
};

void release_array(ArrowArray* a) {
    if (a == nullptr || a->release == nullptr) return;
    ArrayHolder* h = (ArrayHolder*)a->private_data;
    if (h) {
        for (ArrowArray* c : h->children) {
            if (c) {
                if (c->release) c->release(c);
                delete c;
            }
        }
        slab_unref(h->slab);
        delete h;
    }
    a->private_data = nullptr;
    a->release = nullptr;  // the consumer-visible "released" marker of the C Data Interface
}

void release_schema(ArrowSchema* s) {
    if (s == nullptr || s->release == nullptr) return;
    SchemaHolder* h = (SchemaHolder*)s->private_data;
    if (h) {
        for (ArrowSchema* c : h->children) {
            if (c) {
                if (c->release) c->release(c);
                delete c;
            }
        }
        delete h;
    }
    s->private_data = nullptr;
    s->release = nullptr;
}

void fill_schema(ArrowSchema* s, const char* format, const char* name, bool nullable, SchemaHolder* h) {
    h->format = format;
    h->name = name ? name : "";
    memset(s, 0, sizeof(*s));
    s->format = h->format.c_str();
    s->name = h->name.c_str();
    s->metadata = nullptr;
    // ARROW_FLAG_NULLABLE = 2: the value the reference's import tests (arrow_c_ffi.rs:2631) and its stream export
    // writes (:2152, :2303); its array export writes 1 (:1784), which consumers read as DICTIONARY_ORDERED.
    s->flags = nullable ? 2 : 0;
    s->n_children = (int64_t)h->children.size();
    s->children = h->children.empty() ? nullptr : h->children.data();
    s->dictionary = nullptr;
    s->release = release_schema;
    s->private_data = h;
}

size_t format_size(char c) { return (c == 'l' || c == 'L' || c == 'g') ? 8 : 4; }

// Result format of arithmetic_dispatch's type matrix (src/kernels/routing/arithmetic.rs:278-406); 0 = unsupported.
char result_format(const char* lf, const char* rf) {
    if (!lf || !rf || lf[0] == 0 || rf[0] == 0 || lf[1] != 0 || rf[1] != 0) return 0;
    const char l = lf[0], r = rf[0];
    const char* prim = "iIlLfg";
    if (!strchr(prim, l) || !strchr(prim, r)) return 0;
    if (l == r) return l;
    if (l == 'i' && (r == 'g' || r == 'f')) return r;
    if (r == 'i' && (l == 'g' || l == 'f')) return l;
    return 0;
}

struct ColumnPlan {
    char fmt;
    size_t n;
    size_t values_off, validity_off;  // byte offsets into the slab (64-byte aligned, arrow_c_ffi.rs:1722-1738)
};

// Validates one column pair against the routing rules and reserves its place in the slab.
ma_status plan_column(const ArrowArray* lhs, const ArrowSchema* ls, const ArrowArray* rhs, const ArrowSchema* rs, size_t* cursor,
                      ColumnPlan* out) {
    MA_REQUIRE(lhs && ls && rhs && rs, MA_ERR_INVALID_ARGUMENT, "ArrowArray or ArrowSchema is NULL");
    const char fmt = result_format(ls->format, rs->format);
    if (fmt == 0) {
        set_error("Unsupported array type combination for arithmetic operations (\"%s\" vs \"%s\")",
                  ls->format ? ls->format : "(null)", rs->format ? rs->format : "(null)");
        return MA_ERR_UNSUPPORTED;
    }
    MA_REQUIRE(lhs->length >= 0 && rhs->length >= 0, MA_ERR_INVALID_ARGUMENT, "negative length");
    const size_t nl = (size_t)lhs->length, nr = (size_t)rhs->length;
    if (nl != nr && nl != 1 && nr != 1) {
        set_error("cannot broadcast arrays of length %zu and %zu", nl, nr);
        return MA_ERR_LENGTH_MISMATCH;
    }
    out->fmt = fmt;
    out->n = nl == nr ? nl : (nl == 1 ? nr : nl);
    auto reserve = [&](size_t bytes) {
        const size_t at = *cursor;
        *cursor += (bytes + 63) & ~(size_t)63;
        return at;
    };
    out->values_off = reserve(out->n * format_size(fmt));
    out->validity_off = reserve(((out->n + 63) / 64) * 8 + 8);  // reserved even if the routing ends up dense
    return MA_OK;
}

// lhs[c] (op) rhs[c] for n_cols column pairs: ONE pinned slab, every kernel enqueued back to back (the context is
// enqueue-only for the duration: ma::NoSync), ONE synchronise, then the owned ArrowArray / ArrowSchema pairs are filled in.
// null_count follows create_arrow_export (arrow_c_ffi.rs:1750): 0 without a validity buffer, -1 (unknown) with one.
ma_status export_columns(ma_ctx* ctx, int32_t op, size_t n_cols, const ArrowArray* const* lhs, const ArrowSchema* const* ls,
                         const ArrowArray* const* rhs, const ArrowSchema* const* rs, const char* const* names,
                         ArrowArray* const* out, ArrowSchema* const* out_schema) {
    std::vector<ColumnPlan> plan(n_cols);
    size_t bytes = 0;
    for (size_t c = 0; c < n_cols; ++c) MA_TRY(plan_column(lhs[c], ls[c], rhs[c], rs[c], &bytes, &plan[c]));
    Slab* slab = new (std::nothrow) Slab();
    MA_REQUIRE(slab != nullptr, MA_ERR_DEVICE, "out of host memory");
    ma_status st = ma_alloc64_pinned(bytes ? bytes : 64, &slab->base);
    if (st != MA_OK) {
        delete slab;
        return st;
    }
    std::vector<int32_t> has_validity(n_cols, 0);
    ma_status sync = MA_OK;
    {
        // One lane for the whole sequence (the per-column calls re-use it), enqueue-only without touching the context's
        // user-visible mode: other threads sharing the context keep their synchronous semantics meanwhile.
        MA_ENTER(ctx);
        NoSync enqueue_only;
        for (size_t c = 0; c < n_cols && st == MA_OK; ++c)
            st = ma_apply_arrow(ctx, op, lhs[c], ls[c], rhs[c], rs[c], (char*)slab->base + plan[c].values_off,
                                (uint8_t*)slab->base + plan[c].validity_off, &has_validity[c]);
        // everything enqueued so far must drain before the buffers are handed out (or freed); a dense integer division
        // by zero recorded by any column surfaces here
        (void)hipSetDevice(ctx->device);
        sync = sync_and_check(ctx);
    }
    if (st == MA_OK) st = sync;
    if (st != MA_OK) {
        (void)ma_free_pinned(slab->base);
        delete slab;
        return st;
    }
    std::vector<ArrayHolder*> holders(n_cols, nullptr);
    std::vector<SchemaHolder*> sholders(n_cols, nullptr);
    bool oom = false;
    for (size_t c = 0; c < n_cols; ++c) {
        holders[c] = new (std::nothrow) ArrayHolder();
        sholders[c] = new (std::nothrow) SchemaHolder();
        oom = oom || !holders[c] || !sholders[c];
    }
    if (oom) {
        for (size_t c = 0; c < n_cols; ++c) {
            delete holders[c];
            delete sholders[c];
        }
        (void)ma_free_pinned(slab->base);
        delete slab;
        set_error("out of host memory");
        return MA_ERR_DEVICE;
    }
    slab->refs.store((long)n_cols + 1, std::memory_order_relaxed);  // +1: this frame, dropped below
    for (size_t c = 0; c < n_cols; ++c) {
        ArrayHolder* h = holders[c];
        h->slab = slab;
        h->values = (char*)slab->base + plan[c].values_off;
        h->validity = has_validity[c] ? (char*)slab->base + plan[c].validity_off : nullptr;
        h->buffers[0] = h->validity;
        h->buffers[1] = h->values;
        ArrowArray* a = out[c];
        memset(a, 0, sizeof(*a));
        a->length = (int64_t)plan[c].n;
        a->null_count = has_validity[c] ? -1 : 0;
        a->offset = 0;  // arrow_c_ffi.rs:1773
        a->n_buffers = 2;
        a->buffers = h->buffers;
        a->release = release_array;
        a->private_data = h;
        const char f[2] = {plan[c].fmt, 0};
        fill_schema(out_schema[c], f, names[c], has_validity[c] != 0, sholders[c]);
    }
    slab_unref(slab);  // n_cols == 0: frees the (empty) slab
    return MA_OK;
}

ma_status check_struct(const ArrowArray* a, const ArrowSchema* s, const char* side) {
    MA_REQUIRE(a != nullptr && s != nullptr, MA_ERR_INVALID_ARGUMENT, "%s batch or schema is NULL", side);
    MA_REQUIRE(s->format != nullptr && strcmp(s->format, "+s") == 0, MA_ERR_UNSUPPORTED,
               "%s: a record batch is a struct array (format \"+s\"), got \"%s\"", side, s->format ? s->format : "(null)");
    MA_REQUIRE(a->n_children == s->n_children && a->n_children >= 0, MA_ERR_INVALID_ARGUMENT,
               "%s: array has %lld children, schema %lld", side, (long long)a->n_children, (long long)s->n_children);
    MA_REQUIRE(a->n_children == 0 || (a->children != nullptr && s->children != nullptr), MA_ERR_INVALID_ARGUMENT,
               "%s: children table is NULL", side);
    MA_REQUIRE(a->n_buffers < 1 || a->buffers == nullptr || a->buffers[0] == nullptr || a->null_count == 0,
               MA_ERR_UNSUPPORTED, "%s: struct-level validity is not a Table concept", side);
    return MA_OK;
}

}  // namespace

extern "C" {

ma_status ma_apply_arrow_export(ma_ctx* ctx, int32_t op, const struct ArrowArray* lhs, const struct ArrowSchema* lhs_schema,
                                const struct ArrowArray* rhs, const struct ArrowSchema* rhs_schema, const char* name,
                                struct ArrowArray* out_array, struct ArrowSchema* out_schema) {
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    MA_REQUIRE(out_array != nullptr && out_schema != nullptr, MA_ERR_INVALID_ARGUMENT, "output struct is NULL");
    MA_NO_CAPTURE(ctx, "ma_apply_arrow_export");
    out_array->release = nullptr;
    out_schema->release = nullptr;
    const char* nm = name ? name : (lhs_schema ? lhs_schema->name : nullptr);
    return export_columns(ctx, op, 1, &lhs, &lhs_schema, &rhs, &rhs_schema, &nm, &out_array, &out_schema);
}

// Table (op) Table — broadcast_table_with_operator, src/kernels/broadcast/table.rs:31-63: the tables must have the
// same number of columns ("Table column count mismatch: {} vs {}"), column i of the result is
// resolve_binary_arithmetic(op, lhs.cols[i], rhs.cols[i]) under the left table's field name (:55-57), and the result
// carries the left table's name (:62).
ma_status ma_apply_arrow_batch_export(ma_ctx* ctx, int32_t op, const struct ArrowArray* lhs_batch,
                                      const struct ArrowSchema* lhs_schema, const struct ArrowArray* rhs_batch,
                                      const struct ArrowSchema* rhs_schema, struct ArrowArray* out_batch,
                                      struct ArrowSchema* out_schema) {
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    MA_REQUIRE(out_batch != nullptr && out_schema != nullptr, MA_ERR_INVALID_ARGUMENT, "output struct is NULL");
    MA_NO_CAPTURE(ctx, "ma_apply_arrow_batch_export");
    out_batch->release = nullptr;
    out_schema->release = nullptr;
    MA_TRY(check_struct(lhs_batch, lhs_schema, "lhs"));
    MA_TRY(check_struct(rhs_batch, rhs_schema, "rhs"));
    if (lhs_batch->n_children != rhs_batch->n_children) {
        set_error("Table column count mismatch: %lld vs %lld", (long long)lhs_batch->n_children,
                  (long long)rhs_batch->n_children);
        return MA_ERR_LENGTH_MISMATCH;
    }
    const size_t n_cols = (size_t)lhs_batch->n_children;
    ArrayHolder* h = new (std::nothrow) ArrayHolder();
    SchemaHolder* sh = new (std::nothrow) SchemaHolder();
    MA_REQUIRE(h && sh, MA_ERR_DEVICE, "out of host memory");
    // A struct's offset applies to its children on top of their own (Arrow C Data Interface).
    std::vector<ArrowArray> lkids(n_cols), rkids(n_cols);
    std::vector<const ArrowArray*> lp(n_cols), rp(n_cols);
    std::vector<const ArrowSchema*> lsp(n_cols), rsp(n_cols);
    std::vector<const char*> names(n_cols);
    bool oom = false;
    for (size_t c = 0; c < n_cols; ++c) {
        lkids[c] = *lhs_batch->children[c];
        rkids[c] = *rhs_batch->children[c];
        lkids[c].offset += lhs_batch->offset;
        rkids[c].offset += rhs_batch->offset;
        if (lhs_batch->offset || lhs_batch->children[c]->length > lhs_batch->length) lkids[c].length = lhs_batch->length;
        if (rhs_batch->offset || rhs_batch->children[c]->length > rhs_batch->length) rkids[c].length = rhs_batch->length;
        lp[c] = &lkids[c];
        rp[c] = &rkids[c];
        lsp[c] = lhs_schema->children[c];
        rsp[c] = rhs_schema->children[c];
        names[c] = lhs_schema->children[c]->name;
        ArrowArray* oa = new (std::nothrow) ArrowArray();
        ArrowSchema* os = new (std::nothrow) ArrowSchema();
        if (oa) memset(oa, 0, sizeof(*oa));
        if (os) memset(os, 0, sizeof(*os));
        oom = oom || !oa || !os;
        h->children.push_back(oa);
        sh->children.push_back(os);
    }
    ma_status st = MA_OK;
    if (oom) {
        set_error("out of host memory");
        st = MA_ERR_DEVICE;
    } else {
        st = export_columns(ctx, op, n_cols, lp.data(), lsp.data(), rp.data(), rsp.data(), names.data(), h->children.data(),
                            sh->children.data());
    }
    if (st != MA_OK) {  // nothing was produced: the child structs are empty shells (release == NULL)
        ArrowArray tmp{};
        tmp.release = release_array;
        tmp.private_data = h;
        release_array(&tmp);
        ArrowSchema tmps{};
        tmps.release = release_schema;
        tmps.private_data = sh;
        release_schema(&tmps);
        return st;
    }
    memset(out_batch, 0, sizeof(*out_batch));
    out_batch->length = n_cols ? h->children[0]->length : lhs_batch->length;
    out_batch->null_count = 0;
    out_batch->offset = 0;
    out_batch->n_buffers = 1;  // a struct array has one (validity) buffer; none is attached
    h->buffers[0] = nullptr;
    out_batch->buffers = h->buffers;
    out_batch->n_children = (int64_t)n_cols;
    out_batch->children = h->children.empty() ? nullptr : h->children.data();
    out_batch->dictionary = nullptr;
    out_batch->release = release_array;
    out_batch->private_data = h;
    fill_schema(out_schema, "+s", lhs_schema->name, false, sh);
    return MA_OK;
}

}  // extern "C"
