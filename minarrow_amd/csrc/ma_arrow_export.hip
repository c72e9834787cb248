// Results handed back through the Arrow C Data Interface — the producer side of the boundary.
//
// The reference exports with create_arrow_export (src/ffi/arrow_c_ffi.rs:1742-1821): a heap ArrowArray whose
// `private_data` is a Holder keeping the buffers alive, `offset` 0, `n_buffers` 2 for primitives, `null_count` 0
// when there is no validity buffer, 64-byte aligned buffers (check_alignment, :1722-1738) and `release` callbacks
// that drop the Holder (:193-262). The structs produced here follow the same contract; the Holder owns pinned host
// allocations (ma_alloc64_pinned — page aligned, device-mapped) that the kernels wrote directly, and `release`
// returns them with ma_free_pinned. Record batches are struct arrays ("+s") with one child per column, the shape
// the reference's record-batch stream yields (arrow_c_ffi.rs:1823-1834, 2104-2260).
#include <new>
#include <string>

#include "ma_common.hpp"

using namespace ma;

namespace {

struct ArrayHolder {
    void* values = nullptr;    // pinned
    void* validity = nullptr;  // pinned, or nullptr
    const void* buffers[2] = {nullptr, nullptr};
    std::vector<ArrowArray*> children;  // struct arrays only (each heap-allocated, released with the parent)
};

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
        (void)ma_free_pinned(h->values);
        (void)ma_free_pinned(h->validity);
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

// Computes lhs (op) rhs into freshly allocated pinned buffers and wraps them in an owned ArrowArray / ArrowSchema.
ma_status export_one(ma_ctx* ctx, int32_t op, const ArrowArray* lhs, const ArrowSchema* ls, const ArrowArray* rhs,
                     const ArrowSchema* rs, const char* name, ArrowArray* out, ArrowSchema* out_schema) {
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
    const size_t n = nl == nr ? nl : (nl == 1 ? nr : nl);
    ArrayHolder* h = new (std::nothrow) ArrayHolder();
    SchemaHolder* sh = new (std::nothrow) SchemaHolder();
    ma_status st = (h && sh) ? MA_OK : MA_ERR_DEVICE;
    if (st == MA_OK) st = ma_alloc64_pinned(n * format_size(fmt), &h->values);
    // The validity words are always provided; whether they become buffers[0] is decided by the routing.
    if (st == MA_OK) st = ma_alloc64_pinned(((n + 63) / 64) * 8 + 8, &h->validity);
    int32_t has_validity = 0;
    if (st == MA_OK)
        st = ma_apply_arrow(ctx, op, lhs, ls, rhs, rs, h->values, (uint8_t*)h->validity, &has_validity);
    int64_t null_count = 0;
    if (st == MA_OK && has_validity) {
        uint64_t ones = 0;
        st = ma_popcount_mask(ctx, (const uint8_t*)h->validity, 0, n, &ones);
        null_count = (int64_t)n - (int64_t)ones;
    }
    if (st != MA_OK) {
        if (h) {
            (void)ma_free_pinned(h->values);
            (void)ma_free_pinned(h->validity);
        }
        delete h;
        delete sh;
        return st;
    }
    if (!has_validity) {  // create_arrow_export: no validity buffer <=> null_count 0 (arrow_c_ffi.rs:1750)
        (void)ma_free_pinned(h->validity);
        h->validity = nullptr;
    }
    h->buffers[0] = h->validity;
    h->buffers[1] = h->values;
    memset(out, 0, sizeof(*out));
    out->length = (int64_t)n;
    out->null_count = null_count;
    out->offset = 0;  // arrow_c_ffi.rs:1773
    out->n_buffers = 2;
    out->n_children = 0;
    out->buffers = h->buffers;
    out->children = nullptr;
    out->dictionary = nullptr;
    out->release = release_array;
    out->private_data = h;
    const char f[2] = {fmt, 0};
    fill_schema(out_schema, f, name, has_validity != 0, sh);
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
    return export_one(ctx, op, lhs, lhs_schema, rhs, rhs_schema, name ? name : (lhs_schema ? lhs_schema->name : nullptr),
                      out_array, out_schema);
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
    ma_status st = MA_OK;
    for (size_t c = 0; c < n_cols && st == MA_OK; ++c) {
        // A struct's offset applies to its children on top of their own (Arrow C Data Interface).
        ArrowArray l = *lhs_batch->children[c], r = *rhs_batch->children[c];
        l.offset += lhs_batch->offset;
        r.offset += rhs_batch->offset;
        if (lhs_batch->offset || lhs_batch->children[c]->length > lhs_batch->length) l.length = lhs_batch->length;
        if (rhs_batch->offset || rhs_batch->children[c]->length > rhs_batch->length) r.length = rhs_batch->length;
        ArrowArray* oa = new (std::nothrow) ArrowArray();
        ArrowSchema* os = new (std::nothrow) ArrowSchema();
        if (!oa || !os) {
            delete oa;
            delete os;
            set_error("out of host memory");
            st = MA_ERR_DEVICE;
            break;
        }
        st = export_one(ctx, op, &l, lhs_schema->children[c], &r, rhs_schema->children[c], lhs_schema->children[c]->name,
                        oa, os);
        if (st != MA_OK) {
            delete oa;
            delete os;
            break;
        }
        h->children.push_back(oa);
        sh->children.push_back(os);
    }
    if (st != MA_OK) {
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
