// SuperTable (op) SuperTable as a streaming operator over the Arrow C Stream Interface.
//
// The reference computes it batch by batch — broadcast_super_table_with_operator,
// src/kernels/broadcast/super_table.rs:37-72: chunk counts must match ("SuperTable chunk count mismatch: {} vs {}"),
// batch i of the result = broadcast_table_with_operator(op, lhs_i, rhs_i) — and moves SuperTables across its FFI as
// record-batch streams (src/ffi/arrow_c_ffi.rs:2104-2260, one struct array per batch). Here the two are joined: the
// operator CONSUMES two ArrowArrayStreams and IS an ArrowArrayStream; each get_next pulls one batch from either side,
// runs every column pair on the GPU (ma_apply_arrow_batch_export) and hands back an owned struct array in pinned
// memory. Nothing is materialised beyond the batch in flight, so a table larger than HBM streams through.
//
// Small batches (round 3). A SuperTable rechunked at RechunkStrategy::Auto travels as 8192-row batches, and a result slab,
// two to six staged copies, the kernels and a synchronisation PER BATCH cost 145-366 us (1-2.7 GB/s of input + output). Batch
// pairs under kGatherBytes per column are therefore GATHERED: their columns are copied into pinned tiles (validity appended
// at bit granularity), a full tile — up to kGatherRows rows — is ONE call of the batch operator (kernels reading and writing
// pinned memory in place), and the result batches are handed out as SLICES of the tile's result (Arrow `offset`, one shared
// owner). Batch boundaries, names, types, errors and their positions are those of the batch-by-batch form: should the tile
// call fail (a dense integer division by zero somewhere), the held input batches are replayed one by one.
#include <atomic>
#include <cerrno>
#include <deque>
#include <new>
#include <string>
#include <vector>

#include "ma_common.hpp"

using namespace ma;

namespace {

constexpr size_t kGatherBytes = (size_t)1 << 20;  // per column and batch: below this a batch pair is gathered
constexpr size_t kGatherRows = (size_t)1 << 20;   // rows per tile: 8 MiB per 8-byte column

struct SideTile {  // the gathered columns of one side: pinned blocks (ma_alloc64_pinned: recycled)
    std::vector<char*> values;
    std::vector<uint64_t*> bits;
    std::vector<char> masked;
    std::vector<size_t> elem;
};

struct OpStream {
    ma_ctx* ctx = nullptr;
    int32_t op = 0;
    ArrowArrayStream lhs{}, rhs{};    // moved in: released with the operator
    ArrowSchema lhs_schema{}, rhs_schema{};
    bool have_schemas = false;
    uint64_t batches = 0;             // result batches produced so far (queued ones included)
    std::string last_error;
    // gathering
    int gather_ok = -1;               // -1: not decided; 0: some column is not a plain numeric primitive; 1: yes
    SideTile lt, rt;
    size_t tile_rows = 0;
    std::vector<size_t> tile_lens;    // rows of each gathered batch pair
    std::vector<std::vector<char>> tile_validity;  // per gathered pair and column: did either side carry validity?
    std::vector<ArrowArray> held_l, held_r;  // the gathered input batches, kept until their tile has been computed
    std::deque<ArrowArray> ready;     // result batches not yet handed out
    bool inputs_done = false;
    int pending_rc = 0;               // an error to report once `ready` has drained
    std::string pending_error;
};

// A tile's result (one struct array over all gathered rows) and the slices handed out from it.
struct TileResult {
    ArrowArray whole{};
    std::atomic<long> refs{0};
};
void drop_tile_ref(TileResult* tr) {
    if (tr->refs.fetch_sub(1, std::memory_order_acq_rel) == 1) {
        if (tr->whole.release) tr->whole.release(&tr->whole);
        delete tr;
    }
}
// What a slice child owns: a reference on the tile's result and ITS OWN buffers array — the Arrow C Data Interface lets a
// consumer move a child out of its parent and release the parent first, so nothing a child points to may live in the parent.
struct SliceChild {
    TileResult* tr = nullptr;
    const void* buffers[2] = {nullptr, nullptr};
};
struct SliceHolder {
    TileResult* tr = nullptr;
    std::vector<ArrowArray> kids;
    std::vector<ArrowArray*> kid_ptrs;
    const void* struct_buffers[1] = {nullptr};
};
void release_slice_child(ArrowArray* a) {
    if (!a || !a->release) return;
    SliceChild* sc = (SliceChild*)a->private_data;
    TileResult* tr = sc->tr;
    delete sc;
    a->release = nullptr;
    drop_tile_ref(tr);
}
void release_slice(ArrowArray* a) {
    if (!a || !a->release) return;
    SliceHolder* h = (SliceHolder*)a->private_data;
    for (ArrowArray& k : h->kids) release_slice_child(&k);  // children the consumer did not move out
    TileResult* tr = h->tr;
    delete h;
    a->release = nullptr;
    drop_tile_ref(tr);
}
// Rows [row0, row0 + n) of the tile result as an owned struct array (children share the tile's buffers through `offset`).
// had_validity[c]: whether either input column of THIS batch pair carried validity — when neither did the slice's child has
// no validity buffer and null_count 0, exactly what the batch-by-batch form (and the reference's mask union,
// src/kernels/broadcast/super_array.rs:224) gives, even though another batch of the same tile made the tile's column nullable.
bool make_slice(TileResult* tr, size_t row0, size_t n, const std::vector<char>& had_validity, ArrowArray* out) {
    SliceHolder* h = new (std::nothrow) SliceHolder();
    if (!h) return false;
    const size_t nc = (size_t)tr->whole.n_children;
    h->tr = tr;
    h->kids.resize(nc);
    h->kid_ptrs.resize(nc);
    for (size_t c = 0; c < nc; ++c) {
        const ArrowArray* w = tr->whole.children[c];
        ArrowArray& k = h->kids[c];
        memset(&k, 0, sizeof(k));
        SliceChild* sc = new (std::nothrow) SliceChild();
        if (!sc) {
            for (size_t j = 0; j < c; ++j) release_slice_child(&h->kids[j]);
            delete h;
            return false;
        }
        sc->tr = tr;
        const bool nullable = w->buffers[0] != nullptr && (c >= had_validity.size() || had_validity[c]);
        sc->buffers[0] = nullable ? w->buffers[0] : nullptr;
        sc->buffers[1] = w->buffers[1];
        k.length = (int64_t)n;
        k.null_count = nullable ? -1 : 0;
        k.offset = w->offset + (int64_t)row0;
        k.n_buffers = 2;
        k.buffers = sc->buffers;
        k.private_data = sc;
        k.release = release_slice_child;
        tr->refs.fetch_add(1, std::memory_order_relaxed);
        h->kid_ptrs[c] = &k;
    }
    memset(out, 0, sizeof(*out));
    out->length = (int64_t)n;
    out->null_count = 0;
    out->n_buffers = 1;
    out->buffers = h->struct_buffers;
    out->n_children = (int64_t)nc;
    out->children = h->kid_ptrs.data();
    out->private_data = h;
    out->release = release_slice;
    tr->refs.fetch_add(1, std::memory_order_relaxed);
    return true;
}

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

size_t plain_elem(const ArrowSchema* c) {
    const char* f = c ? c->format : nullptr;
    if (!f || f[0] == 0 || f[1] != 0 || c->dictionary || !strchr("iIlLfg", f[0])) return 0;
    return (f[0] == 'l' || f[0] == 'L' || f[0] == 'g') ? 8 : 4;
}

void decide_gather(OpStream* o) {
    if (o->gather_ok >= 0) return;
    o->gather_ok = 0;
    const ArrowSchema &ls = o->lhs_schema, &rs = o->rhs_schema;
    if (!ls.format || !rs.format || strcmp(ls.format, "+s") != 0 || strcmp(rs.format, "+s") != 0) return;
    if (ls.n_children != rs.n_children || ls.n_children <= 0) return;
    const size_t nc = (size_t)ls.n_children;
    for (SideTile* t : {&o->lt, &o->rt}) {
        t->values.assign(nc, nullptr);
        t->bits.assign(nc, nullptr);
        t->masked.assign(nc, 0);
        t->elem.assign(nc, 0);
    }
    for (size_t c = 0; c < nc; ++c) {
        o->lt.elem[c] = plain_elem(ls.children[c]);
        o->rt.elem[c] = plain_elem(rs.children[c]);
        if (!o->lt.elem[c] || !o->rt.elem[c]) return;
    }
    o->gather_ok = 1;
}

void free_tiles(OpStream* o) {
    for (SideTile* t : {&o->lt, &o->rt}) {
        for (char* p : t->values) (void)ma_free_pinned(p);
        for (uint64_t* p : t->bits) (void)ma_free_pinned(p);
        t->values.clear();
        t->bits.clear();
    }
}

// Both sides' next batches. *end: both streams ended. Returns 0 or an errno with o->last_error set.
int pull_pair(OpStream* o, ArrowArray* l, ArrowArray* r, bool* end) {
    *end = false;
    memset(l, 0, sizeof(*l));
    memset(r, 0, sizeof(*r));
    if (o->lhs.get_next(&o->lhs, l) != 0) {
        const char* e = o->lhs.get_last_error ? o->lhs.get_last_error(&o->lhs) : nullptr;
        return fail(o, EIO, std::string("lhs stream get_next failed: ") + (e ? e : "(no message)"));
    }
    if (o->rhs.get_next(&o->rhs, r) != 0) {
        const char* e = o->rhs.get_last_error ? o->rhs.get_last_error(&o->rhs) : nullptr;
        if (l->release) l->release(l);
        return fail(o, EIO, std::string("rhs stream get_next failed: ") + (e ? e : "(no message)"));
    }
    const bool l_end = l->release == nullptr, r_end = r->release == nullptr;
    if (l_end && r_end) {
        *end = true;
        return 0;
    }
    if (l_end != r_end) {  // super_table.rs:46-55
        if (l->release) l->release(l);
        if (r->release) r->release(r);
        const uint64_t seen = o->batches + o->tile_lens.size();  // pairs read so far (the gathered ones included)
        return fail(o, EINVAL, "SuperTable chunk count mismatch: " + std::to_string(seen + (l_end ? 0 : 1)) + " vs " +
                                   std::to_string(seen + (r_end ? 0 : 1)) + " (one stream ended first)");
    }
    return 0;
}

// One batch pair through the batch operator (consumes both inputs). Returns 0 or an errno with o->last_error set.
int compute_pair(OpStream* o, ArrowArray* l, ArrowArray* r, ArrowArray* out) {
    ArrowSchema result_schema{};
    ma_status st = ma_apply_arrow_batch_export(o->ctx, o->op, l, &o->lhs_schema, r, &o->rhs_schema, out, &result_schema);
    std::string msg = st == MA_OK ? "" : ma_last_error_string();
    l->release(l);  // the inputs of this batch are consumed (the result owns its own pinned buffers)
    r->release(r);
    if (st != MA_OK) {
        memset(out, 0, sizeof(*out));
        return fail(o, (st == MA_ERR_UNSUPPORTED || st == MA_ERR_LENGTH_MISMATCH || st == MA_ERR_INVALID_ARGUMENT) ? EINVAL : EIO,
                    "batch " + std::to_string(o->batches) + ": " + msg);
    }
    release_schema_if(&result_schema);
    ++o->batches;
    return 0;
}

// A batch pair the gather can take: regular struct arrays of equal, small row counts with plain two-buffer children.
bool small_pair(const OpStream* o, const ArrowArray* l, const ArrowArray* r) {
    const int64_t nc = o->lhs_schema.n_children;
    if (l->length != r->length || l->length < 0) return false;
    for (const ArrowArray* b : {l, r}) {
        if (b->n_children != nc || !b->children || b->offset < 0) return false;
        for (int64_t c = 0; c < nc; ++c) {
            const ArrowArray* k = b->children[c];
            if (!k || k->n_buffers != 2 || !k->buffers || k->offset < 0 || k->length < b->offset + b->length) return false;
            if (b->length > 0 && !k->buffers[1]) return false;
        }
    }
    size_t widest = 4;
    for (int64_t c = 0; c < nc; ++c) widest = std::max(widest, std::max(o->lt.elem[(size_t)c], o->rt.elem[(size_t)c]));
    return (size_t)l->length * widest < kGatherBytes;
}

// May this pair share the current tile? Always — except under integer Div / Rem / FloorDiv, where the dense kernel and the
// Bitmask-gated one DISAGREE on a zero divisor: dense raises MA_ERR_DIVIDE_BY_ZERO (the reference's panic,
// src/kernels/arithmetic/std.rs:53-77), gated writes 0 and clears the row's validity bit (std.rs:95-138). A tile column is
// gated as soon as ONE of its batches carried validity, so a dense batch gathered next to a nullable one would have its zero
// divisor turned into a quiet 0 (and, its slice carrying no bitmap, a VALID 0) where the batch-by-batch form raises. Under
// those operators a tile therefore holds, per integer column, either only batches with validity or only batches without.
bool joins_tile(const OpStream* o, const ArrowArray* l, const ArrowArray* r) {
    if (o->tile_rows == 0 || o->tile_lens.empty()) return true;
    if (o->op != MA_OP_DIVIDE && o->op != MA_OP_REMAINDER && o->op != MA_OP_FLOORDIV) return true;
    const size_t nc = o->lt.values.size();
    const std::vector<char>& tile_had = o->tile_validity.front();  // uniform over the tile's batches by this very rule
    for (size_t c = 0; c < nc; ++c) {
        const char fl = o->lhs_schema.children[c]->format[0], fr = o->rhs_schema.children[c]->format[0];
        if ((fl == 'f' || fl == 'g') || (fr == 'f' || fr == 'g')) continue;  // a float result: no division error either way
        const ArrowArray *kl = l->children[c], *kr = r->children[c];
        const bool had = (kl->null_count != 0 && kl->buffers[0]) || (kr->null_count != 0 && kr->buffers[0]);
        if (had != (tile_had[c] != 0)) return false;
    }
    return true;
}

ma_status ensure_tiles(OpStream* o) {
    for (SideTile* t : {&o->lt, &o->rt})
        for (size_t c = 0; c < t->values.size(); ++c) {
            if (t->values[c]) continue;
            void* p = nullptr;
            MA_TRY(ma_alloc64_pinned(kGatherRows * t->elem[c] + 64, &p));
            t->values[c] = (char*)p;
            MA_TRY(ma_alloc64_pinned((kGatherRows / 64 + 2) * 8, &p));
            t->bits[c] = (uint64_t*)p;
        }
    return MA_OK;
}

// Copies the pair's columns behind the rows gathered so far and keeps the batches (released once their tile is computed).
void gather_pair(OpStream* o, ArrowArray* l, ArrowArray* r) {
    const size_t n = (size_t)l->length, at = o->tile_rows;
    const ArrowArray* sides[2] = {l, r};
    SideTile* tiles[2] = {&o->lt, &o->rt};
    std::vector<char> had(o->lt.values.size(), 0);
    for (int sd = 0; sd < 2; ++sd) {
        const ArrowArray* b = sides[sd];
        SideTile* t = tiles[sd];
        for (size_t c = 0; c < t->values.size(); ++c) {
            const ArrowArray* k = b->children[c];
            const size_t off = (size_t)k->offset + (size_t)b->offset;  // a struct's offset shifts every child
            if (n) memcpy(t->values[c] + at * t->elem[c], (const char*)k->buffers[1] + off * t->elem[c], n * t->elem[c]);
            const uint8_t* validity = k->null_count == 0 ? nullptr : (const uint8_t*)k->buffers[0];
            if (validity) had[c] = 1;
            if (validity && !t->masked[c]) {  // the column's first batch with nulls in this tile: the rows so far are valid
                memset(t->bits[c], 0, (kGatherRows / 64 + 2) * 8);
                append_bits(t->bits[c], 0, nullptr, 0, 0, at);
                t->masked[c] = 1;
            }
            if (t->masked[c]) append_bits(t->bits[c], at, validity, (off + n + 7) >> 3, off, n);
        }
    }
    o->held_l.push_back(*l);
    o->held_r.push_back(*r);
    o->tile_lens.push_back(n);
    o->tile_validity.push_back(std::move(had));
    o->tile_rows += n;
}

// The gathered rows through the batch operator in ONE call; the result batches (slices of its result) go to o->ready. If
// the call fails the held batches are replayed one by one, so that the error carries its batch number and every batch in
// front of it is still delivered.
void process_tile(OpStream* o) {
    const size_t nb = o->tile_lens.size();
    if (nb == 0) return;
    const size_t nc = o->lt.values.size();
    std::vector<ArrowArray> kids(2 * nc);
    std::vector<ArrowArray*> lp(nc), rp(nc);
    std::vector<const void*> bufs(4 * nc);
    SideTile* tiles[2] = {&o->lt, &o->rt};
    for (int sd = 0; sd < 2; ++sd)
        for (size_t c = 0; c < nc; ++c) {
            ArrowArray& k = kids[(size_t)sd * nc + c];
            memset(&k, 0, sizeof(k));
            const void** kb = &bufs[((size_t)sd * nc + c) * 2];
            kb[0] = tiles[sd]->masked[c] ? tiles[sd]->bits[c] : nullptr;
            kb[1] = tiles[sd]->values[c];
            k.length = (int64_t)o->tile_rows;
            k.null_count = tiles[sd]->masked[c] ? -1 : 0;
            k.n_buffers = 2;
            k.buffers = kb;
            (sd ? rp : lp)[c] = &k;
        }
    const void* no_buffer[1] = {nullptr};
    ArrowArray lt{}, rt{};
    for (ArrowArray* b : {&lt, &rt}) {
        b->length = (int64_t)o->tile_rows;
        b->n_buffers = 1;
        b->buffers = no_buffer;
        b->n_children = (int64_t)nc;
    }
    lt.children = lp.data();
    rt.children = rp.data();
    TileResult* tr = new (std::nothrow) TileResult();
    ArrowSchema result_schema{};
    ma_status st = tr ? ma_apply_arrow_batch_export(o->ctx, o->op, &lt, &o->lhs_schema, &rt, &o->rhs_schema, &tr->whole, &result_schema)
                      : MA_ERR_DEVICE;
    bool sliced = false;
    if (st == MA_OK) {
        release_schema_if(&result_schema);
        tr->refs.store(1, std::memory_order_relaxed);  // this frame
        sliced = true;
        size_t row0 = 0;
        std::vector<ArrowArray> outs(nb);
        for (size_t k = 0; k < nb && sliced; ++k) {
            sliced = make_slice(tr, row0, o->tile_lens[k], o->tile_validity[k], &outs[k]);
            row0 += o->tile_lens[k];
            if (!sliced)
                for (size_t j = 0; j < k; ++j) outs[j].release(&outs[j]);
        }
        if (sliced)
            for (size_t k = 0; k < nb; ++k) {
                o->ready.push_back(outs[k]);
                ++o->batches;
            }
        drop_tile_ref(tr);
    } else {
        delete tr;
    }
    if (sliced) {
        for (size_t k = 0; k < nb; ++k) {
            o->held_l[k].release(&o->held_l[k]);
            o->held_r[k].release(&o->held_r[k]);
        }
    } else {  // replay batch by batch (the inputs are still held)
        size_t k = 0;
        for (; k < nb; ++k) {
            ArrowArray out{};
            if (int rc = compute_pair(o, &o->held_l[k], &o->held_r[k], &out)) {
                o->pending_rc = rc;
                o->pending_error = o->last_error;
                ++k;
                break;
            }
            o->ready.push_back(out);
        }
        for (; k < nb; ++k) {
            o->held_l[k].release(&o->held_l[k]);
            o->held_r[k].release(&o->held_r[k]);
        }
    }
    o->held_l.clear();
    o->held_r.clear();
    o->tile_lens.clear();
    o->tile_validity.clear();
    o->tile_rows = 0;
    std::fill(o->lt.masked.begin(), o->lt.masked.end(), 0);
    std::fill(o->rt.masked.begin(), o->rt.masked.end(), 0);
}

int op_get_next(ArrowArrayStream* self, ArrowArray* out) {
    OpStream* o = (OpStream*)self->private_data;
    memset(out, 0, sizeof(*out));
    if (int rc = fetch_schemas(o)) return rc;
    decide_gather(o);
    for (;;) {
        if (!o->ready.empty()) {
            *out = o->ready.front();
            o->ready.pop_front();
            return 0;
        }
        if (o->pending_rc) {  // an error met while batches in front of it were still queued
            const int rc = o->pending_rc;
            o->pending_rc = 0;
            o->last_error = o->pending_error;
            return rc;
        }
        if (o->inputs_done) return 0;  // end of stream: out->release == NULL
        while (o->ready.empty() && !o->pending_rc && !o->inputs_done) {
            ArrowArray l{}, r{};
            bool end = false;
            if (int rc = pull_pair(o, &l, &r, &end)) {  // what was gathered is delivered first, then the error
                const std::string msg = o->last_error;
                process_tile(o);
                if (!o->pending_rc) {
                    o->pending_rc = rc;
                    o->pending_error = msg;
                }
                break;
            }
            if (end) {
                process_tile(o);
                o->inputs_done = true;
                break;
            }
            if (o->gather_ok == 1 && small_pair(o, &l, &r) && ensure_tiles(o) == MA_OK) {
                if (o->tile_rows + (size_t)l.length > kGatherRows || !joins_tile(o, &l, &r)) process_tile(o);
                if (o->pending_rc) {  // the tile's replay met an error: this pair is not computed (as batch by batch)
                    l.release(&l);
                    r.release(&r);
                    break;
                }
                gather_pair(o, &l, &r);
                continue;
            }
            process_tile(o);  // a large or irregular pair: the gathered ones go first
            if (o->pending_rc) {
                l.release(&l);
                r.release(&r);
                break;
            }
            ArrowArray res{};
            if (int rc = compute_pair(o, &l, &r, &res)) {
                o->pending_rc = rc;
                o->pending_error = o->last_error;
            } else {
                o->ready.push_back(res);
            }
        }
    }
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
        for (ArrowArray& a : o->ready)
            if (a.release) a.release(&a);
        for (ArrowArray& a : o->held_l)
            if (a.release) a.release(&a);
        for (ArrowArray& a : o->held_r)
            if (a.release) a.release(&a);
        free_tiles(o);
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
