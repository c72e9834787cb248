// Single-process, multi-GPU row-chunk reduction: what the reference's Rayon path
// (`slice.par_chunks(1 << 20).map(simd_sum).sum()`, benches/benchmark_parallel_simd.rs:81-98) becomes for a host
// that drives all GPUs of a node from one process (e.g. the Rust library itself). One ma_ctx per device; every
// device scans its own row chunk concurrently (enqueue-only launches on independent streams) and writes
// {sum | hi, lo, count} into its 64-byte record of the reduction's column. The records are then exchanged:
//
//   RCCL  (MA_GROUP_EXCHANGE_RCCL)  ncclCommInitAll over the group's devices; ONE grouped ncclAllGather of the
//         members' record blocks over xGMI, enqueued on the members' streams, followed on every device by the
//         member-ordered fold kernel (wrapping adds; error-free two-sum for the double-double pairs): every GPU ends
//         up holding bit-identical finals — an all-reduce whose f64 result stays within 1 ULP, which ncclAllReduce's
//         own rounding sum cannot give. Nothing but the finals ever reaches the host.
//   host  (default)  the kernels write their records straight into pinned host memory and the host folds the
//         G x 64 bytes after the streams drain — no collective is needed inside one process.
//
// The multi-PROCESS form of the same exchange is ma_comm_sum_exchange (ma_rccl.hip).
#include <cstdlib>
#include <vector>

#include "ma_rccl.hpp"

struct ma_group {
    std::vector<ma_ctx*> ctxs;
    std::mutex mu;
    bool use_rccl = false;
    std::vector<ncclComm_t> comms;
    // RCCL: per member a device block of kColumns records (`local`), a device block of G x kColumns gathered records
    // and a pinned host block of kColumns x 4 finals the fold kernel writes. host: `local[i]` points into `host_records`.
    std::vector<uint64_t*> local, gathered, finals;
    uint64_t* host_records = nullptr;  // pinned, G x kColumns records (host exchange)
    uint64_t* host_finals = nullptr;   // pinned (RCCL: G x kColumns x 4) or plain (host: kColumns x 4) finals
    // ma_group_consolidate_column: per destination member a grow-only device arena the chunks' validity bytes are
    // gathered into before the bit-granular join (re-used across calls in stream order)
    std::vector<void*> mask_stage;
    std::vector<size_t> mask_stage_bytes;
    char note[256] = "";
};

using namespace ma;

namespace {

constexpr int kColumns = MA_GROUP_MAX_COLUMNS;
constexpr size_t kBlockWords = (size_t)kColumns * kRecordWords;

struct HostFoldDD {
    uint64_t isum = 0, icnt = 0, fcnt = 0;
    double hi = 0.0, lo = 0.0;
    void add(const uint64_t* p) {
        isum += p[0];
        icnt += p[1];
        double h, l;
        memcpy(&h, &p[2], 8);
        memcpy(&l, &p[3], 8);
        const double t = hi + h;
        const double bp = t - hi;
        const double e = (hi - (t - bp)) + (h - bp);
        hi = t;
        lo += e + l;
        fcnt += p[4];
    }
    double total() const {
        const bool finite = (hi - hi == 0.0) && (lo - lo == 0.0);
        return finite ? hi + lo : hi;
    }
};

// Which device a pointer's memory is resident on (-1: pageable / pinned host memory, or NULL). A chunked column's
// thousands of pointers run through a handful of allocations: each allocation is asked about once per call.
class DeviceLookup {
  public:
    int device_of(const void* p) {
        const uintptr_t a = (uintptr_t)p;
        for (const Seen& r : seen_)
            if (a >= r.lo && a < r.hi) return r.device;
        hipPointerAttribute_t attr;
        if (p == nullptr || hipPointerGetAttributes(&attr, p) != hipSuccess) {
            (void)hipGetLastError();
            return -1;
        }
        if (attr.type != hipMemoryTypeDevice) return -1;
        hipDeviceptr_t base = nullptr;
        size_t size = 0;
        if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)p) == hipSuccess && size)
            seen_.push_back({(uintptr_t)base, (uintptr_t)base + size, attr.device});
        else
            (void)hipGetLastError();
        return attr.device;
    }

  private:
    struct Seen {
        uintptr_t lo, hi;
        int device;
    };
    std::vector<Seen> seen_;
};

// Frees the exchange's buffers and communicators (either kind); the members stay.
void release_exchange(ma_group* g) {
    for (size_t i = 0; i < g->ctxs.size(); ++i) {
        (void)hipSetDevice(g->ctxs[i]->device);
        (void)hipStreamSynchronize(g->ctxs[i]->stream);
    }
    if (g->use_rccl || !g->comms.empty()) {
        const RcclApi* api = rccl();
        for (ncclComm_t c : g->comms)
            if (c && api) (void)api->CommDestroy(c);
        for (size_t i = 0; i < g->ctxs.size(); ++i) {
            (void)hipSetDevice(g->ctxs[i]->device);
            if (i < g->local.size() && g->local[i]) (void)hipFree(g->local[i]);
            if (i < g->gathered.size() && g->gathered[i]) (void)hipFree(g->gathered[i]);
        }
        if (g->host_finals) (void)hipHostFree(g->host_finals);
    } else {
        if (g->host_records) (void)hipHostFree(g->host_records);
        free(g->host_finals);
    }
    g->host_records = g->host_finals = nullptr;
    g->comms.clear();
    g->local.clear();
    g->gathered.clear();
    g->finals.clear();
    g->use_rccl = false;
}

void destroy_members(ma_group* g) {
    release_exchange(g);  // drains every member's stream first
    for (size_t i = 0; i < g->mask_stage.size() && i < g->ctxs.size(); ++i)
        if (g->mask_stage[i]) {
            (void)hipSetDevice(g->ctxs[i]->device);
            (void)hipFree(g->mask_stage[i]);
        }
    g->mask_stage.clear();
    g->mask_stage_bytes.clear();
    for (ma_ctx* c : g->ctxs) ma_ctx_destroy(c);
    g->ctxs.clear();
}

// RCCL exchange set-up. Returns MA_OK with g->use_rccl set, or a status + the thread's error string.
ma_status setup_rccl(ma_group* g) {
    const size_t n = g->ctxs.size();
    for (size_t i = 0; i < n; ++i)
        for (size_t j = i + 1; j < n; ++j)
            MA_REQUIRE(g->ctxs[i]->device != g->ctxs[j]->device, MA_ERR_UNSUPPORTED,
                       "an RCCL communicator needs distinct devices (members %zu and %zu share device %d)", i, j,
                       g->ctxs[i]->device);
    const RcclApi* api = rccl();
    if (!api) return MA_ERR_UNSUPPORTED;
    std::vector<int> devs(n);
    for (size_t i = 0; i < n; ++i) devs[i] = g->ctxs[i]->device;
    g->comms.assign(n, nullptr);  // non-empty from here on: release_exchange frees the RCCL-side resources
    MA_NCCL(api, CommInitAll(g->comms.data(), (int)n, devs.data()));
    g->use_rccl = true;
    g->local.assign(n, nullptr);
    g->gathered.assign(n, nullptr);
    g->finals.assign(n, nullptr);
    MA_HIP(hipHostMalloc((void**)&g->host_finals, n * kColumns * 4 * 8, hipHostMallocPortable | hipHostMallocMapped));
    memset(g->host_finals, 0, n * kColumns * 4 * 8);
    for (size_t i = 0; i < n; ++i) {
        MA_HIP(hipSetDevice(devs[i]));
        MA_HIP(device_malloc(devs[i], (void**)&g->local[i], kBlockWords * 8));
        MA_HIP(device_malloc(devs[i], (void**)&g->gathered[i], n * kBlockWords * 8));
        MA_HIP(hipMemset(g->local[i], 0, kBlockWords * 8));
        MA_HIP(hipMemset(g->gathered[i], 0, n * kBlockWords * 8));
        g->finals[i] = g->host_finals + i * kColumns * 4;
    }
    return MA_OK;
}

ma_status setup_host(ma_group* g) {
    const size_t n = g->ctxs.size();
    MA_HIP(hipHostMalloc((void**)&g->host_records, n * kBlockWords * 8, hipHostMallocPortable | hipHostMallocMapped));
    memset(g->host_records, 0, n * kBlockWords * 8);
    g->host_finals = (uint64_t*)calloc((size_t)kColumns * 4, 8);
    MA_REQUIRE(g->host_finals != nullptr, MA_ERR_DEVICE, "out of host memory");
    g->local.assign(n, nullptr);
    for (size_t i = 0; i < n; ++i) g->local[i] = g->host_records + i * kBlockWords;
    return MA_OK;
}

template <typename Launch>
ma_status enqueue_members(ma_group* g, int32_t column, Launch launch) {
    MA_REQUIRE(g != nullptr, MA_ERR_INVALID_ARGUMENT, "group is NULL");
    MA_REQUIRE(column >= 0 && column < kColumns, MA_ERR_INVALID_ARGUMENT, "column %d out of range [0,%d)", column, kColumns);
    std::lock_guard<std::mutex> lock(g->mu);
    for (size_t i = 0; i < g->ctxs.size(); ++i) {
        uint64_t* rec = g->local[i] + (size_t)column * kRecordWords;
        MA_TRY(launch(i, rec));  // enqueue only (the members are in async mode): all devices run concurrently
    }
    return MA_OK;
}

ma_status exchange_locked(ma_group* g) {
    if (!g->use_rccl) return MA_OK;  // host exchange: the records are already in host memory once the streams drain
    const RcclApi* api = rccl();
    if (!api) return MA_ERR_UNSUPPORTED;
    const size_t n = g->ctxs.size();
    MA_NCCL(api, GroupStart());
    for (size_t i = 0; i < n; ++i) {
        ncclResult_t r = api->AllGather(g->local[i], g->gathered[i], kBlockWords * 8, ncclChar, g->comms[i], g->ctxs[i]->stream);
        if (r != ncclSuccess) {
            (void)api->GroupEnd();
            return rccl_fail(r, "AllGather", __FILE__, __LINE__);
        }
    }
    MA_NCCL(api, GroupEnd());
    for (size_t i = 0; i < n; ++i) {
        ma_ctx* c = g->ctxs[i];
        std::lock_guard<std::mutex> lock(c->mu);
        MA_HIP(hipSetDevice(c->device));
        MA_TRY(enqueue_fold_columns(c, g->gathered[i], n, kBlockWords, kColumns, g->finals[i]));
    }
    return MA_OK;
}

ma_status synchronize_locked(ma_group* g) {
    ma_status st = MA_OK;
    for (ma_ctx* c : g->ctxs) {
        ma_status s = ma_ctx_synchronize(c);
        if (st == MA_OK) st = s;
    }
    MA_TRY(st);
    if (!g->use_rccl) {
        for (int col = 0; col < kColumns; ++col) {
            HostFoldDD f;
            for (size_t i = 0; i < g->ctxs.size(); ++i) f.add(g->local[i] + (size_t)col * kRecordWords);
            uint64_t* out = g->host_finals + (size_t)col * 4;
            const double total = f.total();
            out[0] = f.isum;
            out[1] = f.icnt;
            memcpy(&out[2], &total, 8);
            out[3] = f.fcnt;
        }
    }
    return MA_OK;
}

const uint64_t* finals_of(const ma_group* g, size_t member, int32_t column) {
    const uint64_t* base = g->use_rccl ? g->finals[member] : g->host_finals;
    return base + (size_t)column * 4;
}

}  // namespace

extern "C" {

ma_status ma_group_create_ex(const int32_t* device_ordinals, int32_t n_members, uint32_t flags, ma_group** out_group) {
    MA_REQUIRE(out_group != nullptr, MA_ERR_INVALID_ARGUMENT, "out_group is NULL");
    *out_group = nullptr;
    MA_REQUIRE(device_ordinals != nullptr && n_members > 0 && n_members <= 1024, MA_ERR_INVALID_ARGUMENT,
               "a group needs 1..1024 members");
    ma_group* g = new ma_group();
    for (int32_t i = 0; i < n_members; ++i) {
        ma_ctx* c = nullptr;
        ma_status st = ma_ctx_create(device_ordinals[i], &c);
        if (st == MA_OK) st = ma_ctx_set_async(c, 1);  // members only ever enqueue; the group synchronises them
        if (st != MA_OK) {
            if (c) ma_ctx_destroy(c);
            destroy_members(g);
            delete g;
            return st;
        }
        g->ctxs.push_back(c);
    }
    ma_status st = MA_OK;
    if (flags & MA_GROUP_EXCHANGE_RCCL) {
        st = setup_rccl(g);
        if (st != MA_OK && (flags & MA_GROUP_EXCHANGE_FALLBACK_HOST)) {
            snprintf(g->note, sizeof(g->note), "host fold instead of RCCL: %s", ma_last_error_string());
            release_exchange(g);  // whatever the attempt allocated; the members stay
            st = setup_host(g);
        }
    } else {
        st = setup_host(g);
    }
    if (st != MA_OK) {
        destroy_members(g);
        delete g;
        return st;
    }
    *out_group = g;
    return MA_OK;
}

ma_status ma_group_create(const int32_t* device_ordinals, int32_t n_members, ma_group** out_group) {
    // MINARROW_HIP_GROUP_EXCHANGE=rccl makes RCCL the default exchange (with the host fold as its fallback).
    const char* env = getenv("MINARROW_HIP_GROUP_EXCHANGE");
    const uint32_t flags = (env && strcmp(env, "rccl") == 0) ? (MA_GROUP_EXCHANGE_RCCL | MA_GROUP_EXCHANGE_FALLBACK_HOST) : 0u;
    return ma_group_create_ex(device_ordinals, n_members, flags, out_group);
}

void ma_group_destroy(ma_group* group) {
    if (!group) return;
    destroy_members(group);
    delete group;
}

int32_t ma_group_size(ma_group* group) { return group ? (int32_t)group->ctxs.size() : 0; }

ma_ctx* ma_group_ctx(ma_group* group, int32_t index) {
    if (!group || index < 0 || (size_t)index >= group->ctxs.size()) return nullptr;
    return group->ctxs[(size_t)index];
}

int32_t ma_group_exchange_kind(ma_group* group) { return group && group->use_rccl ? 1 : 0; }
const char* ma_group_exchange_note(ma_group* group) { return group ? group->note : ""; }

ma_status ma_group_enqueue_sum_i64(ma_group* group, int32_t column, const int64_t* const* chunk_data,
                                   const size_t* chunk_lens, const uint8_t* const* chunk_masks,
                                   const size_t* chunk_mask_offsets) {
    MA_REQUIRE(group && chunk_data && chunk_lens, MA_ERR_INVALID_ARGUMENT, "NULL argument");
    return enqueue_members(group, column, [&](size_t i, uint64_t* rec) {
        return ma_i64_sum(group->ctxs[i], chunk_data[i], chunk_lens[i], chunk_masks ? chunk_masks[i] : nullptr,
                          chunk_mask_offsets ? chunk_mask_offsets[i] : 0, -1, (int64_t*)&rec[0], &rec[1]);
    });
}

ma_status ma_group_enqueue_sum_f64(ma_group* group, int32_t column, const double* const* chunk_data,
                                   const size_t* chunk_lens, const uint8_t* const* chunk_masks,
                                   const size_t* chunk_mask_offsets) {
    MA_REQUIRE(group && chunk_data && chunk_lens, MA_ERR_INVALID_ARGUMENT, "NULL argument");
    return enqueue_members(group, column, [&](size_t i, uint64_t* rec) {
        return ma_f64_sum_dd(group->ctxs[i], chunk_data[i], chunk_lens[i], chunk_masks ? chunk_masks[i] : nullptr,
                             chunk_mask_offsets ? chunk_mask_offsets[i] : 0, -1, (double*)&rec[2], (double*)&rec[3], &rec[4]);
    });
}

// route_super_array_broadcast over the GPUs of a group — src/kernels/broadcast/super_array.rs:180-251, whose chunk loop
// is sequential ("// TODO: Parallelise", :193). Chunk pair i belongs to member i % G: the pairs of one member run as ONE
// launch on its device (ma_route_super_array_broadcast), all members concurrently; no bytes cross between GPUs and the
// result stays chunked where its inputs are ("output stays sharded").
ma_status ma_group_route_super_array_broadcast(ma_group* group, int32_t format_code, int32_t op, size_t n_chunks,
                                               const void* const* lhs_data, const size_t* lhs_lens,
                                               const uint8_t* const* lhs_masks, const void* const* rhs_data,
                                               const size_t* rhs_lens, const uint8_t* const* rhs_masks,
                                               const uint8_t* const* member_mask_overrides, void* const* out_data,
                                               uint8_t* const* out_masks, int32_t* out_has_mask) {
    MA_REQUIRE(group != nullptr, MA_ERR_INVALID_ARGUMENT, "group is NULL");
    MA_REQUIRE(n_chunks == 0 || (lhs_data && lhs_lens && rhs_data && rhs_lens && out_data), MA_ERR_INVALID_ARGUMENT,
               "NULL chunk table");
    for (size_t i = 0; i < n_chunks; ++i)  // the whole SuperArray is checked before any member starts
        if (lhs_lens[i] != rhs_lens[i]) {
            set_error("Super Array broadcasting error - Chunk %zu: LHS %zu RHS %zu", i, lhs_lens[i], rhs_lens[i]);
            return MA_ERR_LENGTH_MISMATCH;
        }
    std::lock_guard<std::mutex> lock(group->mu);
    const size_t G = group->ctxs.size();
    // A device-resident chunk must live on its member's GPU: the kernels address it directly. A SuperArray's thousands
    // of chunk pointers run through a handful of allocations: each allocation is asked about once.
    DeviceLookup lookup;
    auto device_of = [&](const void* p) { return lookup.device_of(p); };
    for (size_t i = 0; i < n_chunks; ++i) {
        if (lhs_lens[i] == 0) continue;
        const int want = group->ctxs[i % G]->device;
        const void* ptrs[3] = {lhs_data[i], rhs_data[i], out_data[i]};
        for (const void* p : ptrs) {
            const int dev = device_of(p);
            MA_REQUIRE(dev < 0 || dev == want, MA_ERR_INVALID_ARGUMENT,
                       "chunk %zu belongs to member %zu (device %d) but one of its buffers is resident on device %d", i, i % G,
                       want, dev);
        }
    }
    std::vector<const void*> l, r;
    std::vector<void*> o;
    std::vector<const uint8_t*> lm, rm;
    std::vector<uint8_t*> om;
    std::vector<size_t> ll, rl;
    std::vector<int32_t> has;
    for (size_t m = 0; m < G && m < n_chunks; ++m) {
        l.clear(); r.clear(); o.clear(); lm.clear(); rm.clear(); om.clear(); ll.clear(); rl.clear();
        for (size_t i = m; i < n_chunks; i += G) {
            l.push_back(lhs_data[i]);
            r.push_back(rhs_data[i]);
            o.push_back(out_data[i]);
            ll.push_back(lhs_lens[i]);
            rl.push_back(rhs_lens[i]);
            lm.push_back(lhs_masks ? lhs_masks[i] : nullptr);
            rm.push_back(rhs_masks ? rhs_masks[i] : nullptr);
            om.push_back(out_masks ? out_masks[i] : nullptr);
        }
        has.assign(l.size(), 0);
        MA_TRY(ma_route_super_array_broadcast(group->ctxs[m], format_code, op, l.size(), l.data(), ll.data(), lm.data(), r.data(),
                                              rl.data(), rm.data(), member_mask_overrides ? member_mask_overrides[m] : nullptr,
                                              o.data(), om.data(), has.data()));
        if (out_has_mask)
            for (size_t j = 0; j < has.size(); ++j) out_has_mask[m + j * G] = has[j];
    }
    return MA_OK;
}

// SuperTable::consolidate for a column whose batches live on different GPUs (src/structs/chunked/super_table.rs:657-743,
// src/traits/consolidate.rs:80-207): chunk i is resident on member i % G, the consolidated column lands on
// `dest_member`'s device. Every owner pushes its chunks into place with peer copies on ITS stream (all xGMI links into
// the destination run concurrently); validity bytes are gathered into a staging arena on the destination and joined at
// bit granularity there (ma_consolidate_boolean_column: chunks without a bitmap contribute all-valid bits). The
// destination's stream ends up ordered behind every copy; owners wait for the destination's earlier work before they
// overwrite `out_data`. Enqueue-only. SURVEY.md 8(e): "do it only when a contiguous result is explicitly requested" —
// the per-column reduce of a batch-sharded table needs no bytes moved (ma_group_enqueue_sum_*).
ma_status ma_group_consolidate_column(ma_group* group, int32_t dest_member, size_t elem_size, size_t n_chunks,
                                      const void* const* chunk_data, const size_t* chunk_lens,
                                      const uint8_t* const* chunk_masks, const size_t* chunk_mask_offsets, void* out_data,
                                      uint8_t* out_mask, int32_t* out_has_mask) {
    MA_REQUIRE(group != nullptr, MA_ERR_INVALID_ARGUMENT, "group is NULL");
    MA_REQUIRE(elem_size == 1 || elem_size == 2 || elem_size == 4 || elem_size == 8, MA_ERR_UNSUPPORTED,
               "element size %zu is not supported (1, 2, 4 or 8 bytes)", elem_size);
    MA_REQUIRE(n_chunks > 0, MA_ERR_INVALID_ARGUMENT, "consolidate() called on empty SuperTable");
    MA_REQUIRE(chunk_data != nullptr && chunk_lens != nullptr, MA_ERR_INVALID_ARGUMENT, "NULL chunk table");
    std::lock_guard<std::mutex> lock(group->mu);
    const size_t G = group->ctxs.size();
    MA_REQUIRE(dest_member >= 0 && (size_t)dest_member < G, MA_ERR_INVALID_ARGUMENT, "member %d out of range", dest_member);
    ma_ctx* dest = group->ctxs[(size_t)dest_member];
    bool has_mask = false;
    size_t total = 0;
    for (size_t i = 0; i < n_chunks; ++i) {
        MA_REQUIRE(chunk_lens[i] == 0 || chunk_data[i] != nullptr, MA_ERR_INVALID_ARGUMENT, "chunk %zu data is NULL", i);
        if (chunk_masks && chunk_masks[i] && chunk_lens[i]) has_mask = true;
        total += chunk_lens[i];
    }
    if (out_has_mask) *out_has_mask = has_mask ? 1 : 0;
    if (total == 0) return MA_OK;
    MA_REQUIRE(out_data != nullptr, MA_ERR_INVALID_ARGUMENT, "out_data is NULL");
    MA_REQUIRE(!has_mask || out_mask != nullptr, MA_ERR_INVALID_ARGUMENT, "a chunk carries nulls but out_mask is NULL");
    MA_REQUIRE(!has_mask || ((uintptr_t)out_mask & 7) == 0, MA_ERR_INVALID_ARGUMENT,
               "output bitmap must be 8-byte aligned (got %p)", (const void*)out_mask);
    DeviceLookup lookup;
    auto device_of = [&](const void* p) { return lookup.device_of(p); };
    MA_REQUIRE(device_of(out_data) == dest->device, MA_ERR_INVALID_ARGUMENT,
               "out_data must be device memory of member %d (device %d)", dest_member, dest->device);
    MA_REQUIRE(!has_mask || device_of(out_mask) == dest->device, MA_ERR_INVALID_ARGUMENT,
               "out_mask must be device memory of member %d (device %d)", dest_member, dest->device);
    for (size_t i = 0; i < n_chunks; ++i) {
        if (!chunk_lens[i]) continue;
        const int want = group->ctxs[i % G]->device;
        MA_REQUIRE(device_of(chunk_data[i]) == want, MA_ERR_INVALID_ARGUMENT,
                   "chunk %zu belongs to member %zu: its data must be device memory of device %d", i, i % G, want);
        MA_REQUIRE(!(chunk_masks && chunk_masks[i]) || device_of(chunk_masks[i]) == want, MA_ERR_INVALID_ARGUMENT,
                   "chunk %zu belongs to member %zu: its bitmap must be device memory of device %d", i, i % G, want);
    }
    // validity staging arena on the destination: one 8-byte aligned slot per chunk
    std::vector<size_t> slot_off(n_chunks, 0), slot_bit(n_chunks, 0);
    size_t stage_bytes = 0;
    if (has_mask) {
        for (size_t i = 0; i < n_chunks; ++i) {
            const size_t mo = (chunk_masks && chunk_masks[i] && chunk_mask_offsets) ? chunk_mask_offsets[i] : 0;
            slot_bit[i] = mo & 7;
            slot_off[i] = stage_bytes;
            stage_bytes += (((slot_bit[i] + chunk_lens[i] + 7) >> 3) + 15) & ~(size_t)7;  // bytes + a spare word
        }
        if (group->mask_stage.size() < G) {
            group->mask_stage.resize(G, nullptr);
            group->mask_stage_bytes.resize(G, 0);
        }
        if (stage_bytes > group->mask_stage_bytes[(size_t)dest_member]) {
            MA_HIP(hipSetDevice(dest->device));
            if (group->mask_stage[(size_t)dest_member]) {
                MA_HIP(hipStreamSynchronize(dest->stream));  // an earlier join may still read the old arena
                MA_HIP(hipFree(group->mask_stage[(size_t)dest_member]));
                group->mask_stage[(size_t)dest_member] = nullptr;
                group->mask_stage_bytes[(size_t)dest_member] = 0;
            }
            const size_t want = stage_bytes + stage_bytes / 2 + 4096;
            MA_HIP(device_malloc(dest->device, &group->mask_stage[(size_t)dest_member], want));
            group->mask_stage_bytes[(size_t)dest_member] = want;
        }
    }
    uint8_t* stage = has_mask ? (uint8_t*)group->mask_stage[(size_t)dest_member] : nullptr;

    // Owners start after the destination's earlier work (which may still use out_data or the arena).
    MA_HIP(hipSetDevice(dest->device));
    hipEvent_t ready = nullptr;
    MA_HIP(hipEventCreateWithFlags(&ready, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ready, dest->stream);
    std::vector<hipEvent_t> done(G, nullptr);
    size_t row = 0;
    for (size_t i = 0; i < n_chunks && e == hipSuccess; ++i) {
        const size_t n = chunk_lens[i];
        if (n) {
            ma_ctx* owner = group->ctxs[i % G];
            e = hipSetDevice(owner->device);
            if (e == hipSuccess && !done[i % G]) {  // first use of this owner in the call
                e = hipEventCreateWithFlags(&done[i % G], hipEventDisableTiming);
                if (e == hipSuccess && owner != dest) e = hipStreamWaitEvent(owner->stream, ready, 0);
            }
            if (e == hipSuccess)
                e = hipMemcpyPeerAsync((char*)out_data + row * elem_size, dest->device, chunk_data[i], owner->device, n * elem_size,
                                       owner->stream);
            if (e == hipSuccess && has_mask) {
                uint8_t* slot = stage + slot_off[i];
                const size_t nbytes = (slot_bit[i] + n + 7) >> 3;
                if (chunk_masks && chunk_masks[i]) {
                    const size_t mo = chunk_mask_offsets ? chunk_mask_offsets[i] : 0;
                    e = hipMemcpyPeerAsync(slot, dest->device, chunk_masks[i] + (mo >> 3), owner->device, nbytes, owner->stream);
                } else {  // no bitmap: all rows valid (consolidate.rs:80-105); the arena is the destination's to fill
                    e = hipSetDevice(dest->device);
                    if (e == hipSuccess) e = hipMemsetAsync(slot, 0xFF, nbytes, dest->stream);
                }
            }
        }
        row += n;
    }
    // The destination continues once every owner's copies have landed.
    for (size_t m = 0; m < G && e == hipSuccess; ++m) {
        if (!done[m]) continue;
        e = hipSetDevice(group->ctxs[m]->device);
        if (e == hipSuccess) e = hipEventRecord(done[m], group->ctxs[m]->stream);
        if (e == hipSuccess && group->ctxs[m] != dest) {
            e = hipSetDevice(dest->device);
            if (e == hipSuccess) e = hipStreamWaitEvent(dest->stream, done[m], 0);
        }
    }
    for (hipEvent_t ev : done)
        if (ev) (void)hipEventDestroy(ev);  // released once the recorded work has completed
    (void)hipEventDestroy(ready);
    if (e != hipSuccess) return hip_fail(e, "ma_group_consolidate_column", __FILE__, __LINE__);
    if (!has_mask) return MA_OK;
    // bit-granular join of the staged validity windows on the destination
    std::vector<const uint8_t*> bits(n_chunks);
    for (size_t i = 0; i < n_chunks; ++i) bits[i] = stage + slot_off[i];
    MA_HIP(hipSetDevice(dest->device));
    return ma_consolidate_boolean_column(dest, n_chunks, bits.data(), slot_bit.data(), chunk_lens, nullptr, nullptr, out_mask,
                                         nullptr, nullptr);
}

ma_status ma_group_exchange(ma_group* group) {
    MA_REQUIRE(group != nullptr, MA_ERR_INVALID_ARGUMENT, "group is NULL");
    std::lock_guard<std::mutex> lock(group->mu);
    return exchange_locked(group);
}

ma_status ma_group_synchronize(ma_group* group) {
    MA_REQUIRE(group != nullptr, MA_ERR_INVALID_ARGUMENT, "group is NULL");
    std::lock_guard<std::mutex> lock(group->mu);
    return synchronize_locked(group);
}

ma_status ma_group_member_result(ma_group* group, int32_t member, int32_t column, int64_t* out_int_sum,
                                 uint64_t* out_int_count, double* out_f64_sum, uint64_t* out_f64_count) {
    MA_REQUIRE(group != nullptr, MA_ERR_INVALID_ARGUMENT, "group is NULL");
    MA_REQUIRE(member >= 0 && (size_t)member < group->ctxs.size(), MA_ERR_INVALID_ARGUMENT, "member %d out of range", member);
    MA_REQUIRE(column >= 0 && column < kColumns, MA_ERR_INVALID_ARGUMENT, "column %d out of range [0,%d)", column, kColumns);
    std::lock_guard<std::mutex> lock(group->mu);
    const uint64_t* f = finals_of(group, (size_t)member, column);
    if (out_int_sum) *out_int_sum = (int64_t)f[0];
    if (out_int_count) *out_int_count = f[1];
    if (out_f64_sum) memcpy(out_f64_sum, &f[2], 8);
    if (out_f64_count) *out_f64_count = f[3];
    return MA_OK;
}

ma_status ma_group_result(ma_group* group, int32_t column, int64_t* out_int_sum, uint64_t* out_int_count,
                          double* out_f64_sum, uint64_t* out_f64_count) {
    return ma_group_member_result(group, 0, column, out_int_sum, out_int_count, out_f64_sum, out_f64_count);
}

ma_status ma_group_sum_i64(ma_group* group, const int64_t* const* chunk_data, const size_t* chunk_lens,
                           const uint8_t* const* chunk_masks, const size_t* chunk_mask_offsets, int64_t* out_sum,
                           uint64_t* out_valid_count) {
    MA_TRY(ma_group_enqueue_sum_i64(group, 0, chunk_data, chunk_lens, chunk_masks, chunk_mask_offsets));
    MA_TRY(ma_group_exchange(group));
    MA_TRY(ma_group_synchronize(group));
    return ma_group_result(group, 0, out_sum, out_valid_count, nullptr, nullptr);
}

ma_status ma_group_sum_f64(ma_group* group, const double* const* chunk_data, const size_t* chunk_lens,
                           const uint8_t* const* chunk_masks, const size_t* chunk_mask_offsets, double* out_sum,
                           uint64_t* out_valid_count) {
    MA_TRY(ma_group_enqueue_sum_f64(group, 0, chunk_data, chunk_lens, chunk_masks, chunk_mask_offsets));
    MA_TRY(ma_group_exchange(group));
    MA_TRY(ma_group_synchronize(group));
    return ma_group_result(group, 0, nullptr, nullptr, out_sum, out_valid_count);
}

}  // extern "C"
