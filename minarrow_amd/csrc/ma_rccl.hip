// RCCL loader and the multi-PROCESS communicator of the C ABI (ma_comm_*): one process per GPU, the per-rank
// reduction records {sum | hi, lo, count} exchanged with one all-gather over xGMI and folded in rank order on every
// rank. Replaces the combine step of the reference's Rayon reduction (`par_chunks(1 << 20).map(simd_sum).sum()`,
// benches/benchmark_parallel_simd.rs:81-98) for a row-chunk partition over the GPUs of a node. The single-process
// form (one host thread driving every GPU) is ma_group_* (ma_group.hip).
#include <dlfcn.h>

#include <mutex>

#include "ma_rccl.hpp"

namespace ma {

namespace {
std::once_flag g_rccl_once;
RcclApi g_rccl;
bool g_rccl_ok = false;
char g_rccl_err[256] = "";

void load_rccl() {
    // RCCL must sit on the SAME HIP runtime as this library: its communicators take our streams and device pointers. A
    // process can hold two runtimes — a torch-free host of this library that imports torch.distributed for its gloo
    // rendezvous has /opt/rocm's (ours, loaded first) and PyTorch's bundled one — and a bare dlopen("librccl.so.1") would
    // then resolve to whichever RCCL was loaded first under that SONAME, i.e. PyTorch's, bound to the OTHER runtime
    // (ncclCommInitRank fails with a device error). So: first the RCCL that lives next to the libamdhip64 we are linked
    // against (dladdr of a HIP entry point), by path; only then by name.
    static char beside[2][512];
    Dl_info info;
    if (dladdr((void*)&hipGetDeviceCount, &info) && info.dli_fname) {
        const char* slash = strrchr(info.dli_fname, '/');
        if (slash) {
            const int dir = (int)(slash - info.dli_fname);
            snprintf(beside[0], sizeof(beside[0]), "%.*s/librccl.so.1", dir, info.dli_fname);
            snprintf(beside[1], sizeof(beside[1]), "%.*s/librccl.so", dir, info.dli_fname);
        }
    }
    const char* kNames[] = {beside[0], beside[1], "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* name : kNames) {
        if (!name[0]) continue;
        h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (h) {
            g_rccl.path = name;
            break;
        }
    }
    if (!h) {
        snprintf(g_rccl_err, sizeof(g_rccl_err), "cannot open librccl.so.1: %s", dlerror());
        return;
    }
    bool ok = true;
    auto sym = [&](const char* name) -> void* {
        void* p = dlsym(h, name);
        if (!p) {
            ok = false;
            snprintf(g_rccl_err, sizeof(g_rccl_err), "librccl has no symbol %s", name);
        }
        return p;
    };
    g_rccl.GetVersion = (decltype(g_rccl.GetVersion))sym("ncclGetVersion");
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
    g_rccl.CommInitAll = (decltype(g_rccl.CommInitAll))sym("ncclCommInitAll");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
    g_rccl.CommCount = (decltype(g_rccl.CommCount))sym("ncclCommCount");
    g_rccl.CommAbort = (decltype(g_rccl.CommAbort))sym("ncclCommAbort");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))sym("ncclAllGather");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))sym("ncclAllReduce");
    g_rccl.GroupStart = (decltype(g_rccl.GroupStart))sym("ncclGroupStart");
    g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))sym("ncclGroupEnd");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
    g_rccl_ok = ok;
}
}  // namespace

const RcclApi* rccl() {
    std::call_once(g_rccl_once, load_rccl);
    if (!g_rccl_ok) {
        set_error("RCCL is not available: %s", g_rccl_err);
        return nullptr;
    }
    return &g_rccl;
}

ma_status rccl_fail(ncclResult_t r, const char* what, const char* file, int line) {
    const RcclApi* api = g_rccl_ok ? &g_rccl : nullptr;
    set_error("RCCL error %d (%s) in %s at %s:%d", (int)r, api ? api->GetErrorString(r) : "?", what, file, line);
    (void)hipGetLastError();
    return MA_ERR_DEVICE;
}

}  // namespace ma

using namespace ma;

struct ma_comm {
    ma_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, n_ranks = 1;
    // Overlapped exchanges (ma_comm_sum_exchange_overlapped): an internal context with its own stream — the all-gather and
    // the fold of record set `slot` run there while the context's stream already scans the next step — and per slot one
    // event each way (no timing, device-scope release: the dependency never leaves the GPU).
    ma_ctx* side = nullptr;
    hipEvent_t ready[2] = {nullptr, nullptr}, done[2] = {nullptr, nullptr};
    bool used[2] = {false, false};
    bool no_wait_value = false;  // hipStreamWaitValue64 failed once: overlapped exchanges are ordered by events from then on
    ma::ExchangeTimer timer;  // every 4th exchange: all-gather / fold durations (ma_comm_exchange_stats)
};

namespace ma {
ma_status make_lane(ma_ctx* root, ma_ctx** out);  // ma_ctx.hip: an internal context of root's device, own stream + scratch
}

extern "C" {

const char* ma_rccl_path(void) {
    const RcclApi* api = rccl();
    return api ? api->path : "";
}

int32_t ma_rccl_version(void) {
    const RcclApi* api = rccl();
    int v = 0;
    if (!api || api->GetVersion(&v) != ncclSuccess) return 0;
    return v;
}

ma_status ma_comm_unique_id(uint8_t* out_id) {
    MA_REQUIRE(out_id != nullptr, MA_ERR_INVALID_ARGUMENT, "out_id is NULL");
    static_assert(MA_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "ma_comm ids carry an ncclUniqueId");
    const RcclApi* api = rccl();
    if (!api) return MA_ERR_UNSUPPORTED;
    ncclUniqueId id;
    MA_NCCL(api, GetUniqueId(&id));
    memcpy(out_id, id.internal, MA_COMM_ID_BYTES);
    return MA_OK;
}

ma_status ma_comm_create(ma_ctx* ctx, const uint8_t* id, int32_t rank, int32_t n_ranks, ma_comm** out_comm) {
    MA_REQUIRE(out_comm != nullptr, MA_ERR_INVALID_ARGUMENT, "out_comm is NULL");
    *out_comm = nullptr;
    MA_REQUIRE(ctx != nullptr && id != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx or id is NULL");
    MA_REQUIRE(n_ranks >= 1 && rank >= 0 && rank < n_ranks, MA_ERR_INVALID_ARGUMENT, "rank %d of %d", rank, n_ranks);
    const RcclApi* api = rccl();
    if (!api) return MA_ERR_UNSUPPORTED;
    MA_ENTER_PRIMARY(ctx);
    MA_NO_CAPTURE(ctx, "ma_comm_create");
    MA_HIP(hipSetDevice(ctx->device));
    ncclUniqueId uid;
    memcpy(uid.internal, id, MA_COMM_ID_BYTES);
    ncclComm_t comm = nullptr;
    MA_NCCL(api, CommInitRank(&comm, n_ranks, uid, rank));
    ma_comm* c = new ma_comm();
    c->ctx = ctx;
    c->comm = comm;
    c->rank = rank;
    c->n_ranks = n_ranks;
    *out_comm = c;
    return MA_OK;
}

void ma_comm_destroy(ma_comm* comm) {
    if (!comm) return;
    (void)hipSetDevice(comm->ctx->device);
    if (comm->side) {
        (void)hipStreamSynchronize(comm->side->stream);
        for (int k = 0; k < 2; ++k) {
            if (comm->ready[k]) (void)hipEventDestroy(comm->ready[k]);
            if (comm->done[k]) (void)hipEventDestroy(comm->done[k]);
        }
    }
    if (comm->comm) {
        (void)hipStreamSynchronize(comm->ctx->stream);
        const RcclApi* api = rccl();
        if (api) (void)api->CommDestroy(comm->comm);
    }
    if (comm->side) ma_ctx_destroy(comm->side);
    comm->timer.destroy();
    delete comm;
}

int32_t ma_comm_rank(ma_comm* comm) { return comm ? comm->rank : -1; }
int32_t ma_comm_size(ma_comm* comm) { return comm ? comm->n_ranks : 0; }

ma_status ma_comm_all_gather(ma_comm* comm, const void* send, void* recv, size_t bytes_per_rank) {
    MA_REQUIRE(comm != nullptr, MA_ERR_INVALID_ARGUMENT, "comm is NULL");
    if (bytes_per_rank == 0) return MA_OK;
    MA_REQUIRE(send != nullptr && recv != nullptr, MA_ERR_INVALID_ARGUMENT, "NULL buffer");
    MA_REQUIRE(pointer_kind(send) != kPageable && pointer_kind(recv) != kPageable, MA_ERR_INVALID_ARGUMENT,
               "collectives need device-reachable buffers");
    const RcclApi* api = rccl();
    if (!api) return MA_ERR_UNSUPPORTED;
    ma_ctx* ctx = comm->ctx;
    MA_ENTER_PRIMARY(ctx);  // collectives are tied to the context's own stream
    MA_NO_CAPTURE(ctx, "a collective");
    MA_HIP(hipSetDevice(ctx->device));
    MA_NCCL(api, AllGather(send, recv, bytes_per_rank, ncclChar, comm->comm, ctx->stream));
    if (!is_async(ctx)) MA_HIP(hipStreamSynchronize(ctx->stream));
    return MA_OK;
}

ma_status ma_comm_all_reduce_sum_i64(ma_comm* comm, const int64_t* send, int64_t* recv, size_t count) {
    MA_REQUIRE(comm != nullptr, MA_ERR_INVALID_ARGUMENT, "comm is NULL");
    if (count == 0) return MA_OK;
    MA_REQUIRE(send != nullptr && recv != nullptr, MA_ERR_INVALID_ARGUMENT, "NULL buffer");
    MA_REQUIRE(pointer_kind(send) != kPageable && pointer_kind(recv) != kPageable, MA_ERR_INVALID_ARGUMENT,
               "collectives need device-reachable buffers");
    const RcclApi* api = rccl();
    if (!api) return MA_ERR_UNSUPPORTED;
    ma_ctx* ctx = comm->ctx;
    MA_ENTER_PRIMARY(ctx);  // collectives are tied to the context's own stream
    MA_NO_CAPTURE(ctx, "a collective");
    MA_HIP(hipSetDevice(ctx->device));
    MA_NCCL(api, AllReduce(send, recv, count, ncclInt64, ncclSum, comm->comm, ctx->stream));  // wrapping, like the scans
    if (!is_async(ctx)) MA_HIP(hipStreamSynchronize(ctx->stream));
    return MA_OK;
}

ma_status ma_comm_sum_exchange(ma_comm* comm, const uint64_t* local_records, size_t slots_per_rank, size_t n_columns,
                               uint64_t* gathered, uint64_t* out_finals) {
    MA_REQUIRE(comm != nullptr, MA_ERR_INVALID_ARGUMENT, "comm is NULL");
    MA_REQUIRE(local_records && gathered && out_finals, MA_ERR_INVALID_ARGUMENT, "NULL buffer");
    MA_REQUIRE(slots_per_rank >= 1 && n_columns >= 1, MA_ERR_INVALID_ARGUMENT, "nothing to exchange");
    MA_REQUIRE(pointer_kind(local_records) != kPageable && pointer_kind(gathered) != kPageable &&
                   pointer_kind(out_finals) != kPageable,
               MA_ERR_INVALID_ARGUMENT, "collectives need device-reachable buffers");
    const RcclApi* api = rccl();
    if (!api) return MA_ERR_UNSUPPORTED;
    ma_ctx* ctx = comm->ctx;
    MA_ENTER_PRIMARY(ctx);  // collectives are tied to the context's own stream
    MA_NO_CAPTURE(ctx, "a collective");
    MA_HIP(hipSetDevice(ctx->device));
    const size_t per_rank_words = slots_per_rank * n_columns * kRecordWords;
    const int tk = comm->timer.begin(ctx->stream);
    MA_NCCL(api, AllGather(local_records, gathered, per_rank_words * 8, ncclChar, comm->comm, ctx->stream));
    comm->timer.mark(tk, 1, ctx->stream);
    // column c is folded over (rank, slot) in that order: records c, c + n_columns, ...
    MA_TRY(enqueue_fold_columns(ctx, gathered, (size_t)comm->n_ranks * slots_per_rank, n_columns * kRecordWords, n_columns,
                                out_finals));
    comm->timer.mark(tk, 2, ctx->stream);
    if (!is_async(ctx)) MA_HIP(hipStreamSynchronize(ctx->stream));
    return MA_OK;
}

static ma_status exchange_overlapped(ma_comm* comm, int32_t slot, uint64_t* stamp, uint64_t stamp_value, const uint64_t* local_records,
                                     size_t slots_per_rank, size_t n_columns, uint64_t* gathered, uint64_t* out_finals);

ma_status ma_comm_sum_exchange_overlapped(ma_comm* comm, int32_t slot, const uint64_t* local_records, size_t slots_per_rank,
                                          size_t n_columns, uint64_t* gathered, uint64_t* out_finals) {
    return exchange_overlapped(comm, slot, nullptr, 0, local_records, slots_per_rank, n_columns, gathered, out_finals);
}

ma_status ma_comm_sum_exchange_overlapped_on_stamp(ma_comm* comm, int32_t slot, uint64_t* stamp, uint64_t stamp_value,
                                                   const uint64_t* local_records, size_t slots_per_rank, size_t n_columns,
                                                   uint64_t* gathered, uint64_t* out_finals) {
    MA_REQUIRE(stamp != nullptr, MA_ERR_INVALID_ARGUMENT, "stamp is NULL");
    return exchange_overlapped(comm, slot, stamp, stamp_value, local_records, slots_per_rank, n_columns, gathered, out_finals);
}

static ma_status exchange_overlapped(ma_comm* comm, int32_t slot, uint64_t* stamp, uint64_t stamp_value, const uint64_t* local_records,
                                     size_t slots_per_rank, size_t n_columns, uint64_t* gathered, uint64_t* out_finals) {
    MA_REQUIRE(comm != nullptr, MA_ERR_INVALID_ARGUMENT, "comm is NULL");
    MA_REQUIRE(slot == 0 || slot == 1, MA_ERR_INVALID_ARGUMENT, "slot must be 0 or 1 (two record sets in flight)");
    MA_REQUIRE(local_records && gathered && out_finals, MA_ERR_INVALID_ARGUMENT, "NULL buffer");
    MA_REQUIRE(slots_per_rank >= 1 && n_columns >= 1, MA_ERR_INVALID_ARGUMENT, "nothing to exchange");
    MA_REQUIRE(pointer_kind(local_records) != kPageable && pointer_kind(gathered) != kPageable &&
                   pointer_kind(out_finals) != kPageable,
               MA_ERR_INVALID_ARGUMENT, "collectives need device-reachable buffers");
    const RcclApi* api = rccl();
    if (!api) return MA_ERR_UNSUPPORTED;
    ma_ctx* ctx = comm->ctx;
    MA_ENTER_PRIMARY(ctx);
    MA_NO_CAPTURE(ctx, "a collective");
    MA_HIP(hipSetDevice(ctx->device));
    if (!comm->side) {
        MA_TRY(make_lane(ctx, &comm->side));
        for (int k = 0; k < 2; ++k) {
            MA_HIP(hipEventCreateWithFlags(&comm->ready[k], hipEventDisableTiming | hipEventReleaseToDevice));
            MA_HIP(hipEventCreateWithFlags(&comm->done[k], hipEventDisableTiming | hipEventReleaseToDevice));
        }
    }
    ma_ctx* side = comm->side;
    std::lock_guard<std::mutex> lock(side->mu);
    bool waited = false;
    if (stamp && !comm->no_wait_value) {
        // ... behind the stamp the slot's last scan stores after its results: nothing at all goes onto the context's stream
        if (hipStreamWaitValue64(side->stream, stamp, stamp_value, hipStreamWaitValueGte, ~(uint64_t)0) == hipSuccess) {
            waited = true;
        } else {  // a runtime without stream memory operations: the event below orders the same thing (the stamped launch is
            (void)hipGetLastError();  // already on the context's stream), from now on without trying again
            comm->no_wait_value = true;
        }
    }
    if (!waited) {
        // behind everything the context's stream has been given so far (the scans that wrote this slot's records) ...
        MA_HIP(hipEventRecord(comm->ready[slot], ctx->stream));
        MA_HIP(hipStreamWaitEvent(side->stream, comm->ready[slot], 0));
    }
    const size_t per_rank_words = slots_per_rank * n_columns * kRecordWords;
    const int tk = comm->timer.begin(side->stream);
    MA_NCCL(api, AllGather(local_records, gathered, per_rank_words * 8, ncclChar, comm->comm, side->stream));
    comm->timer.mark(tk, 1, side->stream);
    MA_TRY(enqueue_fold_columns(side, gathered, (size_t)comm->n_ranks * slots_per_rank, n_columns * kRecordWords, n_columns,
                                out_finals));
    comm->timer.mark(tk, 2, side->stream);
    // ... and ma_comm_slot_wait(slot) puts the context's stream behind it
    MA_HIP(hipEventRecord(comm->done[slot], side->stream));
    comm->used[slot] = true;
    return MA_OK;
}

ma_status ma_comm_slot_wait(ma_comm* comm, int32_t slot) {
    MA_REQUIRE(comm != nullptr, MA_ERR_INVALID_ARGUMENT, "comm is NULL");
    MA_REQUIRE(slot == 0 || slot == 1, MA_ERR_INVALID_ARGUMENT, "slot must be 0 or 1");
    if (!comm->side || !comm->used[slot]) return MA_OK;
    ma_ctx* ctx = comm->ctx;
    MA_ENTER_PRIMARY(ctx);
    MA_HIP(hipSetDevice(ctx->device));
    // Two steps ago, normally long finished: then the host's look at the event is the whole wait, and the stream is spared a
    // barrier packet per step (each costs it a few us: the in-order queue stalls on it even when the event has fired).
    const hipError_t q = hipEventQuery(comm->done[slot]);
    if (q == hipSuccess) return MA_OK;
    if (q != hipErrorNotReady) {
        (void)hipGetLastError();
        return hip_fail(q, "hipEventQuery(exchange done)", __FILE__, __LINE__);
    }
    (void)hipGetLastError();
    MA_HIP(hipStreamWaitEvent(ctx->stream, comm->done[slot], 0));
    return MA_OK;
}

ma_status ma_comm_exchange_stats(ma_comm* comm, double* out_all_gather_us, double* out_fold_us, int32_t* out_samples,
                                 int32_t* out_rccl_ranks) {
    MA_REQUIRE(comm != nullptr, MA_ERR_INVALID_ARGUMENT, "comm is NULL");
    ma_ctx* ctx = comm->ctx;
    MA_ENTER_PRIMARY(ctx);
    MA_HIP(hipSetDevice(ctx->device));
    if (comm->side) {
        std::lock_guard<std::mutex> lock(comm->side->mu);
        comm->timer.report(out_all_gather_us, out_fold_us, out_samples);
    } else {
        comm->timer.report(out_all_gather_us, out_fold_us, out_samples);
    }
    if (out_rccl_ranks) {
        int n = 0;
        const RcclApi* api = rccl();
        if (!api || !api->CommCount || api->CommCount(comm->comm, &n) != ncclSuccess) n = 0;
        *out_rccl_ranks = n;
    }
    return MA_OK;
}

ma_status ma_comm_synchronize(ma_comm* comm) {
    MA_REQUIRE(comm != nullptr, MA_ERR_INVALID_ARGUMENT, "comm is NULL");
    if (comm->side) {
        MA_HIP(hipSetDevice(comm->ctx->device));
        MA_HIP(hipStreamSynchronize(comm->side->stream));
    }
    return ma_ctx_synchronize(comm->ctx);
}

}  // extern "C"
