// RCCL loader and the multi-PROCESS communicator of the C ABI (ma_comm_*): one process per GPU, the per-rank
// reduction records {sum | hi, lo, count} exchanged with one all-gather over xGMI and folded in rank order on every
// rank. Replaces the combine step of the reference's Rayon reduction (`par_chunks(1 << 20).map(simd_sum).sum()`,
// benches/benchmark_parallel_simd.rs:81-98) for a row-chunk partition over the GPUs of a node. The single-process
// form (one host thread driving every GPU) is ma_group_* (ma_group.hip).
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "minarrow_hip_testing.h"

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
    // MINARROW_HIP_RCCL_PATH first, and nothing else when it is set: the host named the collective library — another RCCL
    // build, or the loopback double of tests/loopback_rccl that rehearses the multi-rank paths on one GPU. A path that does
    // not open is an error, not a reason to fall back silently to the system's RCCL.
    static char named[512];
    const char* over = getenv("MINARROW_HIP_RCCL_PATH");
    if (over && over[0]) {
        snprintf(named, sizeof(named), "%s", over);
        h = dlopen(named, RTLD_NOW | RTLD_LOCAL);
        if (!h) {
            snprintf(g_rccl_err, sizeof(g_rccl_err), "cannot open MINARROW_HIP_RCCL_PATH=%s: %s", named, dlerror());
            return;
        }
        g_rccl.path = named;
        g_rccl.named = true;
    }
    for (const char* name : kNames) {
        if (h) break;
        if (!name[0]) continue;
        h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (h) g_rccl.path = name;
    }
    if (!h) {
        snprintf(g_rccl_err, sizeof(g_rccl_err), "cannot open librccl.so.1: %s", dlerror());
        return;
    }
    // only the loopback double has this symbol: it, unlike a fabric, takes several ranks on one device
    if (auto info = (const char* (*)(void))dlsym(h, "ncclLoopbackDoubleInfo")) {
        g_rccl.loopback = true;
        static char labelled[600];
        snprintf(labelled, sizeof(labelled), "REHEARSAL (loopback collective double, not RCCL): %s", g_rccl.path);
        g_rccl.path = labelled;
        (void)info;
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
    // A bounded wait ran out (or another rank's did: ma_comm_abort): the communicator was aborted. Every collective refuses
    // from then on; the object can only be destroyed. drained: every stream ran empty after the abort.
    bool broken = false, drained = true;
    // Testing hooks (ma_comm_test_*): the next exchange is held in front of its all-gather behind stall_word / one gathered word
    // is flipped in front of the fold. The stall word holds the sequence of the last release; a stall waits for the next.
    bool stall_next = false, corrupt_next = false, stall_armed = false;
    uint64_t* stall_word = nullptr;
    uint64_t stall_seq = 0, stall_release = 0;
    // The word the last overlapped-on-stamp exchange made the exchange stream wait on, and the value it waits for. The CALLER
    // owns that word: the abort path writes into it only while the library still knows it as a live stamp (ma_stamp_is_signal
    // >= 0: not freed since), and puts the awaited sequence back once the streams have run empty.
    uint64_t* last_stamp = nullptr;
    uint64_t last_stamp_value = 0;
    hipStream_t rescue = nullptr;    // nothing else is ever enqueued here: release values are written through it
};

static const char* const kCommBroken =
    "this communicator was aborted (a bounded wait ran out, here or on another rank): destroy it; the ranks may agree on a new one";

namespace ma {
ma_status make_lane(ma_ctx* root, ma_ctx** out, int cls = 0);  // ma_ctx.hip: an internal context of root's device, own stream + scratch
}

namespace {

hipStream_t comm_rescue(ma_comm* comm) {
    if (!comm->rescue) {  // the low priority class: its own hardware-queue pool, never behind a held ordinary (or high-class) stream
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) (void)hipGetLastError();
        if (hipStreamCreateWithPriority(&comm->rescue, hipStreamNonBlocking, least) != hipSuccess) {
            (void)hipGetLastError();
            comm->rescue = nullptr;
        }
    }
    return comm->rescue;
}

void comm_write_word(ma_comm* comm, uint64_t* word, const uint64_t* value) {
    if (stamp_host_store(word, *value)) return;  // host memory (signal memory is): no GPU queue involved
    hipStream_t s = comm_rescue(comm);
    if (!s) return;  // never the null stream: it would wait for the very streams that are held
    if (hipStreamWriteValue64(s, word, *value, 0) == hipSuccess) return;
    (void)hipGetLastError();
    if (hipMemcpyAsync(word, value, 8, hipMemcpyHostToDevice, s) != hipSuccess) (void)hipGetLastError();
}

// ma_comm_test_stall_next_exchange: `stream` waits for a value nobody writes until the abort path (or the communicator's
// destruction) releases it.
ma_status hook_stall(ma_comm* comm, hipStream_t stream) {
    if (!comm->stall_next) return MA_OK;
    comm->stall_next = false;
    if (!comm->stall_word) MA_TRY(stamp_alloc_kind(comm->ctx, &comm->stall_word, true));  // host-releasable
    (void)comm_rescue(comm);
    if (hipStreamWaitValue64(stream, comm->stall_word, comm->stall_seq + 1, hipStreamWaitValueGte, ~(uint64_t)0) != hipSuccess) {
        (void)hipGetLastError();
        set_error("this runtime has no stream memory operations: the stall hook cannot hold a stream");
        return MA_ERR_UNSUPPORTED;
    }
    comm->stall_armed = true;
    return MA_OK;
}

ma_status hook_corrupt(ma_comm* comm, uint64_t* gathered, hipStream_t stream) {
    if (!comm->corrupt_next) return MA_OK;
    comm->corrupt_next = false;
    static const uint64_t kFlip = 0x5A5A5A5A5A5A5A5Aull;  // rank 0's record 0, integer sum, as THIS rank gathered it
    MA_HIP(hipMemcpyAsync(gathered, &kFlip, 8, hipMemcpyHostToDevice, stream));
    return MA_OK;
}

void comm_release_waits(ma_comm* comm, bool stamps_too) {
    static const uint64_t kAll = ~(uint64_t)0;
    if (comm->stall_armed && comm->stall_word) {
        comm->stall_release = ++comm->stall_seq;
        comm_write_word(comm, comm->stall_word, &comm->stall_release);
        comm->stall_armed = false;
    }
    if (stamps_too && comm->last_stamp) {
        if (ma_stamp_is_signal(comm->last_stamp) >= 0) comm_write_word(comm, comm->last_stamp, &kAll);
        else comm->last_stamp = nullptr;  // freed by its owner since: nothing of ours can still wait on it, nothing is written
    }
}

// True when the context's stream and the exchange stream have run empty within timeout_ms; *which = the one still pending.
bool comm_wait_streams(ma_comm* comm, double timeout_ms, const char** which, hipError_t* error) {
    const auto t0 = std::chrono::steady_clock::now();
    bool main_done = false, side_done = comm->side == nullptr;
    *error = hipSuccess;
    for (;;) {
        for (int k = 0; k < 2; ++k) {
            bool& done = k == 0 ? main_done : side_done;
            if (done) continue;
            const hipError_t q = hipStreamQuery(k == 0 ? comm->ctx->stream : comm->side->stream);
            if (q == hipSuccess) {
                done = true;
            } else {
                (void)hipGetLastError();
                if (q != hipErrorNotReady) {
                    *error = q;
                    *which = k == 0 ? "the context's stream" : "the exchange stream";
                    return false;
                }
            }
        }
        if (main_done && side_done) return true;
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (us >= timeout_ms * 1e3) {
            *which = !main_done && !side_done ? "the context's stream (scans, or the wait for an earlier exchange) and the exchange stream"
                     : !main_done             ? "the context's stream (scan, all-gather or fold)"
                                              : "the exchange stream (the wait for the scan's hand-off, the all-gather or the fold)";
            return false;
        }
        // a benchmark's timed region ends in this wait: poll back to back for the first 10 ms, then sleep 1/200 of the time
        // waited so far (the overshoot stays under ~1 %)
        if (us < 10e3)
            __builtin_ia32_pause();
        else
            std::this_thread::sleep_for(std::chrono::microseconds((long)std::min(500.0, std::max(20.0, us / 200.0))));
    }
}

// Same order as the group's abort (ma_group_guard.hip): release what a stream may be held behind and give a queued
// collective a moment to start on the living communicator, abort, bounded drain.
void comm_abort(ma_comm* comm) {
    if (comm->broken) return;
    (void)hipSetDevice(comm->ctx->device);
    guard_log("rank %d abort: releasing held streams; giving queued work 200 ms", comm->rank);
    comm_release_waits(comm, true);
    const char* which = "";
    hipError_t e = hipSuccess;
    (void)comm_wait_streams(comm, 200.0, &which, &e);
    const RcclApi* api = rccl();
    guard_log("rank %d abort: ncclCommAbort", comm->rank);
    if (comm->comm && api && api->CommAbort) {
        ncclComm_t c = comm->comm;
        const int dev = comm->ctx->device;
        auto abort_fn = api->CommAbort;
        if (!call_bounded([c, dev, abort_fn] { (void)hipSetDevice(dev); (void)abort_fn(c); }, 5000.0))
            guard_log("rank %d abort: ncclCommAbort has not returned within 5 s: left behind", comm->rank);
    }
    comm->comm = nullptr;
    guard_log("rank %d abort: waiting up to 5 s for the streams", comm->rank);
    comm->drained = comm_wait_streams(comm, 5000.0, &which, &e);
    guard_log("rank %d abort: streams %s", comm->rank, comm->drained ? "have run empty" : "are STILL busy");
    if (comm->drained && comm->last_stamp && ma_stamp_is_signal(comm->last_stamp) >= 0) {
        // nothing waits on the caller's stamp any more: it gets the sequence that was being waited for back instead of the
        // all-ones that released the wait — a stamp that kept all-ones would satisfy every wait of a later communicator at once
        comm_write_word(comm, comm->last_stamp, &comm->last_stamp_value);
        if (comm->rescue) (void)hipStreamSynchronize(comm->rescue);  // its own queue, two 8-byte writes: cannot be held
    }
    comm->last_stamp = nullptr;
    comm->broken = true;
    comm->stall_next = comm->corrupt_next = false;
}

}  // namespace

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
    // The context is NOT held while the ranks rendezvous (ncclCommInitRank returns once all n_ranks have joined — or never,
    // when one does not): a host that runs this call under its own deadline keeps a usable context either way.
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
    if (!comm->broken) {  // never an unbounded wait below: 10 s for what is in flight, then the abort path (itself bounded)
        comm_release_waits(comm, false);
        const char* which = "";
        hipError_t e = hipSuccess;
        if (!comm_wait_streams(comm, destroy_wait_ms(), &which, &e)) comm_abort(comm);
    }
    if (comm->broken && !comm->drained) {  // a stream that never ran empty: nothing here may wait for it
        delete comm;
        return;
    }
    comm_release_waits(comm, false);  // a stream held by a stall hook runs empty before it is waited for
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
    if (comm->stall_word) (void)ma_stamp_free(comm->ctx, comm->stall_word);
    if (comm->rescue) (void)hipStreamDestroy(comm->rescue);
    comm->timer.destroy();
    delete comm;
}

int32_t ma_comm_rank(ma_comm* comm) { return comm ? comm->rank : -1; }
int32_t ma_comm_size(ma_comm* comm) { return comm ? comm->n_ranks : 0; }

ma_status ma_comm_all_gather(ma_comm* comm, const void* send, void* recv, size_t bytes_per_rank) {
    MA_REQUIRE(comm != nullptr, MA_ERR_INVALID_ARGUMENT, "comm is NULL");
    MA_REQUIRE(!comm->broken, MA_ERR_DEVICE, "%s", kCommBroken);
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
    MA_REQUIRE(!comm->broken, MA_ERR_DEVICE, "%s", kCommBroken);
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
    MA_REQUIRE(!comm->broken, MA_ERR_DEVICE, "%s", kCommBroken);
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
    MA_TRY(hook_stall(comm, ctx->stream));
    const int tk = comm->timer.begin(ctx->stream);
    MA_NCCL(api, AllGather(local_records, gathered, per_rank_words * 8, ncclChar, comm->comm, ctx->stream));
    comm->timer.mark(tk, 1, ctx->stream);
    MA_TRY(hook_corrupt(comm, gathered, ctx->stream));
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
    MA_REQUIRE(!comm->broken, MA_ERR_DEVICE, "%s", kCommBroken);
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
            comm->last_stamp = stamp;
            comm->last_stamp_value = stamp_value;
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
    MA_TRY(hook_stall(comm, side->stream));
    const int tk = comm->timer.begin(side->stream);
    MA_NCCL(api, AllGather(local_records, gathered, per_rank_words * 8, ncclChar, comm->comm, side->stream));
    comm->timer.mark(tk, 1, side->stream);
    MA_TRY(hook_corrupt(comm, gathered, side->stream));
    MA_TRY(enqueue_fold_columns(side, gathered, (size_t)comm->n_ranks * slots_per_rank, n_columns * kRecordWords, n_columns,
                                out_finals));
    comm->timer.mark(tk, 2, side->stream);
    // ... and ma_comm_slot_wait(slot) puts the context's stream behind it
    MA_HIP(hipEventRecord(comm->done[slot], side->stream));
    comm->used[slot] = true;
    return MA_OK;
}

ma_status ma_comm_slot_wait(ma_comm* comm, int32_t slot) { return ma_comm_slot_wait_on(comm, slot, comm ? comm->ctx : nullptr); }

ma_status ma_comm_slot_wait_on(ma_comm* comm, int32_t slot, ma_ctx* ctx) {
    MA_REQUIRE(comm != nullptr, MA_ERR_INVALID_ARGUMENT, "comm is NULL");
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    MA_REQUIRE(ctx->device == comm->ctx->device, MA_ERR_INVALID_ARGUMENT, "the context must be on the communicator's device");
    MA_REQUIRE(slot == 0 || slot == 1, MA_ERR_INVALID_ARGUMENT, "slot must be 0 or 1");
    if (!comm->side || !comm->used[slot]) return MA_OK;
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
    MA_REQUIRE(!comm->broken, MA_ERR_DEVICE, "%s", kCommBroken);  // its sampled events may never fire
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
    MA_REQUIRE(!comm->broken || comm->drained, MA_ERR_DEVICE, "%s", kCommBroken);
    if (comm->side) {
        MA_HIP(hipSetDevice(comm->ctx->device));
        MA_HIP(hipStreamSynchronize(comm->side->stream));
    }
    return ma_ctx_synchronize(comm->ctx);
}

ma_status ma_comm_synchronize_for(ma_comm* comm, double timeout_ms) {
    MA_REQUIRE(comm != nullptr, MA_ERR_INVALID_ARGUMENT, "comm is NULL");
    MA_REQUIRE(!comm->broken, MA_ERR_DEVICE, "%s", kCommBroken);
    if (!(timeout_ms > 0)) return ma_comm_synchronize(comm);
    MA_HIP(hipSetDevice(comm->ctx->device));
    const char* which = "";
    hipError_t e = hipSuccess;
    if (!comm_wait_streams(comm, timeout_ms, &which, &e)) {
        comm_abort(comm);
        if (e != hipSuccess)
            set_error("rank %d: %s failed (%s); the communicator was aborted", comm->rank, which, hipGetErrorString(e));
        else
            set_error("rank %d of %d did not finish within %.0f ms — still pending: %s. The communicator was aborted and the streams %s",
                      comm->rank, comm->n_ranks, timeout_ms, which,
                      comm->drained ? "have run empty since" : "are STILL busy: the device may need a reset");
        return MA_ERR_DEVICE;
    }
    return ma_comm_synchronize(comm);
}

ma_status ma_comm_abort(ma_comm* comm) {
    MA_REQUIRE(comm != nullptr, MA_ERR_INVALID_ARGUMENT, "comm is NULL");
    comm_abort(comm);
    return MA_OK;
}

int32_t ma_comm_is_broken(ma_comm* comm) { return !comm || !comm->broken ? 0 : (comm->drained ? 1 : 2); }

ma_status ma_comm_test_stall_next_exchange(ma_comm* comm) {
    MA_REQUIRE(comm != nullptr, MA_ERR_INVALID_ARGUMENT, "comm is NULL");
    MA_TRY(test_hooks_enabled());
    comm->stall_next = true;
    return MA_OK;
}

ma_status ma_comm_test_corrupt_next_exchange(ma_comm* comm) {
    MA_REQUIRE(comm != nullptr, MA_ERR_INVALID_ARGUMENT, "comm is NULL");
    MA_TRY(test_hooks_enabled());
    comm->corrupt_next = true;
    return MA_OK;
}

ma_status ma_comm_selftest(ma_comm* comm, uint32_t what, double timeout_ms, ma_selftest_report* out_report) {
    MA_REQUIRE(comm != nullptr, MA_ERR_INVALID_ARGUMENT, "comm is NULL");
    MA_REQUIRE(timeout_ms > 0, MA_ERR_INVALID_ARGUMENT, "the self-test needs a deadline (timeout_ms > 0)");
    ma_selftest_report local_report;
    ma_selftest_report* rep = out_report ? out_report : &local_report;
    memset(rep, 0, sizeof(*rep));
    rep->struct_bytes = (uint32_t)sizeof(*rep);
    rep->failed_form = rep->failed_member = -1;
    MA_REQUIRE(!comm->broken, MA_ERR_DEVICE, "%s", kCommBroken);
    if (what == 0 || (what & MA_SELFTEST_EXCHANGE_ALL_FORMS)) what = MA_SELFTEST_EXCHANGE | MA_SELFTEST_OVERLAP_EVENT | MA_SELFTEST_OVERLAP_STAMP;
    ma_ctx* ctx = comm->ctx;
    const size_t n = (size_t)comm->n_ranks;
    constexpr size_t kCols = 4, kWords = kCols * kRecordWords;
    rep->n_members = rep->n_devices = comm->n_ranks;
    rep->exchange_kind = 1;
    {
        int ranks = 0;
        const RcclApi* api = rccl();
        if (api && api->CommCount && api->CommCount(comm->comm, &ranks) == ncclSuccess) rep->rccl_ranks = ranks;
    }
    std::string text;
    char buf[400];
    snprintf(buf, sizeof(buf), "rank %d of %d, RCCL exchange", comm->rank, comm->n_ranks);
    text = buf;
    MA_HIP(hipSetDevice(ctx->device));
    uint64_t *local = nullptr, *gathered = nullptr, *finals = nullptr, *stamp = nullptr;
    auto cleanup = [&] {
        if (comm->broken && !comm->drained) return;  // a stuck stream may still use them
        if (local) (void)hipFree(local);
        if (gathered) (void)hipFree(gathered);
        if (finals) (void)hipFree(finals);
        if (stamp) (void)ma_stamp_free(ctx, stamp);
    };
    hipError_t e = hipMalloc((void**)&local, kWords * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&gathered, n * kWords * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&finals, kCols * 4 * 8);
    if (e != hipSuccess) {
        cleanup();
        return hip_fail(e, "self-test buffers", __FILE__, __LINE__);
    }
    ma_status st = ma_comm_synchronize_for(comm, timeout_ms);  // whatever the host had in flight first
    if (st == MA_OK && ma_stamp_alloc(ctx, &stamp) != MA_OK) stamp = nullptr;
    uint64_t stamp_seq = 0, round = 0;
    NoSync enqueue_only;  // the exchanges below only enqueue, whatever the context's mode: every wait here is a bounded one
    struct Form {
        int bit, slot;  // slot -1: in-stream
        bool on_stamp;
    };
    const Form forms[] = {{MA_SELFTEST_FORM_IN_STREAM_CALLER, -1, false}, {MA_SELFTEST_FORM_OVERLAP_EVENT_CALLER, 0, false},
                          {MA_SELFTEST_FORM_OVERLAP_EVENT_CALLER, 1, false}, {MA_SELFTEST_FORM_OVERLAP_STAMP_CALLER, 0, true},
                          {MA_SELFTEST_FORM_OVERLAP_STAMP_CALLER, 1, true}};
    static const char* const kNames[8] = {"", "in-stream", "", "overlap-event", "", "overlap-stamp", "", ""};
    std::vector<std::vector<uint64_t>> blocks;
    for (const Form& f : forms) {
        if (st != MA_OK) break;
        const uint32_t need = f.slot < 0 ? MA_SELFTEST_EXCHANGE : (f.on_stamp ? MA_SELFTEST_OVERLAP_STAMP : MA_SELFTEST_OVERLAP_EVENT);
        if (!(what & need)) continue;
        if (f.on_stamp && (!stamp || comm->no_wait_value)) continue;
        rep->forms_tried |= 1u << f.bit;
        ++round;
        blocks.assign(n, std::vector<uint64_t>(kWords, 0));
        for (size_t i = 0; i < n; ++i)
            for (size_t c = 0; c < kCols; ++c) {
                uint64_t* r = &blocks[i][c * kRecordWords];
                r[0] = 0x0101010101010101ull * (i + 1) + c + round * 1000003ull;
                r[1] = i + 1;
                const double hi = (double)(i + 1) * 1e3 + (double)c + 0.5, lo = std::ldexp((double)(i + 1), -70);
                memcpy(&r[2], &hi, 8);
                memcpy(&r[3], &lo, 8);
                r[4] = 3 * (i + 1);
                for (size_t w = 5; w < kRecordWords; ++w) r[w] = 0xA5A5000000000000ull ^ (i << 16) ^ (c << 8) ^ w ^ (round << 32);
            }
        const auto t0 = std::chrono::steady_clock::now();
        hipError_t he = hipMemcpyAsync(local, blocks[(size_t)comm->rank].data(), kWords * 8, hipMemcpyHostToDevice, ctx->stream);
        if (he == hipSuccess && f.on_stamp) he = launch_stamp_store(ctx->stream, stamp, ++stamp_seq);
        if (he != hipSuccess) {
            st = hip_fail(he, "self-test records", __FILE__, __LINE__);
            break;
        }
        if (f.slot < 0)
            st = ma_comm_sum_exchange(comm, local, 1, kCols, gathered, finals);
        else if (f.on_stamp)
            st = ma_comm_sum_exchange_overlapped_on_stamp(comm, f.slot, stamp, stamp_seq, local, 1, kCols, gathered, finals);
        else
            st = ma_comm_sum_exchange_overlapped(comm, f.slot, local, 1, kCols, gathered, finals);
        if (st == MA_OK) st = ma_comm_synchronize_for(comm, timeout_ms);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (us > rep->form_us[f.bit]) rep->form_us[f.bit] = us;
        if (st != MA_OK) {
            rep->failed_form = f.bit;
            rep->failed_member = comm->rank;
            rep->timed_out = comm->broken ? 1 : 0;
            text += std::string("; ") + kNames[f.bit] + ": " + ma_last_error_string();
            break;
        }
        std::vector<uint64_t> got(n * kWords), fin(kCols * 4);
        he = hipMemcpy(got.data(), gathered, got.size() * 8, hipMemcpyDeviceToHost);
        if (he == hipSuccess) he = hipMemcpy(fin.data(), finals, fin.size() * 8, hipMemcpyDeviceToHost);
        if (he != hipSuccess) {
            st = hip_fail(he, "self-test read-back", __FILE__, __LINE__);
            break;
        }
        const char* bad = nullptr;
        for (size_t i = 0; i < n && !bad; ++i)
            if (memcmp(&got[i * kWords], blocks[i].data(), kWords * 8) != 0) bad = "the gathered blocks are not the ranks' records in rank order";
        for (size_t c = 0; c < kCols && !bad; ++c) {
            uint64_t isum = 0, icnt = 0, fcnt = 0;
            double hi = 0.0, lo = 0.0;
            for (size_t i = 0; i < n; ++i) {  // the rank-ordered fold: wrapping adds, error-free two-sum for the pairs
                const uint64_t* p = &blocks[i][c * kRecordWords];
                isum += p[0];
                icnt += p[1];
                double h, l;
                memcpy(&h, &p[2], 8);
                memcpy(&l, &p[3], 8);
                const double t = hi + h, bp = t - hi;
                const double err = (hi - (t - bp)) + (h - bp);
                hi = t;
                lo += err + l;
                fcnt += p[4];
            }
            const double total = hi + lo;
            uint64_t want[4] = {isum, icnt, 0, fcnt};
            memcpy(&want[2], &total, 8);
            if (memcmp(&fin[c * 4], want, 32) != 0) bad = "finals differ from the rank-ordered fold of the tagged records";
        }
        if (bad) {
            rep->failed_form = f.bit;
            rep->failed_member = comm->rank;
            set_error("self-test, rank %d, form %s: %s", comm->rank, kNames[f.bit], bad);
            text += std::string("; ") + kNames[f.bit] + ": " + bad;
            st = MA_ERR_DEVICE;
            break;
        }
        if (f.slot != 0) {  // an overlapped form has passed once both of its record sets have
            rep->forms_ok |= 1u << f.bit;
            snprintf(buf, sizeof(buf), "; %s ok %.0f us", kNames[f.bit], rep->form_us[f.bit]);
            text += buf;
        }
    }
    if (!comm->broken) {
        const std::string keep = st != MA_OK ? ma_last_error_string() : "";
        const ma_status drained = ma_comm_synchronize_for(comm, timeout_ms);
        if (st == MA_OK) st = drained;
        else set_error("%s", keep.c_str());
        if (!comm->broken) {
            comm->last_stamp = nullptr;  // the test's stamp goes away with it
            if (comm->side) {
                std::lock_guard<std::mutex> lock(comm->side->mu);
                comm->timer.report(nullptr, nullptr, nullptr);
            } else {
                comm->timer.report(nullptr, nullptr, nullptr);
            }
        }
    }
    cleanup();
    snprintf(rep->text, sizeof(rep->text), "%s%s", st == MA_OK ? "PASS: " : "FAIL: ", text.c_str());
    return st;
}

}  // extern "C"
