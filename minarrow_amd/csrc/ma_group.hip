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
//
// Who issues. The reference's Rayon pool issues from all cores (benches/benchmark_parallel_simd.rs:83-87). One host
// thread issuing for 8 GPUs is launch-bound below ~0.25 ms per step (3 launches x 8 members x ~6-8 us each) — exactly the
// step a 10^9-row column partitioned over 8 GPUs has (0.14 ms of scan per GPU per column). Every member therefore has a
// persistent ISSUE THREAD (round 3): a group entry point validates its arguments on the calling thread, hands one job to
// all members' threads, each thread enqueues its member's launches (its own hipSetDevice, its own stream, its own RCCL
// communicator — the "one thread per device" form RCCL supports without ncclGroupStart/End), and the call returns when
// every member has ENQUEUED — so the semantics of round 2 hold (work is on the streams when the call returns, launch
// errors are returned by the call) while the host cost per call is that of ONE member's launches plus a hand-off.
// Threads spin for ~200 us after a job (a stepping host never pays a wake-up) and then sleep on a condition variable.
// MA_GROUP_ISSUE_CALLER / MINARROW_HIP_GROUP_ISSUE=caller keeps the round-2 form (the calling thread loops over the
// members; the all-gathers inside one ncclGroupStart/End) for A/B measurements and as a fallback.
#include <cstdlib>
#include <map>

#include "minarrow_hip_testing.h"

#include "ma_group.hpp"

using namespace ma;
using namespace ma::grp;

namespace {

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

}  // namespace

namespace ma {
namespace grp {

// Frees the exchange's buffers and communicators (either kind); the members stay.
void release_exchange(ma_group* g) {
    release_waits(g, false);  // a stream held by a stall hook must run empty before it is waited for
    for (size_t i = 0; i < g->ctxs.size(); ++i) {
        (void)hipSetDevice(g->ctxs[i]->device);
        (void)hipStreamSynchronize(g->ctxs[i]->stream);
    }
    for (size_t i = 0; i < g->side.size(); ++i) {
        if (!g->side[i]) continue;
        (void)hipSetDevice(g->ctxs[i]->device);
        (void)hipStreamSynchronize(g->side[i]->stream);
    }
    for (size_t i = 0; i < g->scan2.size(); ++i) {
        if (!g->scan2[i]) continue;
        (void)hipSetDevice(g->ctxs[i]->device);
        (void)hipStreamSynchronize(g->scan2[i]->stream);
    }
    if (g->use_rccl || !g->comms.empty()) {
        const RcclApi* api = rccl();
        for (ncclComm_t c : g->comms)
            if (c && api) (void)api->CommDestroy(c);
        for (size_t i = 0; i < g->ctxs.size(); ++i) {
            (void)hipSetDevice(g->ctxs[i]->device);
            if (i < g->local.size() && g->local[i]) (void)hipFree(g->local[i]);
            if (i < g->gathered.size() && g->gathered[i]) (void)hipFree(g->gathered[i]);
            if (i < g->local1.size() && g->local1[i]) (void)hipFree(g->local1[i]);
            if (i < g->gathered1.size() && g->gathered1[i]) (void)hipFree(g->gathered1[i]);
            for (int k = 0; k < 2; ++k) {
                if (i < g->ev_ready[k].size() && g->ev_ready[k][i]) (void)hipEventDestroy(g->ev_ready[k][i]);
                if (i < g->ev_done[k].size() && g->ev_done[k][i]) (void)hipEventDestroy(g->ev_done[k][i]);
            }
            if (i < g->side.size() && g->side[i]) ma_ctx_destroy(g->side[i]);
            if (i < g->scan2.size() && g->scan2[i]) ma_ctx_destroy(g->scan2[i]);
            if (i < g->ev_lane.size() && g->ev_lane[i]) (void)hipEventDestroy(g->ev_lane[i]);
            for (int k = 0; k < 2; ++k)
                if (i < g->stamp[k].size() && g->stamp[k][i]) (void)ma_stamp_free(g->ctxs[i], g->stamp[k][i]);
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
    g->local1.clear();
    g->gathered1.clear();
    g->finals1.clear();
    g->side.clear();
    g->scan2.clear();
    g->ev_lane.clear();
    g->seen_calls.clear();
    g->lanes2 = false;
    g->prev_set = -1;
    for (int k = 0; k < 2; ++k) {
        g->ev_ready[k].clear();
        g->ev_done[k].clear();
        g->set_used[k] = false;
        g->stamp[k].clear();
        g->stamp_ok[k] = false;
    }
    for (int k = 0; k < 2; ++k) g->enq_mask[k] = g->exchanged_mask[k] = 0;
    g->overlap = false;
    g->cur = g->last = 0;
    g->use_rccl = false;
    g->timer.destroy();
}

}  // namespace grp
}  // namespace ma

namespace {

void destroy_members(ma_group* g) {
    stop_workers(g);
    if (g->broken && !g->drained) {
        // a stream that never ran empty after the abort: waiting for it (hipStreamSynchronize, hipFree) would block for good.
        // What the group holds goes with the process instead.
        g->ctxs.clear();
        return;
    }
    release_exchange(g);  // drains every member's stream first
    for (size_t i = 0; i < g->mask_stage.size() && i < g->ctxs.size(); ++i)
        if (g->mask_stage[i]) {
            (void)hipSetDevice(g->ctxs[i]->device);
            (void)hipFree(g->mask_stage[i]);
        }
    g->mask_stage.clear();
    g->mask_stage_bytes.clear();
    for (size_t i = 0; i < g->ctxs.size(); ++i) {
        (void)hipSetDevice(g->ctxs[i]->device);
        if (i < g->stall_word.size() && g->stall_word[i]) (void)ma_stamp_free(g->ctxs[i], g->stall_word[i]);
        if (i < g->rescue.size() && g->rescue[i]) {  // one per device, shared by that device's members
            bool first = true;
            for (size_t j = 0; j < i; ++j) first = first && g->rescue[j] != g->rescue[i];
            if (first) (void)hipStreamDestroy(g->rescue[i]);
        }
    }
    g->stall_word.clear();
    g->rescue.clear();
    for (ma_ctx* c : g->ctxs) ma_ctx_destroy(c);
    g->ctxs.clear();
}

}  // namespace

namespace ma {
namespace grp {

// RCCL exchange set-up. Returns MA_OK with g->use_rccl set, or a status + the thread's error string.
ma_status setup_rccl(ma_group* g, bool overlap, bool lanes) {
    const size_t n = g->ctxs.size();
    const RcclApi* api = rccl();
    if (!api) return MA_ERR_UNSUPPORTED;
    for (size_t i = 0; i < n && !api->loopback; ++i)  // the loopback double (a rehearsal on one GPU) takes ranks that share a device
        for (size_t j = i + 1; j < n; ++j)
            MA_REQUIRE(g->ctxs[i]->device != g->ctxs[j]->device, MA_ERR_UNSUPPORTED,
                       "an RCCL communicator needs distinct devices (members %zu and %zu share device %d)", i, j,
                       g->ctxs[i]->device);
    std::vector<int> devs(n);
    for (size_t i = 0; i < n; ++i) devs[i] = g->ctxs[i]->device;
    g->comms.assign(n, nullptr);  // non-empty from here on: release_exchange frees the RCCL-side resources
    MA_NCCL(api, CommInitAll(g->comms.data(), (int)n, devs.data()));
    g->use_rccl = true;
    g->local.assign(n, nullptr);
    g->gathered.assign(n, nullptr);
    g->finals.assign(n, nullptr);
    const size_t sets = overlap ? 2 : 1;
    MA_HIP(hipHostMalloc((void**)&g->host_finals, sets * n * kColumns * 4 * 8, hipHostMallocPortable | hipHostMallocMapped));
    memset(g->host_finals, 0, sets * n * kColumns * 4 * 8);
    if (overlap) {
        g->local1.assign(n, nullptr);
        g->gathered1.assign(n, nullptr);
        g->finals1.assign(n, nullptr);
        g->side.assign(n, nullptr);
        for (int k = 0; k < 2; ++k) {
            g->ev_ready[k].assign(n, nullptr);
            g->ev_done[k].assign(n, nullptr);
        }
    }
    for (size_t i = 0; i < n; ++i) {
        MA_HIP(hipSetDevice(devs[i]));
        MA_HIP(device_malloc(devs[i], (void**)&g->local[i], kBlockWords * 8));
        MA_HIP(device_malloc(devs[i], (void**)&g->gathered[i], n * kBlockWords * 8));
        MA_HIP(hipMemset(g->local[i], 0, kBlockWords * 8));
        MA_HIP(hipMemset(g->gathered[i], 0, n * kBlockWords * 8));
        g->finals[i] = g->host_finals + i * kColumns * 4;
        if (overlap) {
            MA_HIP(device_malloc(devs[i], (void**)&g->local1[i], kBlockWords * 8));
            MA_HIP(device_malloc(devs[i], (void**)&g->gathered1[i], n * kBlockWords * 8));
            MA_HIP(hipMemset(g->local1[i], 0, kBlockWords * 8));
            MA_HIP(hipMemset(g->gathered1[i], 0, n * kBlockWords * 8));
            g->finals1[i] = g->host_finals + (n + i) * kColumns * 4;
            MA_TRY(make_lane(g->ctxs[i], &g->side[i], g->carrier_class));
            for (int k = 0; k < 2; ++k) {
                MA_HIP(hipEventCreateWithFlags(&g->ev_ready[k][i], hipEventDisableTiming | hipEventReleaseToDevice));
                MA_HIP(hipEventCreateWithFlags(&g->ev_done[k][i], hipEventDisableTiming | hipEventReleaseToDevice));
                if (g->stamp[k].size() < n) g->stamp[k].assign(n, nullptr);
                if (ma_stamp_alloc(g->ctxs[i], &g->stamp[k][i]) != MA_OK) g->stamp[k][i] = nullptr;  // -> the event path
                // does this runtime let a stream wait on that word? (0 >= 0: satisfied at once) — else the event path
                if (g->stamp[k][i] &&
                    (hipStreamWaitValue64(g->side[i]->stream, g->stamp[k][i], 0, hipStreamWaitValueGte, ~(uint64_t)0) != hipSuccess ||
                     hipStreamSynchronize(g->side[i]->stream) != hipSuccess)) {
                    (void)hipGetLastError();
                    (void)ma_stamp_free(g->ctxs[i], g->stamp[k][i]);
                    g->stamp[k][i] = nullptr;
                }
            }
        }
    }
    g->overlap = overlap;
    // two scan lanes: only with stamps on every member (the early stamp is word 1 of the stamp's 64-byte device line)
    bool all_stamps = overlap;
    for (int k = 0; k < 2 && all_stamps; ++k) {
        all_stamps = g->stamp[k].size() == n;
        for (size_t i = 0; all_stamps && i < n; ++i) all_stamps = g->stamp[k][i] != nullptr && ma_stamp_is_signal(g->stamp[k][i]) == 0;
    }
    if (lanes && all_stamps) {
        g->scan2.assign(n, nullptr);
        g->ev_lane.assign(n, nullptr);
        g->seen_calls.assign(n, 0);
        for (size_t i = 0; i < n; ++i) {
            // An independent context (its own stream, partials, tickets). What the lanes gain depends on how the runtime maps the
            // member's streams onto hardware queues, process by process: the step of the 8-way share 0.287 -> 0.275-0.279 ms on most
            // boxes, nothing (or +1 %) on some, where lane 1's wait for the early stamp ends up behind lane 0's whole scan in one
            // hardware queue. Putting lane 1 into the HIGH priority class (its own queue pool: MINARROW_HIP_SCAN_LANE_CLASS=high)
            // makes the bare scans gain every time (tools/probe_early_stamp.py high) but loses the gain once the exchange stream
            // is in the picture (tools/run_share_lanes.sh: 0.287 against 0.275-0.279) — so: the ordinary class, and a host that
            // MEASURES both forms before it settles on one (ma_group_set_scan_lanes; bench.py's un-timed trial).
            // (a rehearsal — members sharing one device, their own streams in the high class, ma_group_create_ex — puts the second
            // lanes there too: in a lower class than the first lanes they would only run when those are idle)
            const char* cls = tuning_env("MINARROW_HIP_SCAN_LANE_CLASS");
            MA_TRY(create_ctx_in_class(g->ctxs[i]->ordinal, g->carrier_class ? g->carrier_class : (cls && cls[0] == 'h') ? +1 : 0, &g->scan2[i]));
            MA_TRY(ma_ctx_set_async(g->scan2[i], 1));
            MA_HIP(hipSetDevice(devs[i]));
            MA_HIP(hipEventCreateWithFlags(&g->ev_lane[i], hipEventDisableTiming | hipEventReleaseToDevice));
            g->seen_calls[i] = g->ctxs[i]->calls.load();
        }
        g->lanes2 = true;
        g->lanes_on = true;
    }
    return MA_OK;
}

ma_status order_lane_if_foreign(ma_group* g, size_t i, uint64_t now) {
    if (!g->lanes2 || !g->lanes_on || now == g->seen_calls[i]) return MA_OK;
    MA_HIP(hipSetDevice(g->ctxs[i]->device));
    MA_HIP(hipEventRecord(g->ev_lane[i], g->ctxs[i]->stream));
    MA_HIP(hipStreamWaitEvent(g->scan2[i]->stream, g->ev_lane[i], 0));
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

// ---- issue threads ------------------------------------------------------------------------------------------------

static void worker_main(ma_group* g, size_t i, uint64_t seen) {  // seen = the job sequence at the thread's start
    (void)hipSetDevice(g->ctxs[i]->device);
    for (;;) {
        // wait for job `seen + 1`: spin for ~200 us (a stepping host never pays a wake-up), then sleep
        const auto t0 = std::chrono::steady_clock::now();
        unsigned spins = 0;
        while (g->job_seq.load() == seen && !g->stop.load()) {
            __builtin_ia32_pause();
            if ((++spins & 255) == 0 &&
                std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() >= 200) {
                std::unique_lock<std::mutex> lock(g->sleep_mu);
                g->sleepers.fetch_add(1);
                g->work_cv.wait(lock, [&] { return g->job_seq.load() != seen || g->stop.load(); });
                g->sleepers.fetch_sub(1);
            }
        }
        if (g->job_seq.load() == seen) return;  // stop, nothing posted
        ++seen;
        ma_group::Slot& slot = g->slots[i];
        slot.status = (*g->job)(i);
        if (slot.status != MA_OK) slot.message = ma_last_error_string();
        // seq_cst on purpose (here and in run_on_members): "publish done, then look whether the caller sleeps" against the
        // caller's "announce sleeping, then look at done" is a store-load pattern on both sides; with a release store the
        // load may pass it and both could miss each other — a caller asleep for good
        slot.done.store(seen);
        if (g->caller_waiting.load() != 0) {
            { std::lock_guard<std::mutex> lock(g->sleep_mu); }
            g->done_cv.notify_all();
        }
    }
}

void start_workers(ma_group* g) {
    const size_t n = g->ctxs.size();
    g->slots = std::vector<ma_group::Slot>(n);
    g->workers.reserve(n);
    const uint64_t base = g->job_seq.load();  // a set of workers started later (ma_group_rebuild_exchange) joins here
    for (size_t i = 0; i < n; ++i) g->workers.emplace_back(worker_main, g, i, base);
    g->threads = true;
}

void stop_workers(ma_group* g) {
    if (!g->threads) return;
    g->stop.store(true);
    { std::lock_guard<std::mutex> lock(g->sleep_mu); }
    g->work_cv.notify_all();
    for (std::thread& t : g->workers)
        if (t.joinable()) t.join();
    g->workers.clear();
    g->threads = false;
    g->stop.store(false);  // the group may start a fresh set later (ma_group_rebuild_exchange)
}

// Runs fn(member) for every member — on the members' issue threads, concurrently, or in a loop on the calling thread — and
// returns once all have returned: the first failing member's status, its message in the CALLER's error string.
// The caller holds g->mu (one job at a time; two calling threads post their collectives in one order to all members).
ma_status run_on_members(ma_group* g, const std::function<ma_status(size_t)>& fn) {
    const size_t n = g->ctxs.size();
    if (!g->threads) {
        for (size_t i = 0; i < n; ++i) MA_TRY(fn(i));
        return MA_OK;
    }
    g->job = &fn;
    const uint64_t seq = g->job_seq.fetch_add(1) + 1;
    if (g->sleepers.load() != 0) {
        { std::lock_guard<std::mutex> lock(g->sleep_mu); }
        g->work_cv.notify_all();
    }
    auto all_done = [&] {
        for (size_t i = 0; i < n; ++i)
            if (g->slots[i].done.load() != seq) return false;
        return true;
    };
    // launches take microseconds, a synchronise takes milliseconds: spin briefly, then block
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (!all_done()) {
        __builtin_ia32_pause();
        if ((++spins & 255) == 0 &&
            std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() >= 100) {
            std::unique_lock<std::mutex> lock(g->sleep_mu);
            g->caller_waiting.store(1);
            g->done_cv.wait(lock, all_done);
            g->caller_waiting.store(0);
        }
    }
    g->job = nullptr;
    for (size_t i = 0; i < n; ++i)
        if (g->slots[i].status != MA_OK) {
            set_error("%s", g->slots[i].message.c_str());
            return g->slots[i].status;
        }
    return MA_OK;
}

}  // namespace grp
}  // namespace ma

namespace {

// A chunk handed to member `member` must be resident on that member's device, or be host memory the device can
// address: the kernels dereference it directly, and on a multi-GPU node a pointer into another GPU's HBM is a memory
// fault, not a status. (one row chunk per GPU.)
ma_status require_resident(ma_group* g, DeviceLookup& lookup, size_t member, const void* p, const char* what, size_t chunk) {
    if (p == nullptr) return MA_OK;
    const int want = g->home[member];
    const int dev = lookup.device_of(p);
    MA_REQUIRE(dev < 0 || dev == want, MA_ERR_INVALID_ARGUMENT,
               "chunk %zu belongs to member %zu (device %d) but its %s is resident on device %d", chunk, member, want, what, dev);
    return MA_OK;
}

ma_status enqueue_sum_members(ma_group* g, int32_t column, const void* const* chunk_data, const size_t* chunk_lens,
                              const uint8_t* const* chunk_masks, const std::function<ma_status(size_t, ma_ctx*, uint64_t*)>& launch) {
    MA_REQUIRE(g != nullptr, MA_ERR_INVALID_ARGUMENT, "group is NULL");
    MA_REQUIRE(column >= 0 && column < kColumns, MA_ERR_INVALID_ARGUMENT, "column %d out of range [0,%d)", column, kColumns);
    std::lock_guard<std::recursive_mutex> lock(g->mu);
    MA_REQUIRE(!g->broken, MA_ERR_DEVICE, "%s", kBrokenMessage);
    DeviceLookup lookup;
    for (size_t i = 0; i < g->ctxs.size(); ++i) {
        if (chunk_lens[i] == 0) continue;
        MA_TRY(require_resident(g, lookup, i, chunk_data[i], "data", i));
        if (chunk_masks) MA_TRY(require_resident(g, lookup, i, chunk_masks[i], "validity bitmap", i));
    }
    const int cur_set = g->overlap ? g->cur : 0;
    g->enq_mask[cur_set] |= 1u << column;
    g->stamp_ok[cur_set] = false;  // not a stamped launch: this set's exchange waits on an event
    g->prev_set = -1;
    // enqueue only (the members are in async mode): all devices run concurrently
    return run_on_members(g, [&](size_t i) {
        uint64_t* set = (g->overlap && g->cur == 1) ? g->local1[i] : g->local[i];
        if (cur_set == 1) MA_TRY(order_lane_if_foreign(g, i, g->ctxs[i]->calls.load(std::memory_order_relaxed)));
        return launch(i, scan_ctx(g, cur_set, i), set + (size_t)column * kRecordWords);
    });
}

}  // namespace

namespace ma {
namespace grp {

ma_status exchange_locked(ma_group* g) {
    MA_REQUIRE(!g->broken, MA_ERR_DEVICE, "%s", kBrokenMessage);
    if (!g->use_rccl) {  // host exchange: the records are already in host memory once the streams drain
        const int m = g->stall_member;
        g->stall_member = -1;
        if (m >= 0) MA_TRY(enqueue_stall(g, (size_t)m, g->ctxs[(size_t)m]->stream));
        return MA_OK;
    }
    const RcclApi* api = rccl();
    if (!api) return MA_ERR_UNSUPPORTED;
    const size_t n = g->ctxs.size();
    const int set = g->overlap ? g->cur : 0;
    auto local = [g, set](size_t i) { return set ? g->local1[i] : g->local[i]; };
    auto gathered = [g, set](size_t i) { return set ? g->gathered1[i] : g->gathered[i]; };
    auto finals = [g, set](size_t i) { return set ? g->finals1[i] : g->finals[i]; };
    // which context issues the exchange of member i: its own (in-stream), or its side context (overlapped: behind the
    // scans that filled this set — an event — while the member's stream goes on with the other set)
    // ma_group_set_handoff(event) — or variant bit 4096 on member 0, the tuning harness's switch — always takes the event
    const bool on_stamp = g->overlap && g->stamp_ok[set] && g->handoff == 0 && !(tuning_variant(g->ctxs[0]) & 4096);
    auto before = [g, set, on_stamp](size_t i) -> ma_status {
        if (!g->overlap) return MA_OK;
        if (on_stamp) {  // the scan stream carries nothing for the hand-off: the exchange stream waits for the kernel's stamp
            MA_HIP(hipStreamWaitValue64(g->side[i]->stream, g->stamp[set][i], g->stamp_seq[set], hipStreamWaitValueGte, ~(uint64_t)0));
            return MA_OK;
        }
        MA_HIP(hipEventRecord(g->ev_ready[set][i], scan_ctx(g, set, i)->stream));
        MA_HIP(hipStreamWaitEvent(g->side[i]->stream, g->ev_ready[set][i], 0));
        return MA_OK;
    };
    auto after = [g, set](size_t i) -> ma_status {
        if (!g->overlap) return MA_OK;
        MA_HIP(hipEventRecord(g->ev_done[set][i], g->side[i]->stream));
        // the member's stream fills the OTHER set next: behind that set's last exchange (long finished, normally)
        if (g->set_used[set ^ 1]) {
            const hipError_t q = hipEventQuery(g->ev_done[set ^ 1][i]);  // finished already: no barrier packet on the stream
            if (q == hipErrorNotReady) {
                (void)hipGetLastError();
                MA_HIP(hipStreamWaitEvent(scan_ctx(g, set ^ 1, i)->stream, g->ev_done[set ^ 1][i], 0));  // the lane that fills it
            } else if (q != hipSuccess) {
                (void)hipGetLastError();
                return hip_fail(q, "hipEventQuery(exchange done)", __FILE__, __LINE__);
            }
        }
        return MA_OK;
    };
    auto fold = [g, n, gathered, finals](size_t i) -> ma_status {
        ma_ctx* c = g->overlap ? g->side[i] : g->ctxs[i];
        std::lock_guard<std::mutex> lock(c->mu);
        MA_HIP(hipSetDevice(c->device));
        return enqueue_fold_columns(c, gathered(i), n, kBlockWords, kColumns, finals(i));
    };
    auto xstream = [g](size_t i) { return g->overlap ? g->side[i]->stream : g->ctxs[i]->stream; };
    // testing hooks: a member whose exchange never starts (held behind its stall word), a member that folds flipped records
    const int stall = g->stall_member, corrupt = g->corrupt_member;
    g->stall_member = g->corrupt_member = -1;
    auto hooks_before = [&](size_t i) -> ma_status { return (int)i == stall ? enqueue_stall(g, i, xstream(i)) : MA_OK; };
    auto hooks_gathered = [&](size_t i) -> ma_status {
        if ((int)i != corrupt) return MA_OK;
        static const uint64_t kFlip = 0x5A5A5A5A5A5A5A5Aull;  // member i's copy of member 0's record 0, integer sum
        MA_HIP(hipMemcpyAsync(gathered(i), &kFlip, 8, hipMemcpyHostToDevice, xstream(i)));
        return MA_OK;
    };
    ma_status st = MA_OK;
    // One thread per device, each issuing its own rank's all-gather with no ncclGroup: the form RCCL documents for one thread
    // per device; the members' launches and collectives are enqueued concurrently (8 members: 26 us per step instead of 53 from
    // one thread, profiles/r03_group_issue.json). The price: if one member fails before its call
    // (hipSetDevice, the event record, the all-gather itself) the others have already enqueued theirs and those collectives
    // can never complete. That case is handled below instead of avoided: every communicator is aborted (ncclCommAbort ends the
    // kernels in flight) and the group is marked broken, so that the next synchronize returns an error instead of blocking.
    if (g->threads) {
        st = run_on_members(g, [&](size_t i) -> ma_status {
            MA_HIP(hipSetDevice(g->ctxs[i]->device));
            MA_TRY(before(i));
            if ((int)i == g->fail_member) {
                set_error("member %zu: injected failure in front of its all-gather (ma_group_test_fail_next_exchange)", i);
                return MA_ERR_DEVICE;
            }
            MA_TRY(hooks_before(i));
            const int tk = i == 0 ? g->timer.begin(xstream(0)) : -1;
            MA_NCCL(api, AllGather(local(i), gathered(i), kBlockWords * 8, ncclChar, g->comms[i], xstream(i)));
            if (i == 0) g->timer.mark(tk, 1, xstream(0));
            MA_TRY(hooks_gathered(i));
            MA_TRY(fold(i));
            if (i == 0) g->timer.mark(tk, 2, xstream(0));
            return after(i);
        });
    } else {
        for (size_t i = 0; i < n && st == MA_OK; ++i) {
            st = hipSetDevice(g->ctxs[i]->device) == hipSuccess ? before(i) : MA_ERR_DEVICE;
            if (st == MA_OK) st = hooks_before(i);
        }
        MA_TRY(st);
        MA_HIP(hipSetDevice(g->ctxs[0]->device));
        const int tk = g->timer.begin(xstream(0));
        MA_NCCL(api, GroupStart());
        for (size_t i = 0; i < n; ++i) {
            ncclResult_t r = api->AllGather(local(i), gathered(i), kBlockWords * 8, ncclChar, g->comms[i], xstream(i));
            if (r != ncclSuccess) {
                (void)api->GroupEnd();
                return rccl_fail(r, "AllGather", __FILE__, __LINE__);
            }
        }
        MA_NCCL(api, GroupEnd());
        MA_HIP(hipSetDevice(g->ctxs[0]->device));
        g->timer.mark(tk, 1, xstream(0));
        // the folds (and the events that let the scan streams go on) go out from the members' issue threads; member 0's
        // timing mark sits right behind its fold
        st = run_on_members(g, [&](size_t i) -> ma_status {
            MA_HIP(hipSetDevice(g->ctxs[i]->device));
            MA_TRY(hooks_gathered(i));
            MA_TRY(fold(i));
            if (i == 0) g->timer.mark(tk, 2, xstream(0));
            return after(i);
        });
    }
    const bool injected = g->fail_member >= 0;
    g->fail_member = -1;
    if (st != MA_OK && g->threads && (n > 1 || injected)) {
        // some member's all-gather is on its stream without its peers': abort every communicator so that nothing waits for it
        const std::string why = ma_last_error_string();
        abort_locked(g, why.c_str());
        set_error("the group's exchange failed on a member and its communicators were aborted (ma_group_rebuild_exchange, or "
                  "destroy the group): %s", why.c_str());
        return st;
    }
    MA_TRY(st);
    if (g->overlap) {
        g->set_used[set] = true;
        g->last = set;
        g->cur = set ^ 1;
        g->exchanged_mask[set] = g->enq_mask[set];
        g->enq_mask[set] = 0;
        g->enq_mask[set ^ 1] = 0;  // the set about to be filled starts empty
        g->stamp_ok[set] = false;
    }
    return MA_OK;
}

ma_status synchronize_locked(ma_group* g) {
    MA_REQUIRE(!g->broken, MA_ERR_DEVICE, "%s", kBrokenMessage);
    // every member is waited for even when one reports (a latched division by zero is cleared by its report)
    std::vector<ma_status> st(g->ctxs.size(), MA_OK);
    std::vector<std::string> msg(g->ctxs.size());
    (void)run_on_members(g, [&](size_t i) -> ma_status {
        st[i] = ma_ctx_synchronize(g->ctxs[i]);
        if (st[i] != MA_OK) msg[i] = ma_last_error_string();
        if (g->lanes2 && st[i] == MA_OK) {
            st[i] = ma_ctx_synchronize(g->scan2[i]);
            if (st[i] != MA_OK) msg[i] = ma_last_error_string();
        }
        if (g->overlap && st[i] == MA_OK && hipStreamSynchronize(g->side[i]->stream) != hipSuccess) {
            st[i] = MA_ERR_DEVICE;
            msg[i] = "the exchange stream of a group member failed";
        }
        return MA_OK;
    });
    for (size_t i = 0; i < st.size(); ++i)
        if (st[i] != MA_OK) {
            set_error("%s", msg[i].c_str());
            return st[i];
        }
    if (g->lanes2)  // everything has finished: whatever the host enqueued itself is behind us
        for (size_t i = 0; i < g->ctxs.size(); ++i) g->seen_calls[i] = g->ctxs[i]->calls.load(std::memory_order_relaxed);
    if (!g->use_rccl) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int col = 0; col < kColumns; ++col) {
            HostFoldDD f;
            for (size_t i = 0; i < g->ctxs.size(); ++i) f.add(g->local[i] + (size_t)col * kRecordWords);
            uint64_t* out = g->host_finals + (size_t)col * 4;
            const double total = f.total();
            if (g->corrupt_member >= 0) f.isum ^= 0x5A5A5A5A5A5A5A5Aull;  // testing hook: finals that are wrong
            out[0] = f.isum;
            out[1] = f.icnt;
            memcpy(&out[2], &total, 8);
            out[3] = f.fcnt;
        }
        g->corrupt_member = -1;
        g->host_fold_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        ++g->host_fold_samples;
    }
    return MA_OK;
}

}  // namespace grp
}  // namespace ma

namespace {

// Peer capability between the members' devices, probed (and enabled) once. Returns how many ordered pairs of DISTINCT
// devices there are and how many of them are peer-capable; the pairs that are not are listed in `missing`.
void probe_peers(ma_group* g, int* out_pairs, int* out_capable, std::string* missing) {
    const size_t n = g->ctxs.size();
    g->peer.assign(n * n, 0);
    std::map<std::pair<int, int>, uint8_t> answered;
    int pairs = 0, capable = 0;
    for (size_t i = 0; i < n; ++i)
        for (size_t j = 0; j < n; ++j) {
            const int di = g->ctxs[i]->device, dj = g->ctxs[j]->device;
            if (di == dj) {
                g->peer[i * n + j] = 1;
                continue;
            }
            auto it = answered.find({di, dj});  // several members may share a device
            if (it != answered.end()) {
                g->peer[i * n + j] = it->second;
                continue;
            }
            ++pairs;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, di, dj) != hipSuccess) {
                (void)hipGetLastError();
                can = 0;
            }
            if (can) {
                hipError_t e = hipSetDevice(di);
                if (e == hipSuccess) e = hipDeviceEnablePeerAccess(dj, 0);
                if (e == hipErrorPeerAccessAlreadyEnabled) e = hipSuccess;
                (void)hipGetLastError();
                can = e == hipSuccess ? 1 : 0;
            }
            g->peer[i * n + j] = (uint8_t)can;
            answered[{di, dj}] = (uint8_t)can;
            if (can) ++capable;
            else if (missing->size() < 160) *missing += (missing->empty() ? "" : ",") + std::to_string(di) + "->" + std::to_string(dj);
        }
    *out_pairs = pairs;
    *out_capable = capable;
}

}  // namespace

namespace ma {
namespace grp {

// Sets up the exchange `flags` ask for on a group that has members but no exchange (creation, ma_group_rebuild_exchange),
// starts the issue threads, writes the note.
ma_status configure_exchange(ma_group* g, uint32_t flags) {
    ma_status st = MA_OK;
    char why[256] = "";
    if (flags & MA_GROUP_EXCHANGE_RCCL) {
        st = setup_rccl(g, (flags & MA_GROUP_EXCHANGE_OVERLAP) != 0,
                        (flags & MA_GROUP_SCAN_LANES) != 0 && (flags & MA_GROUP_EXCHANGE_OVERLAP) != 0);
        if (st != MA_OK && (flags & MA_GROUP_EXCHANGE_FALLBACK_HOST)) {
            snprintf(why, sizeof(why), "host fold instead of RCCL: %s", ma_last_error_string());
            release_exchange(g);  // whatever the attempt allocated; the members stay
            st = setup_host(g);
        }
    } else {
        st = setup_host(g);
    }
    MA_TRY(st);
    g->flags = flags;
    // Issue threads unless the caller (flag) or the environment asks for the calling-thread loop.
    const char* issue = getenv("MINARROW_HIP_GROUP_ISSUE");
    bool threads = (flags & MA_GROUP_ISSUE_CALLER) == 0;
    if (issue && strcmp(issue, "caller") == 0) threads = false;
    if (issue && strcmp(issue, "threads") == 0) threads = true;
    if (threads && !g->threads) start_workers(g);
    if (!threads && g->threads) stop_workers(g);
    // the note: why the exchange is what it is (if it is not what was asked for), then the peer matrix
    const char* handoff = "";
    if (g->overlap) {
        int signal = 0, words = 0;
        for (int k = 0; k < 2; ++k)
            for (uint64_t* w : g->stamp[k]) {
                if (!w) continue;
                ++words;
                if (ma_stamp_is_signal(w) == 1) ++signal;
            }
        handoff = words < 2 * (int)g->ctxs.size() ? " (hand-off: events; no waitable stamp words on this runtime)"
                  : signal == words               ? " (hand-off: stamps in signal memory)"
                                                  : " (hand-off: stamps in device memory)";
    }
    snprintf(g->note, sizeof(g->note), "%s%s%s; issue: %s%s%s", why, why[0] ? "; " : "", g->peer_note.c_str(),
             g->threads ? "one thread per member" : "calling thread", g->overlap ? "; exchange overlapped on side streams" : "", handoff);
    if (g->use_rccl) {
        const RcclApi* api = rccl();
        if (api && api->loopback) {  // first in the note: nothing measured through the double is a multi-GPU figure
            char rest[sizeof(g->note)];
            snprintf(rest, sizeof(rest), "%s", g->note);
            snprintf(g->note, sizeof(g->note), "REHEARSAL: collectives by the loopback double (MINARROW_HIP_RCCL_PATH), not RCCL; %s", rest);
        }
    }
    if (g->lanes2) strncat(g->note, "; two scan lanes per member, gated on the early stamp", sizeof(g->note) - strlen(g->note) - 1);
    else if (flags & MA_GROUP_SCAN_LANES)
        strncat(g->note, "; two scan lanes asked for but not set up (they need the overlapped RCCL exchange with stamps in device words)",
                sizeof(g->note) - strlen(g->note) - 1);
    return MA_OK;
}

const uint64_t* finals_of(const ma_group* g, size_t member, int32_t column) {
    const uint64_t* base = !g->use_rccl ? g->host_finals : (g->overlap && g->last == 1) ? g->finals1[member] : g->finals[member];
    return base + (size_t)column * 4;
}

}  // namespace grp
}  // namespace ma

extern "C" {

ma_status ma_group_create_ex(const int32_t* device_ordinals, int32_t n_members, uint32_t flags, ma_group** out_group) {
    MA_REQUIRE(out_group != nullptr, MA_ERR_INVALID_ARGUMENT, "out_group is NULL");
    *out_group = nullptr;
    MA_REQUIRE(device_ordinals != nullptr && n_members > 0 && n_members <= 1024, MA_ERR_INVALID_ARGUMENT,
               "a group needs 1..1024 members");
    ma_group* g = new ma_group();
    // A rehearsal (the loopback collective double stands in for RCCL and the members share a device): every stream that can
    // carry a collective — the members' own and their exchange streams — goes into the HIGH priority class, whose pool of
    // hardware queues those streams then have to themselves (GPU_MAX_HW_QUEUES >= members). A rank's collective kernel spins until
    // its peers' have run, and streams that share a hardware queue run in order; everything else (second scan lanes, copy
    // streams) may share queues, because it only ever waits for work that was enqueued before it. Few queues matter as much as
    // distinct ones: a device runs 23 of them at a time (tests/loopback_rccl/selfcheck.cpp `slots`), one more and dependent
    // work crawls from time slice to time slice.
    const RcclApi* double_api = (flags & MA_GROUP_EXCHANGE_RCCL) ? rccl() : nullptr;
    g->carrier_class = (double_api && double_api->loopback) ? 1 : 0;
    for (int32_t i = 0; i < n_members; ++i) {
        ma_ctx* c = nullptr;
        ma_status st = create_ctx_in_class(device_ordinals[i], g->carrier_class, &c);
        if (st == MA_OK) st = ma_ctx_set_async(c, 1);  // members only ever enqueue; the group synchronises them
        if (st != MA_OK) {
            if (c) ma_ctx_destroy(c);
            destroy_members(g);
            delete g;
            return st;
        }
        g->ctxs.push_back(c);
    }
    for (ma_ctx* c : g->ctxs) g->home.push_back(c->device);
    // Peer capability first (RCCL enables the same pairs itself and tolerates "already enabled").
    int pairs = 0, capable = 0;
    std::string missing;
    probe_peers(g, &pairs, &capable, &missing);
    char peers[320];
    if (pairs == 0)
        snprintf(peers, sizeof(peers), "peer access: n/a (one device)");
    else if (capable == pairs)
        snprintf(peers, sizeof(peers), "peer access: %d/%d ordered device pairs", capable, pairs);
    else
        snprintf(peers, sizeof(peers), "peer access: %d/%d ordered device pairs (not peer-capable: %s)", capable, pairs,
                 missing.c_str());
    g->peer_note = peers;
    const ma_status st = configure_exchange(g, flags);
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
    {   // Never an unbounded wait: releasing the exchange synchronizes the members' streams, and one that is still held (a
        // collective whose peer never came) would keep that for good. 10 s for whatever is in flight; past them the communicators
        // are aborted, and what a stream that STILL has not run empty holds goes with the process (destroy_members).
        std::lock_guard<std::recursive_mutex> lock(group->mu);
        if (!group->broken) (void)synchronize_for_locked(group, destroy_wait_ms());
    }
    destroy_members(group);
    delete group;
}

int32_t ma_group_size(ma_group* group) { return group ? (int32_t)group->ctxs.size() : 0; }

ma_ctx* ma_group_ctx(ma_group* group, int32_t index) {
    if (!group || index < 0 || (size_t)index >= group->ctxs.size()) return nullptr;
    return group->ctxs[(size_t)index];
}

int32_t ma_group_exchange_kind(ma_group* group) { return group && group->use_rccl ? 1 : 0; }
int32_t ma_group_issue_kind(ma_group* group) { return group && group->threads ? 1 : 0; }
int32_t ma_group_peer_access(ma_group* group, int32_t from_member, int32_t to_member) {
    if (!group) return -1;
    const size_t G = group->ctxs.size();
    if (from_member < 0 || to_member < 0 || (size_t)from_member >= G || (size_t)to_member >= G) return -1;
    return group->peer[(size_t)from_member * G + (size_t)to_member] ? 1 : 0;
}
const char* ma_group_exchange_note(ma_group* group) { return group ? group->note : ""; }

ma_status ma_group_test_set_member_device(ma_group* group, int32_t member, int32_t hip_device, int32_t peer_capable) {
    MA_REQUIRE(group != nullptr, MA_ERR_INVALID_ARGUMENT, "group is NULL");
    MA_TRY(test_hooks_enabled());
    std::lock_guard<std::recursive_mutex> lock(group->mu);
    const size_t G = group->ctxs.size();
    MA_REQUIRE(member >= 0 && (size_t)member < G, MA_ERR_INVALID_ARGUMENT, "member %d out of range", member);
    group->home[(size_t)member] = hip_device;
    for (size_t j = 0; j < G; ++j) {
        if (j == (size_t)member) continue;
        const uint8_t can = peer_capable ? 1 : 0;
        group->peer[(size_t)member * G + j] = group->peer[j * G + (size_t)member] = can;
    }
    return MA_OK;
}

ma_status ma_group_test_fail_next_exchange(ma_group* group, int32_t member) {
    MA_REQUIRE(group != nullptr, MA_ERR_INVALID_ARGUMENT, "group is NULL");
    MA_TRY(test_hooks_enabled());
    std::lock_guard<std::recursive_mutex> lock(group->mu);
    MA_REQUIRE(member >= 0 && (size_t)member < group->ctxs.size(), MA_ERR_INVALID_ARGUMENT, "member %d out of range", member);
    group->fail_member = member;
    return MA_OK;
}

ma_status ma_group_enqueue_sum_i64(ma_group* group, int32_t column, const int64_t* const* chunk_data,
                                   const size_t* chunk_lens, const uint8_t* const* chunk_masks,
                                   const size_t* chunk_mask_offsets) {
    MA_REQUIRE(group && chunk_data && chunk_lens, MA_ERR_INVALID_ARGUMENT, "NULL argument");
    return enqueue_sum_members(group, column, (const void* const*)chunk_data, chunk_lens, chunk_masks, [&](size_t i, ma_ctx* c, uint64_t* rec) {
        return ma_i64_sum(c, chunk_data[i], chunk_lens[i], chunk_masks ? chunk_masks[i] : nullptr,
                          chunk_mask_offsets ? chunk_mask_offsets[i] : 0, -1, (int64_t*)&rec[0], &rec[1]);
    });
}

ma_status ma_group_enqueue_sum_f64(ma_group* group, int32_t column, const double* const* chunk_data,
                                   const size_t* chunk_lens, const uint8_t* const* chunk_masks,
                                   const size_t* chunk_mask_offsets) {
    MA_REQUIRE(group && chunk_data && chunk_lens, MA_ERR_INVALID_ARGUMENT, "NULL argument");
    return enqueue_sum_members(group, column, (const void* const*)chunk_data, chunk_lens, chunk_masks, [&](size_t i, ma_ctx* c, uint64_t* rec) {
        return ma_f64_sum_dd(c, chunk_data[i], chunk_lens[i], chunk_masks ? chunk_masks[i] : nullptr,
                             chunk_mask_offsets ? chunk_mask_offsets[i] : 0, -1, (double*)&rec[2], (double*)&rec[3], &rec[4]);
    });
}

// The partitioned step in ONE launch per member (round 4): the sums of up to MA_FUSED_MAX_COLUMNS long 8-byte columns whose
// rows are partitioned over the group — the reference's bench runs its i64 and its f64 loop back to back over the same
// partition (benches/benchmark_parallel_simd.rs:99-125); a table's per-column reduce does the same per column. Member i
// scans chunk i of EVERY listed column with one ma_sum_fused launch and writes the columns' record slots of the current
// set; ma_group_exchange / ma_group_result are unchanged. Per step and member this saves a launch's fixed cost per extra
// column (~3.3 us of 137 us per 125 M-row chunk: profiles/r04_probe_epilogue.jsonl).
ma_status ma_group_enqueue_sum_table(ma_group* group, int32_t n_cols, const int32_t* columns, const int32_t* format_codes,
                                     const void* const* const* chunk_data, const size_t* const* chunk_lens,
                                     const uint8_t* const* const* chunk_masks, const size_t* const* chunk_mask_offsets) {
    MA_REQUIRE(group != nullptr, MA_ERR_INVALID_ARGUMENT, "group is NULL");
    MA_REQUIRE(n_cols >= 1 && n_cols <= MA_FUSED_MAX_COLUMNS, MA_ERR_INVALID_ARGUMENT, "ma_group_enqueue_sum_table takes 1..%d columns",
               MA_FUSED_MAX_COLUMNS);
    MA_REQUIRE(columns && format_codes && chunk_data && chunk_lens, MA_ERR_INVALID_ARGUMENT, "NULL argument");
    for (int32_t k = 0; k < n_cols; ++k) {
        MA_REQUIRE(columns[k] >= 0 && columns[k] < kColumns, MA_ERR_INVALID_ARGUMENT, "column %d out of range [0,%d)", columns[k], kColumns);
        MA_REQUIRE(format_codes[k] == 'l' || format_codes[k] == 'L' || format_codes[k] == 'g', MA_ERR_UNSUPPORTED,
                   "format '%c' (the fused step takes the 8-byte formats l, L and g)", (char)format_codes[k]);
        MA_REQUIRE(chunk_data[k] && chunk_lens[k], MA_ERR_INVALID_ARGUMENT, "column %d: NULL chunk table", k);
        for (int32_t j = 0; j < k; ++j)  // two columns may share a record only as its integer half and its float half
            MA_REQUIRE(columns[j] != columns[k] || (format_codes[j] == 'g') != (format_codes[k] == 'g'), MA_ERR_INVALID_ARGUMENT,
                       "columns %d and %d would write the same slots of record %d", j, k, columns[k]);
    }
    std::lock_guard<std::recursive_mutex> lock(group->mu);
    MA_REQUIRE(!group->broken, MA_ERR_DEVICE, "%s", kBrokenMessage);
    DeviceLookup lookup;
    for (int32_t k = 0; k < n_cols; ++k)
        for (size_t i = 0; i < group->ctxs.size(); ++i) {
            if (chunk_lens[k][i] == 0) continue;
            MA_TRY(require_resident(group, lookup, i, chunk_data[k][i], "data", i));
            if (chunk_masks && chunk_masks[k]) MA_TRY(require_resident(group, lookup, i, chunk_masks[k][i], "validity bitmap", i));
        }
    // overlapped exchanges: this launch stamps the set's hand-off word behind its results (see ma_group::stamp)
    const int cur_set = group->overlap ? group->cur : 0;
    for (int32_t k = 0; k < n_cols; ++k) group->enq_mask[cur_set] |= 1u << columns[k];  // only once nothing can refuse the call
    bool stamped = group->overlap && group->stamp[cur_set].size() == group->ctxs.size();
    for (size_t i = 0; stamped && i < group->ctxs.size(); ++i) stamped = group->stamp[cur_set][i] != nullptr;
    const uint64_t seq = stamped ? ++group->stamp_seq[cur_set] : 0;
    group->stamp_ok[cur_set] = stamped;
    // two scan lanes: this step runs on the lane of its record set; when the call before it was a stamped step on the OTHER lane
    // it starts when that step has begun to drain (its early stamp: FusedArgs::early_word), not beside its whole scan
    const bool lanes = group->lanes2 && group->lanes_on;
    const bool gate = lanes && stamped && group->prev_set >= 0 && group->prev_set != cur_set;
    const int prev_set = group->prev_set;
    const uint64_t prev_seq = group->prev_seq;
    const int mark_from = group->mark_from, mark_to = group->mark_to;
    group->mark_from = group->mark_to = -1;
    if (mark_from >= 0) {
        if (group->mark_lane.size() < (size_t)MA_CTX_MAX_MARKS) group->mark_lane.assign(MA_CTX_MAX_MARKS, 0);
        group->mark_lane[(size_t)mark_from] = group->mark_lane[(size_t)mark_to] = (uint8_t)((group->lanes2 && group->lanes_on && cur_set == 1) ? 1 : 0);
    }
    const ma_status st = run_on_members(group, [&](size_t i) -> ma_status {
        uint64_t* set = (group->overlap && group->cur == 1) ? group->local1[i] : group->local[i];
        ma_ctx* sc = scan_ctx(group, cur_set, i);
        // The member's call counter ONCE, in front of everything: what it shows beyond seen_calls[i] is foreign work. Entries this
        // step makes itself on the member's context are added to the snapshot below; one another host thread makes meanwhile is
        // not, so the next lane-1 step is ordered behind it.
        const uint64_t now = lanes ? group->ctxs[i]->calls.load(std::memory_order_relaxed) : 0;
        const uint64_t own0 = entries_by_this_thread();
        if (lanes) {
            if (sc != group->ctxs[i]) {  // the tuning knobs follow the member's context
                sc->variant = group->ctxs[i]->variant;
                sc->blocks_per_cu = group->ctxs[i]->blocks_per_cu;
                sc->grid_override = group->ctxs[i]->grid_override;
                MA_TRY(order_lane_if_foreign(group, i, now));
            }
            if (gate) {
                MA_HIP(hipSetDevice(sc->device));
                MA_HIP(hipStreamWaitValue64(sc->stream, group->stamp[prev_set][i] + 1, prev_seq, hipStreamWaitValueGte, ~(uint64_t)0));
            }
        }
        ma_fused_column cols[MA_FUSED_MAX_COLUMNS];
        for (int32_t k = 0; k < n_cols; ++k) {
            uint64_t* rec = set + (size_t)columns[k] * kRecordWords;
            cols[k].data = chunk_data[k][i];
            cols[k].n = chunk_lens[k][i];
            cols[k].mask_bits = (chunk_masks && chunk_masks[k]) ? chunk_masks[k][i] : nullptr;
            cols[k].mask_bit_offset = (chunk_mask_offsets && chunk_mask_offsets[k]) ? chunk_mask_offsets[k][i] : 0;
            cols[k].null_count = -1;
            cols[k].format_code = format_codes[k];
            cols[k].reserved = 0;
            cols[k].out = format_codes[k] == 'g' ? rec + 2 : rec;
        }
        if (mark_from >= 0) MA_TRY(ma_ctx_mark(sc, mark_from));  // behind the lane's waits: the marks bracket the scan alone
        MA_TRY(sum_fused_impl(sc, (size_t)n_cols, cols, stamped ? group->stamp[cur_set][i] : nullptr, seq, false,
                              (lanes && stamped) ? group->stamp[cur_set][i] + 1 : nullptr));
        if (mark_to >= 0) MA_TRY(ma_ctx_mark(sc, mark_to));
        if (lanes) group->seen_calls[i] = now + (sc == group->ctxs[i] ? entries_by_this_thread() - own0 : 0);
        return MA_OK;
    });
    // A member that refused the launch (a misaligned pointer, say) never stamps `seq`: an exchange waiting for that value on
    // its side stream would wait for good. The event path orders behind whatever did reach the streams.
    if (st != MA_OK) group->stamp_ok[cur_set] = false;
    if (lanes) {
        group->prev_set = (st == MA_OK && stamped) ? cur_set : -1;
        group->prev_seq = seq;
    }
    return st;
}

// The sum of ONE column held as MANY chunks spread over the group — a SuperArray, or one column of a SuperTable's batches
// (BASELINE config 5 at the reference's own batch sizes: 122 000 batches of 8192 rows per 10^9 rows, not one per GPU).
// Chunk i belongs to member i % G (like the fan-out below); every member sums ITS chunks in one ma_sum_chunks pass
// (a wave per chunk, in-place descriptors) into its record of `column` — integer formats into the integer slots, float
// formats as a (hi, lo) pair into the float slots —, concurrently on the members' issue threads. ma_group_exchange and
// ma_group_result then give the job's total as for one chunk per member: wrapping, or within 1 ULP.
ma_status ma_group_enqueue_sum_chunks(ma_group* group, int32_t column, int32_t format_code, size_t n_chunks,
                                      const void* const* chunk_data, const size_t* chunk_lens,
                                      const uint8_t* const* chunk_masks, const size_t* chunk_mask_offsets) {
    MA_REQUIRE(group != nullptr, MA_ERR_INVALID_ARGUMENT, "group is NULL");
    MA_REQUIRE(column >= 0 && column < kColumns, MA_ERR_INVALID_ARGUMENT, "column %d out of range [0,%d)", column, kColumns);
    MA_REQUIRE(n_chunks == 0 || (chunk_data && chunk_lens), MA_ERR_INVALID_ARGUMENT, "NULL chunk table");
    MA_REQUIRE(format_code > 0 && strchr("cCsSiIlLfg", (char)format_code) != nullptr, MA_ERR_UNSUPPORTED,
               "unsupported element format '%c' (numeric primitives only)", (char)format_code);
    const bool is_float = format_code == 'f' || format_code == 'g';
    std::lock_guard<std::recursive_mutex> lock(group->mu);
    MA_REQUIRE(!group->broken, MA_ERR_DEVICE, "%s", kBrokenMessage);
    const size_t G = group->ctxs.size();
    DeviceLookup lookup;
    for (size_t i = 0; i < n_chunks; ++i) {
        if (chunk_lens[i] == 0) continue;
        MA_TRY(require_resident(group, lookup, i % G, chunk_data[i], "data", i));
        if (chunk_masks) MA_TRY(require_resident(group, lookup, i % G, chunk_masks[i], "validity bitmap", i));
    }
    const int cur_set = group->overlap ? group->cur : 0;
    group->enq_mask[cur_set] |= 1u << column;
    group->stamp_ok[cur_set] = false;
    group->prev_set = -1;
    return run_on_members(group, [&](size_t m) -> ma_status {
        if (cur_set == 1) MA_TRY(order_lane_if_foreign(group, m, group->ctxs[m]->calls.load(std::memory_order_relaxed)));
        std::vector<const void*> d;
        std::vector<size_t> n, o;
        std::vector<const uint8_t*> k;
        for (size_t i = m; i < n_chunks; i += G) {
            d.push_back(chunk_data[i]);
            n.push_back(chunk_lens[i]);
            k.push_back(chunk_masks ? chunk_masks[i] : nullptr);
            o.push_back(chunk_mask_offsets ? chunk_mask_offsets[i] : 0);
        }
        uint64_t* set = (group->overlap && group->cur == 1) ? group->local1[m] : group->local[m];
        uint64_t* rec = set + (size_t)column * kRecordWords;
        ma_ctx* c = scan_ctx(group, cur_set, m);
        return is_float ? sum_chunks_dd(c, format_code, d.size(), d.data(), n.data(), chunk_masks ? k.data() : nullptr, o.data(),
                                        (double*)&rec[2], (double*)&rec[3], nullptr, &rec[4])
                        : sum_chunks_dd(c, format_code, d.size(), d.data(), n.data(), chunk_masks ? k.data() : nullptr, o.data(),
                                        nullptr, nullptr, (int64_t*)&rec[0], &rec[1]);
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
    std::lock_guard<std::recursive_mutex> lock(group->mu);
    const size_t G = group->ctxs.size();
    // A device-resident chunk must live on its member's GPU: the kernels address it directly. A SuperArray's thousands
    // of chunk pointers run through a handful of allocations: each allocation is asked about once.
    DeviceLookup lookup;
    auto device_of = [&](const void* p) { return lookup.device_of(p); };
    for (size_t i = 0; i < n_chunks; ++i) {
        if (lhs_lens[i] == 0) continue;
        const int want = group->home[i % G];
        const void* ptrs[6] = {lhs_data[i], rhs_data[i], out_data[i], lhs_masks ? lhs_masks[i] : nullptr,
                               rhs_masks ? rhs_masks[i] : nullptr, out_masks ? out_masks[i] : nullptr};
        for (const void* p : ptrs) {
            const int dev = device_of(p);
            MA_REQUIRE(dev < 0 || dev == want, MA_ERR_INVALID_ARGUMENT,
                       "chunk %zu belongs to member %zu (device %d) but one of its buffers is resident on device %d", i, i % G,
                       want, dev);
        }
    }
    // every member gathers its own chunk pairs (i, i + G, ...) and issues its ONE launch — descriptor table included —
    // on its own thread: the table of a RechunkStrategy-sized column (10^5 pairs) is built G-way parallel
    return run_on_members(group, [&](size_t m) -> ma_status {
        if (m >= n_chunks) return MA_OK;
        std::vector<const void*> l, r;
        std::vector<void*> o;
        std::vector<const uint8_t*> lm, rm;
        std::vector<uint8_t*> om;
        std::vector<size_t> ll, rl;
        const size_t mine = (n_chunks - m + G - 1) / G;
        l.reserve(mine); r.reserve(mine); o.reserve(mine); lm.reserve(mine); rm.reserve(mine); om.reserve(mine);
        ll.reserve(mine); rl.reserve(mine);
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
        std::vector<int32_t> has(l.size(), 0);
        MA_TRY(ma_route_super_array_broadcast(group->ctxs[m], format_code, op, l.size(), l.data(), ll.data(), lm.data(), r.data(),
                                              rl.data(), rm.data(), member_mask_overrides ? member_mask_overrides[m] : nullptr,
                                              o.data(), om.data(), has.data()));
        if (out_has_mask)
            for (size_t j = 0; j < has.size(); ++j) out_has_mask[m + j * G] = has[j];
        return MA_OK;
    });
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
    std::lock_guard<std::recursive_mutex> lock(group->mu);
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
    const int dest_home = group->home[(size_t)dest_member];
    MA_REQUIRE(device_of(out_data) == dest_home, MA_ERR_INVALID_ARGUMENT,
               "out_data must be device memory of member %d (device %d)", dest_member, dest_home);
    MA_REQUIRE(!has_mask || device_of(out_mask) == dest_home, MA_ERR_INVALID_ARGUMENT,
               "out_mask must be device memory of member %d (device %d)", dest_member, dest_home);
    for (size_t i = 0; i < n_chunks; ++i) {
        if (!chunk_lens[i]) continue;
        const int want = group->home[i % G];
        MA_REQUIRE(device_of(chunk_data[i]) == want, MA_ERR_INVALID_ARGUMENT,
                   "chunk %zu belongs to member %zu: its data must be device memory of device %d", i, i % G, want);
        MA_REQUIRE(!(chunk_masks && chunk_masks[i]) || device_of(chunk_masks[i]) == want, MA_ERR_INVALID_ARGUMENT,
                   "chunk %zu belongs to member %zu: its bitmap must be device memory of device %d", i, i % G, want);
        // the copy is a peer write from the owner into the destination's HBM over xGMI: refused — not left to whatever the
        // runtime does without a link (a staged copy through the host at PCIe rate, or a fault) — when the pair has none
        MA_REQUIRE(group->peer[(i % G) * G + (size_t)dest_member] != 0, MA_ERR_UNSUPPORTED,
                   "chunk %zu: device %d (member %zu) has no peer access to device %d (member %d): consolidate this column "
                   "on a member whose device the owners can reach, or gather through the host",
                   i, want, i % G, dest_home, dest_member);
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

ma_status ma_group_mark_next_scan(ma_group* group, int32_t from_index, int32_t to_index) {
    MA_REQUIRE(group != nullptr, MA_ERR_INVALID_ARGUMENT, "group is NULL");
    MA_REQUIRE(from_index >= 0 && from_index < MA_CTX_MAX_MARKS && to_index >= 0 && to_index < MA_CTX_MAX_MARKS && from_index != to_index,
               MA_ERR_INVALID_ARGUMENT, "marks %d, %d (two different indices in [0,%d))", from_index, to_index, MA_CTX_MAX_MARKS);
    std::lock_guard<std::recursive_mutex> lock(group->mu);
    group->mark_from = from_index;
    group->mark_to = to_index;
    return MA_OK;
}

ma_status ma_group_mark_elapsed_ms(ma_group* group, int32_t member, int32_t from_index, int32_t to_index, float* out_ms) {
    MA_REQUIRE(group != nullptr && out_ms != nullptr, MA_ERR_INVALID_ARGUMENT, "group or out_ms is NULL");
    MA_REQUIRE(member >= 0 && (size_t)member < group->ctxs.size(), MA_ERR_INVALID_ARGUMENT, "member %d out of range", member);
    MA_REQUIRE(from_index >= 0 && from_index < MA_CTX_MAX_MARKS && to_index >= 0 && to_index < MA_CTX_MAX_MARKS, MA_ERR_INVALID_ARGUMENT,
               "mark index out of range");
    std::lock_guard<std::recursive_mutex> lock(group->mu);
    MA_REQUIRE(group->mark_lane.size() == (size_t)MA_CTX_MAX_MARKS, MA_ERR_INVALID_ARGUMENT, "no marks have been recorded (ma_group_mark_next_scan)");
    MA_REQUIRE(group->mark_lane[(size_t)from_index] != 255 && group->mark_lane[(size_t)to_index] != 255, MA_ERR_INVALID_ARGUMENT,
               "these marks were recorded on second scan lanes that ma_group_set_scan_lanes(2) has replaced since");
    MA_REQUIRE(group->mark_lane[(size_t)from_index] == group->mark_lane[(size_t)to_index], MA_ERR_INVALID_ARGUMENT,
               "marks %d and %d were recorded on different scan lanes", from_index, to_index);
    ma_ctx* c = (group->mark_lane[(size_t)from_index] && group->lanes2) ? group->scan2[(size_t)member] : group->ctxs[(size_t)member];
    return ma_ctx_mark_elapsed_ms(c, from_index, to_index, out_ms);
}

ma_status ma_group_exchange(ma_group* group) {
    MA_REQUIRE(group != nullptr, MA_ERR_INVALID_ARGUMENT, "group is NULL");
    std::lock_guard<std::recursive_mutex> lock(group->mu);
    return exchange_locked(group);
}

ma_status ma_group_exchange_stats(ma_group* group, double* out_all_gather_us, double* out_fold_us, int32_t* out_samples,
                                  int32_t* out_rccl_ranks) {
    MA_REQUIRE(group != nullptr, MA_ERR_INVALID_ARGUMENT, "group is NULL");
    std::lock_guard<std::recursive_mutex> lock(group->mu);
    if (!group->use_rccl) {  // host exchange: the records land in pinned memory by themselves, the fold runs in synchronize
        if (out_all_gather_us) *out_all_gather_us = 0.0;
        if (out_fold_us) *out_fold_us = group->host_fold_samples ? group->host_fold_us / group->host_fold_samples : 0.0;
        if (out_samples) *out_samples = group->host_fold_samples;
        if (out_rccl_ranks) *out_rccl_ranks = 0;
        group->host_fold_us = 0.0;
        group->host_fold_samples = 0;
        return MA_OK;
    }
    MA_REQUIRE(!group->broken, MA_ERR_DEVICE, "%s", kBrokenMessage);  // its sampled events may never fire
    MA_HIP(hipSetDevice(group->ctxs[0]->device));
    group->timer.report(out_all_gather_us, out_fold_us, out_samples);
    if (out_rccl_ranks) {
        int n = 0;
        const RcclApi* api = rccl();
        if (!api || !api->CommCount || group->comms.empty() || !group->comms[0] || api->CommCount(group->comms[0], &n) != ncclSuccess) n = 0;
        *out_rccl_ranks = n;
    }
    return MA_OK;
}

ma_status ma_group_synchronize(ma_group* group) {
    MA_REQUIRE(group != nullptr, MA_ERR_INVALID_ARGUMENT, "group is NULL");
    std::lock_guard<std::recursive_mutex> lock(group->mu);
    return synchronize_locked(group);
}

ma_status ma_group_member_result(ma_group* group, int32_t member, int32_t column, int64_t* out_int_sum,
                                 uint64_t* out_int_count, double* out_f64_sum, uint64_t* out_f64_count) {
    MA_REQUIRE(group != nullptr, MA_ERR_INVALID_ARGUMENT, "group is NULL");
    MA_REQUIRE(member >= 0 && (size_t)member < group->ctxs.size(), MA_ERR_INVALID_ARGUMENT, "member %d out of range", member);
    MA_REQUIRE(column >= 0 && column < kColumns, MA_ERR_INVALID_ARGUMENT, "column %d out of range [0,%d)", column, kColumns);
    std::lock_guard<std::recursive_mutex> lock(group->mu);
    MA_REQUIRE(!group->broken, MA_ERR_DEVICE, "%s", kBrokenMessage);
    MA_REQUIRE(!(group->overlap && group->use_rccl) || !group->set_used[group->last] ||
                   (group->exchanged_mask[group->last] >> column) & 1u,
               MA_ERR_INVALID_ARGUMENT,
               "column %d was not enqueued in the step of the last exchange: with overlapped exchanges the group alternates "
               "between two record sets, and a column read after an exchange must have been enqueued in that step", column);
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
    MA_REQUIRE(group != nullptr, MA_ERR_INVALID_ARGUMENT, "group is NULL");
    // one critical section: two calling threads share column 0's records, and the result must be THIS call's
    std::lock_guard<std::recursive_mutex> lock(group->mu);
    MA_TRY(ma_group_enqueue_sum_i64(group, 0, chunk_data, chunk_lens, chunk_masks, chunk_mask_offsets));
    MA_TRY(ma_group_exchange(group));
    MA_TRY(ma_group_synchronize(group));
    return ma_group_result(group, 0, out_sum, out_valid_count, nullptr, nullptr);
}

ma_status ma_group_sum_f64(ma_group* group, const double* const* chunk_data, const size_t* chunk_lens,
                           const uint8_t* const* chunk_masks, const size_t* chunk_mask_offsets, double* out_sum,
                           uint64_t* out_valid_count) {
    MA_REQUIRE(group != nullptr, MA_ERR_INVALID_ARGUMENT, "group is NULL");
    std::lock_guard<std::recursive_mutex> lock(group->mu);
    MA_TRY(ma_group_enqueue_sum_f64(group, 0, chunk_data, chunk_lens, chunk_masks, chunk_mask_offsets));
    MA_TRY(ma_group_exchange(group));
    MA_TRY(ma_group_synchronize(group));
    return ma_group_result(group, 0, nullptr, nullptr, out_sum, out_valid_count);
}

}  // extern "C"
