// RCCL (the ROCm collective library: NCCL's API over xGMI) as libminarrow_hip.so uses it: the exchange that ends a
// row-chunk partitioned reduction — the `.sum()` over per-chunk partials of rayon_simd_sum_*
// (benches/benchmark_parallel_simd.rs:81-98) once the chunks live on several GPUs.
//
// librccl.so.1 is opened on first use (dlopen), not linked: a host that drives one GPU never pays for loading it
// (the library is several hundred MB of device code), and inside a process that already holds an RCCL — PyTorch
// ships its own build under the same SONAME — the handle resolves to that one, so both share one runtime.
// Internal header; the C ABI on top of it is ma_comm_* and ma_group_* (include/minarrow_hip.h).
#pragma once

#include <rccl/rccl.h>

#include <functional>

#include "ma_common.hpp"

namespace ma {

struct RcclApi {
    ncclResult_t (*GetVersion)(int*) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    const char* path = "";  // which library name resolved ("REHEARSAL (...): <path>" for the loopback double)
    bool named = false;     // MINARROW_HIP_RCCL_PATH chose it
    bool loopback = false;  // tests/loopback_rccl's stand-in (exports ncclLoopbackDoubleInfo): ranks may share a device
};

// The process-wide RCCL entry points, or nullptr (with the thread's error string set) when the library cannot be
// opened. Thread safe; the outcome of the first attempt is kept.
const RcclApi* rccl();

ma_status rccl_fail(ncclResult_t r, const char* what, const char* file, int line);

// MINARROW_HIP_GUARD_LOG=1: the bounded waits, aborts and rebuilds of ma_group_* / ma_comm_* say on stderr what they do, step
// by step with a timestamp — the trace a first run on a multi-GPU node leaves behind when something does not return.
void guard_log(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

// ma_ctx.hip: stores by the host when `stamp` (from ma_stamp_alloc) is host memory — signal memory is —; false otherwise.
bool stamp_host_store(uint64_t* stamp, uint64_t value);
// ma_ctx.hip: ma_stamp_alloc with the kind of memory chosen (signal memory: the host can release a wait on it by a store).
ma_status stamp_alloc_kind(ma_ctx* ctx, uint64_t** out_stamp, bool want_signal);
// ma_group_guard.hip: fn() on a helper thread, waited for at most timeout_ms; false when it has not returned (the thread is
// left behind). For runtime calls that are documented to return but wait on the GPU inside (ncclCommAbort).
bool call_bounded(const std::function<void()>& fn, double timeout_ms);

// ma_group_guard.hip: how long ma_group_destroy / ma_comm_destroy wait for work still in flight before they abort (10 s;
// MINARROW_HIP_DESTROY_WAIT_MS).
double destroy_wait_ms();

// ma_group_guard.hip: one thread stores `value` to `*stamp` with a system-scope release, as the fused scan's final thread does.
hipError_t launch_stamp_store(hipStream_t stream, uint64_t* stamp, uint64_t value);

// Where an exchange's time goes (ma_group_exchange_stats / ma_comm_exchange_stats): every 4th exchange carries three HIP
// events on the stream it runs on — in front of the all-gather, behind it, behind the fold — so that the first scaling run
// on a multi-GPU node explains itself (all-gather latency over xGMI vs. the fold kernel). An event costs the stream a few
// microseconds, hence the sampling; on an overlapped exchange they sit on the side stream, off the scans' path.
struct ExchangeTimer {
    static constexpr int kRing = 8, kEvery = 4;
    hipEvent_t ev[kRing][3] = {};
    bool pending[kRing] = {};
    uint64_t calls = 0;
    int next = 0;
    double sum_gather_us = 0.0, sum_fold_us = 0.0;
    int samples = 0;

    // -1: this exchange is not sampled; else the ring slot whose first event was recorded on `s`
    int begin(hipStream_t s) {
        if ((calls++ % kEvery) != 0) return -1;
        const int k = next;
        if (pending[k]) {  // never WAIT here: a host that runs more than kRing samples ahead of its GPU would be throttled to the
            harvest_slot(k, false);  // GPU's pace by its own instrumentation (it was, in round 4's first form: 240 us of "issue
            if (pending[k]) return -1;  // time" per step that were this wait) — the sample is skipped instead
        }
        next = (next + 1) % kRing;
        for (int j = 0; j < 3; ++j)
            if (!ev[k][j] && hipEventCreate(&ev[k][j]) != hipSuccess) {
                (void)hipGetLastError();
                return -1;
            }
        if (hipEventRecord(ev[k][0], s) != hipSuccess) {
            (void)hipGetLastError();
            return -1;
        }
        return k;
    }
    void mark(int k, int which, hipStream_t s) {
        if (k < 0) return;
        if (hipEventRecord(ev[k][which], s) != hipSuccess) (void)hipGetLastError();
        if (which == 2) pending[k] = true;
    }
    void harvest_slot(int k, bool wait) {
        if (!pending[k]) return;
        if (wait ? hipEventSynchronize(ev[k][2]) != hipSuccess : hipEventQuery(ev[k][2]) != hipSuccess) {
            (void)hipGetLastError();
            if (!wait) return;
        }
        float a = 0, b = 0;
        if (hipEventElapsedTime(&a, ev[k][0], ev[k][1]) == hipSuccess && hipEventElapsedTime(&b, ev[k][1], ev[k][2]) == hipSuccess) {
            sum_gather_us += (double)a * 1e3;
            sum_fold_us += (double)b * 1e3;
            ++samples;
        } else {
            (void)hipGetLastError();
        }
        pending[k] = false;
    }
    // averages over the samples taken since the last report (which waits for the sampled exchanges still in flight)
    void report(double* gather_us, double* fold_us, int32_t* n) {
        for (int k = 0; k < kRing; ++k) harvest_slot(k, true);
        if (gather_us) *gather_us = samples ? sum_gather_us / samples : 0.0;
        if (fold_us) *fold_us = samples ? sum_fold_us / samples : 0.0;
        if (n) *n = samples;
        sum_gather_us = sum_fold_us = 0.0;
        samples = 0;
    }
    void destroy() {
        for (auto& slot : ev)
            for (hipEvent_t& e : slot)
                if (e) {
                    (void)hipEventDestroy(e);
                    e = nullptr;
                }
    }
};

}  // namespace ma

#define MA_NCCL(api, call)                                                          \
    do {                                                                            \
        ncclResult_t _r = (api)->call;                                              \
        if (_r != ncclSuccess) return ::ma::rccl_fail(_r, #call, __FILE__, __LINE__); \
    } while (0)
