// Back-to-back sums of mid-size columns on ONE GPU — the reference's hot loop, `for _ in 0..N { sum(&arr) }` over an
// IntegerArray / FloatArray (benches/hotloop_benchmark_avg_std.rs:48-62: ITERATIONS passes, an i64 and an f64 sum each; the pass itself: hotloop_benchmark_std.rs:109-127) — as a pipeline. A launch costs ~3.3 us beyond its
// bytes (ramp + hand-off) and one stream starts scan k + 1 only behind the LAST workgroup of scan k:
// 2^24-row sums run at 0.70-0.73 of peak that way, the 125 M-row step of an 8-way partition at 0.885. ma_scan_lanes_* puts
// consecutive fused scans on two streams of the device in turn and starts each one when the scan in front of it has begun
// to DRAIN (its early stamp: FusedArgs::early_word, ma_reduce_fused.hip) — the ramp of scan k + 1 runs under the stragglers
// of scan k, not beside its whole length. It is what ma_group_* does per member under MA_GROUP_SCAN_LANES (ma_group.hip),
// for a host that drives one GPU without a group.
#include <algorithm>
#include <chrono>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>

#include "minarrow_hip_testing.h"

#include "ma_common.hpp"
#include "ma_rccl.hpp"  // stamp_alloc_kind

namespace ma {
ma_status create_ctx_in_class(int32_t device_ordinal, int cls, ma_ctx** out);
ma_status sum_fused_impl(ma_ctx* ctx, size_t n_cols, const ma_fused_column* cols, uint64_t* stamp, uint64_t stamp_value,
                         bool as_partials = false, uint64_t* early_stamp = nullptr);
ma_status sum_stamped_any(ma_ctx* ctx, int32_t format_code, const void* data, size_t n, const uint8_t* mask_bits,
                          size_t mask_bit_offset, int64_t null_count, void* out_sum, double* out_lo, uint64_t* out_cnt,
                          uint64_t* stamp, uint64_t stamp_value, uint64_t* early_stamp);  // ma_reduce.hip
}  // namespace ma

using namespace ma;

struct ma_scan_lanes {
    std::mutex mu;
    ma_ctx* ctx = nullptr;           // the caller's context: lane 0 is ITS stream (borrowed)
    ma_ctx* lane = nullptr;          // lane 1: a context of the same device with its own stream, partials and tickets (owned)
    uint64_t* stamp[2] = {nullptr, nullptr};  // per lane a 64-byte line: word 0 the final stamp, word 1 the early one
    uint64_t seq[2] = {0, 0};
    int turn = 0;                    // the lane the next scan goes to
    int prev = -1;                   // the lane of the scan before it (-1: none to wait for)
    uint64_t seen_calls = 0;         // ctx->calls when the pipeline last looked: anything else the host enqueued moves it
    hipEvent_t ev = nullptr;         // orders a lane behind foreign work on ctx, and ctx behind lane 1 (join)
    uint64_t scans = 0;
    // ma_scan_lanes_synchronize_for ran out: the gates were released (all-ones into every stamp word, through `rescue`: a stream
    // of the low priority class, made at creation, that nothing else is ever enqueued on) and the pipeline takes no more scans.
    // drained: both streams ran empty afterwards.
    bool broken = false, drained = true;
    hipStream_t rescue = nullptr;
    uint64_t* pinned = nullptr;      // 64 bytes the rescue stream copies stamp words into for the error text
    // ma_scan_lanes_test_hold_next_scan: the next scan sits behind a word nobody writes until the release above
    bool hold_next = false, hold_armed = false;
    uint64_t* hold_word = nullptr;
    uint64_t hold_seq = 0;
};

static const char* const kLanesBroken =
    "this pipeline's bounded wait ran out and its gates were released (ma_scan_lanes_synchronize_for): destroy it; the context "
    "takes a new one";

// One scan on the lane whose turn it is: ordered behind foreign work on the caller's context, gated on the early stamp of the scan
// before it, launched by `launch(scan context, stamp, stamp value, early stamp)` with nothing waited for.
// A single-column scan of fewer bytes than this takes less time on one stream (4.4-5.9 us, launch to launch) than the host needs to
// enqueue a gate and a launch on the other stream (~6 us): 2^20 rows of any type lose through the pipeline, 2^22 8-byte rows and 2^24
// 1-byte rows gain (profiles/r05_scan_lanes_api.jsonl). Such scans stay on the caller's stream. (A FUSED launch costs one stream
// ~9.7 us however few its rows: those always alternate — 6.6 us.)
constexpr size_t kTinyScanBytes = (size_t)12 << 20;

template <typename Launch>
static ma_status enqueue_on_lane(ma_scan_lanes* lanes, size_t scan_bytes, Launch&& launch) {
    MA_REQUIRE(lanes != nullptr, MA_ERR_INVALID_ARGUMENT, "lanes is NULL");
    std::lock_guard<std::mutex> lock(lanes->mu);
    MA_REQUIRE(!lanes->broken, MA_ERR_DEVICE, "%s", kLanesBroken);
    ma_ctx* ctx = lanes->ctx;
    const bool tiny = scan_bytes < kTinyScanBytes;
    const int k = tiny ? 0 : lanes->turn;
    ma_ctx* sc = k == 0 ? ctx : lanes->lane;
    MA_HIP(hipSetDevice(ctx->device));
    // The context's call counter ONCE, in front of everything: what it shows beyond `seen_calls` is foreign work. An entry
    // another host thread makes after this look is not absorbed below (only the entries this call makes itself are added to
    // the snapshot), so the next scan on lane 1 is ordered behind it.
    const uint64_t now = ctx->calls.load(std::memory_order_relaxed);
    if (k == 1) {
        sc->variant = ctx->variant;  // the tuning knobs follow the caller's context
        sc->blocks_per_cu = ctx->blocks_per_cu;
        sc->grid_override = ctx->grid_override;
        // whatever else the host put on ctx's stream since the pipeline last looked (a kernel that WRITES the column this scan
        // reads, say) comes first; lane 0 is that stream itself
        if (now != lanes->seen_calls) {
            MA_HIP(hipEventRecord(lanes->ev, ctx->stream));
            MA_HIP(hipStreamWaitEvent(sc->stream, lanes->ev, 0));
        }
    }
    if (lanes->hold_next) {  // testing hook: this scan never starts by itself
        lanes->hold_next = false;
        if (hipStreamWaitValue64(sc->stream, lanes->hold_word, lanes->hold_seq + 1, hipStreamWaitValueGte, ~(uint64_t)0) != hipSuccess) {
            (void)hipGetLastError();
            set_error("this runtime has no stream memory operations: the hold hook cannot hold a stream");
            return MA_ERR_UNSUPPORTED;
        }
        lanes->hold_armed = true;
    }
    if (lanes->prev >= 0 && lanes->prev != k)
        MA_HIP(hipStreamWaitValue64(sc->stream, lanes->stamp[lanes->prev] + 1, lanes->seq[lanes->prev], hipStreamWaitValueGte, ~(uint64_t)0));
    ma_status st;
    const uint64_t own0 = entries_by_this_thread();
    {
        NoSync enqueue_only;  // whatever mode ctx is in: the results are waited for by ma_scan_lanes_synchronize
        st = launch(sc, lanes->stamp[k], lanes->seq[k] + 1, lanes->stamp[k] + 1);
    }
    // lane 0's launch entered `ctx` itself (lane 1's entered the lane's own context)
    lanes->seen_calls = now + (k == 0 ? entries_by_this_thread() - own0 : 0);
    if (st != MA_OK) return st;  // nothing was launched: the sequence, the turn and the gate stay as they were
    ++lanes->seq[k];
    lanes->prev = k;
    if (!tiny) lanes->turn = k ^ 1;
    ++lanes->scans;
    return MA_OK;
}

extern "C" {

ma_status ma_scan_lanes_create(ma_ctx* ctx, ma_scan_lanes** out_lanes) {
    MA_REQUIRE(out_lanes != nullptr, MA_ERR_INVALID_ARGUMENT, "out_lanes is NULL");
    *out_lanes = nullptr;
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    MA_NO_CAPTURE(ctx, "ma_scan_lanes_create");
    MA_HIP(hipSetDevice(ctx->device));
    std::unique_ptr<ma_scan_lanes> p(new ma_scan_lanes());
    p->ctx = ctx;
    auto fail = [&](ma_status st) {
        for (uint64_t* s : p->stamp)
            if (s) (void)ma_stamp_free(ctx, s);
        if (p->ev) (void)hipEventDestroy(p->ev);
        if (p->lane) ma_ctx_destroy(p->lane);
        if (p->rescue) (void)hipStreamDestroy(p->rescue);
        if (p->pinned) (void)hipHostFree(p->pinned);
        return st;
    };
    ma_status st = create_ctx_in_class(ctx->ordinal, 0, &p->lane);
    if (st != MA_OK) return fail(st);
    if ((st = ma_ctx_set_async(p->lane, 1)) != MA_OK) return fail(st);
    for (int k = 0; k < 2; ++k)  // plain device words: a wait on signal memory holds up the OTHER stream's dispatches (ma_stamp_alloc)
        if ((st = stamp_alloc_kind(ctx, &p->stamp[k], false)) != MA_OK) return fail(st);
    if (hipEventCreateWithFlags(&p->ev, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        set_error("hipEventCreate failed");
        return fail(MA_ERR_DEVICE);
    }
    // stream memory operations are what the gate is made of: one wait that is already satisfied tells whether the runtime has them
    if (hipStreamWaitValue64(p->lane->stream, p->stamp[0] + 1, 0, hipStreamWaitValueGte, ~(uint64_t)0) != hipSuccess) {
        (void)hipGetLastError();
        set_error("this runtime has no stream memory operations (hipStreamWaitValue64): scans cannot be gated on an early stamp");
        return fail(MA_ERR_UNSUPPORTED);
    }
    // what the bounded wait needs is made NOW, not while a stream is stuck: a stream in the low priority class (its own pool of
    // hardware queues: never behind a held ordinary stream) and a pinned line to read stamp words into
    {
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) (void)hipGetLastError();
        if (hipStreamCreateWithPriority(&p->rescue, hipStreamNonBlocking, least) != hipSuccess) {
            (void)hipGetLastError();
            p->rescue = nullptr;  // a bounded wait can then still report, not release
        }
        if (hipHostMalloc((void**)&p->pinned, 64, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            p->pinned = nullptr;
        }
    }
    p->seen_calls = ctx->calls.load(std::memory_order_relaxed);
    *out_lanes = p.release();
    return MA_OK;
}

ma_status ma_scan_lanes_sum_fused(ma_scan_lanes* lanes, size_t n_cols, const ma_fused_column* cols) {
    return enqueue_on_lane(lanes, kTinyScanBytes, [&](ma_ctx* sc, uint64_t* stamp, uint64_t value, uint64_t* early) {
        return sum_fused_impl(sc, n_cols, cols, stamp, value, false, early);
    });
}

ma_status ma_scan_lanes_sum(ma_scan_lanes* lanes, int32_t format_code, const void* data, size_t n, const uint8_t* mask_bits,
                            size_t mask_bit_offset, int64_t null_count, void* out_sum, double* out_lo, uint64_t* out_valid_count) {
    MA_REQUIRE(out_sum == nullptr || pointer_kind(out_sum) != kPageable, MA_ERR_INVALID_ARGUMENT,
               "out_sum must be device-reachable (device or ma_alloc64_pinned memory): the call only enqueues");
    MA_REQUIRE(out_lo == nullptr || pointer_kind(out_lo) != kPageable, MA_ERR_INVALID_ARGUMENT, "out_lo must be device-reachable");
    MA_REQUIRE(out_valid_count == nullptr || pointer_kind(out_valid_count) != kPageable, MA_ERR_INVALID_ARGUMENT,
               "out_valid_count must be device-reachable");
    const size_t elem = strchr("cC", (char)format_code) ? 1 : strchr("sS", (char)format_code) ? 2 : strchr("iIf", (char)format_code) ? 4 : 8;
    return enqueue_on_lane(lanes, format_code > 0 && format_code < 128 ? n * elem : 0, [&](ma_ctx* sc, uint64_t* stamp, uint64_t value, uint64_t* early) {
        return sum_stamped_any(sc, format_code, data, n, mask_bits, mask_bit_offset, null_count, out_sum, out_lo, out_valid_count, stamp,
                               value, early);
    });
}

ma_status ma_scan_lanes_join(ma_scan_lanes* lanes) {
    MA_REQUIRE(lanes != nullptr, MA_ERR_INVALID_ARGUMENT, "lanes is NULL");
    std::lock_guard<std::mutex> lock(lanes->mu);
    MA_REQUIRE(!lanes->broken, MA_ERR_DEVICE, "%s", kLanesBroken);
    MA_HIP(hipSetDevice(lanes->ctx->device));
    MA_HIP(hipEventRecord(lanes->ev, lanes->lane->stream));
    MA_HIP(hipStreamWaitEvent(lanes->ctx->stream, lanes->ev, 0));
    return MA_OK;
}

static ma_status synchronize_locked(ma_scan_lanes* lanes) {
    const ma_status second = ma_ctx_synchronize(lanes->lane);  // its scans wait for the first lane's early stamps only
    const ma_status first = ma_ctx_synchronize(lanes->ctx);
    return first != MA_OK ? first : second;
}

ma_status ma_scan_lanes_synchronize(ma_scan_lanes* lanes) {
    MA_REQUIRE(lanes != nullptr, MA_ERR_INVALID_ARGUMENT, "lanes is NULL");
    std::lock_guard<std::mutex> lock(lanes->mu);
    MA_REQUIRE(!lanes->broken, MA_ERR_DEVICE, "%s", kLanesBroken);
    return synchronize_locked(lanes);
}

namespace {

using Clock = std::chrono::steady_clock;
double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }

// Polls both streams for at most timeout_ms; done[k] = lane k's stream has run empty. hipStreamQuery also pushes out whatever
// the runtime still holds back.
bool wait_lanes(ma_scan_lanes* lanes, double timeout_ms, bool (&done)[2], hipError_t* error) {
    hipStream_t s[2] = {lanes->ctx->stream, lanes->lane->stream};
    const auto t0 = Clock::now();
    *error = hipSuccess;
    for (;;) {
        for (int k = 0; k < 2; ++k) {
            if (done[k]) continue;
            const hipError_t q = hipStreamQuery(s[k]);
            if (q == hipSuccess) {
                done[k] = true;
            } else {
                (void)hipGetLastError();
                if (q != hipErrorNotReady) {
                    *error = q;
                    return false;
                }
            }
        }
        if (done[0] && done[1]) return true;
        const double ms = ms_since(t0);
        if (ms >= timeout_ms) return false;
        if (ms < 10.0) __builtin_ia32_pause();  // a hot loop ends in this wait: back to back at first, then 1/200 of the time waited
        else std::this_thread::sleep_for(std::chrono::microseconds((long)std::min(500.0, std::max(20.0, ms * 5.0))));
    }
}

// A stamp word's value through the rescue stream, itself bounded (50 ms); false when it did not arrive.
bool read_word(ma_scan_lanes* lanes, const uint64_t* word, uint64_t* out) {
    if (!lanes->rescue || !lanes->pinned) return false;
    if (hipMemcpyAsync(lanes->pinned, word, 8, hipMemcpyDeviceToHost, lanes->rescue) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    const auto t0 = Clock::now();
    while (hipStreamQuery(lanes->rescue) != hipSuccess) {
        (void)hipGetLastError();
        if (ms_since(t0) > 50.0) return false;
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    *out = *lanes->pinned;
    return true;
}

void write_word(ma_scan_lanes* lanes, uint64_t* word, const uint64_t* value) {
    if (stamp_host_store(word, *value)) return;  // host memory (signal memory is): no GPU queue involved
    if (!lanes->rescue) return;                   // never the null stream: it would wait for the very streams that are held
    if (hipStreamWriteValue64(lanes->rescue, word, *value, 0) == hipSuccess) return;
    (void)hipGetLastError();
    if (hipMemcpyAsync(word, value, 8, hipMemcpyHostToDevice, lanes->rescue) != hipSuccess) (void)hipGetLastError();
}

// Everything a stream of the pipeline can be held behind gets the value that ends the wait: the hold hook's word, and all-ones
// in both words of both stamp lines (word 1, the early stamp, is what the gates wait on).
void release_gates(ma_scan_lanes* lanes) {
    static const uint64_t kAll = ~(uint64_t)0;  // (static: the source of a copy on the rescue stream must outlive this call)
    if (lanes->hold_armed && lanes->hold_word) {
        write_word(lanes, lanes->hold_word, &kAll);  // the pipeline takes no more scans after a release: the word need not hold again
        lanes->hold_armed = false;
    }
    for (int k = 0; k < 2; ++k) {
        write_word(lanes, lanes->stamp[k], &kAll);
        write_word(lanes, lanes->stamp[k] + 1, &kAll);
    }
}

}  // namespace

ma_status ma_scan_lanes_synchronize_for(ma_scan_lanes* lanes, double timeout_ms) {
    MA_REQUIRE(lanes != nullptr, MA_ERR_INVALID_ARGUMENT, "lanes is NULL");
    std::lock_guard<std::mutex> lock(lanes->mu);
    MA_REQUIRE(!lanes->broken, MA_ERR_DEVICE, "%s", kLanesBroken);
    if (!(timeout_ms > 0)) return synchronize_locked(lanes);
    MA_HIP(hipSetDevice(lanes->ctx->device));
    bool done[2] = {false, false};
    hipError_t e = hipSuccess;
    if (wait_lanes(lanes, timeout_ms, done, &e)) return synchronize_locked(lanes);  // both ran empty: the latched conditions
    // Past the deadline. Say which lane waits for which sequence, release every gate, give the streams 2 s to run empty — the
    // caller's context (lane 0 is ITS stream) must come back usable — and take no more scans.
    std::string text;
    for (int k = 0; k < 2; ++k) {
        if (done[k]) continue;
        char buf[400];
        const int o = k ^ 1;
        uint64_t fin = 0, early = 0, oearly = 0;
        const bool have = read_word(lanes, lanes->stamp[k], &fin) && read_word(lanes, lanes->stamp[k] + 1, &early) &&
                          read_word(lanes, lanes->stamp[o] + 1, &oearly);
        if (have)
            snprintf(buf, sizeof(buf), "%slane %d (%s stream): %llu of its %llu scans have finished (early stamp %llu); its scans are gated "
                     "on the early stamp of lane %d, which holds %llu of sequence %llu", text.empty() ? "" : "; ", k,
                     k == 0 ? "the context's" : "the pipeline's own", (unsigned long long)fin, (unsigned long long)lanes->seq[k],
                     (unsigned long long)early, o, (unsigned long long)oearly, (unsigned long long)lanes->seq[o]);
        else
            snprintf(buf, sizeof(buf), "%slane %d (%s stream): scan sequence %llu still pending", text.empty() ? "" : "; ", k,
                     k == 0 ? "the context's" : "the pipeline's own", (unsigned long long)lanes->seq[k]);
        text += buf;
    }
    guard_log("scan lanes: %.0f ms passed with scans pending: %s", timeout_ms, text.c_str());
    if (e == hipSuccess) {
        release_gates(lanes);
        bool after[2] = {done[0], done[1]};
        hipError_t e2 = hipSuccess;
        lanes->drained = wait_lanes(lanes, 2000.0, after, &e2);
    } else {
        lanes->drained = false;
    }
    lanes->broken = true;
    guard_log("scan lanes: gates released; the streams %s", lanes->drained ? "have run empty" : "are STILL busy");
    if (e != hipSuccess)
        set_error("a stream of the scan pipeline failed (%s); the pipeline takes no more scans: destroy it", hipGetErrorString(e));
    else
        set_error("the scan pipeline did not finish within %.0f ms — still pending: %s. Its gates were released and the streams %s; the "
                  "results of the scans in flight are undefined, the pipeline takes no more scans: destroy it",
                  timeout_ms, text.c_str(), lanes->drained ? "have run empty since" : "are STILL busy: the device may need a reset");
    return MA_ERR_DEVICE;
}

int32_t ma_scan_lanes_is_broken(ma_scan_lanes* lanes) {
    if (!lanes) return 0;
    std::lock_guard<std::mutex> lock(lanes->mu);
    return !lanes->broken ? 0 : (lanes->drained ? 1 : 2);
}

uint64_t ma_scan_lanes_scans(ma_scan_lanes* lanes) {
    if (!lanes) return 0;
    std::lock_guard<std::mutex> lock(lanes->mu);
    return lanes->scans;
}

// Never an unbounded wait: MINARROW_HIP_DESTROY_WAIT_MS (10 s) for the scans in flight, then the release above. What a stream
// that STILL has not run empty holds (the second lane's context, the stamp lines) is left to the process.
void ma_scan_lanes_destroy(ma_scan_lanes* lanes) {
    if (!lanes) return;
    if (!lanes->broken) (void)ma_scan_lanes_synchronize_for(lanes, destroy_wait_ms());
    (void)hipSetDevice(lanes->ctx->device);
    if (!(lanes->broken && !lanes->drained)) {
        if (lanes->hold_armed) release_gates(lanes);
        for (uint64_t* s : lanes->stamp)
            if (s) (void)ma_stamp_free(lanes->ctx, s);
        if (lanes->hold_word) (void)ma_stamp_free(lanes->ctx, lanes->hold_word);
        if (lanes->ev) (void)hipEventDestroy(lanes->ev);
        ma_ctx_destroy(lanes->lane);
        if (lanes->rescue) (void)hipStreamDestroy(lanes->rescue);
        if (lanes->pinned) (void)hipHostFree(lanes->pinned);
    }
    delete lanes;
}

// Testing hook (include/minarrow_hip_testing.h; inert unless MINARROW_HIP_TEST_HOOKS=1 when the library was loaded): the next scan is
// enqueued behind a word nobody writes — a gate that never opens — until a bounded wait releases it.
ma_status ma_scan_lanes_test_hold_next_scan(ma_scan_lanes* lanes) {
    MA_REQUIRE(lanes != nullptr, MA_ERR_INVALID_ARGUMENT, "lanes is NULL");
    MA_TRY(test_hooks_enabled());
    std::lock_guard<std::mutex> lock(lanes->mu);
    if (!lanes->hold_word) MA_TRY(stamp_alloc_kind(lanes->ctx, &lanes->hold_word, true));  // host-releasable when the runtime has it
    lanes->hold_next = true;
    return MA_OK;
}

}  // extern "C"
