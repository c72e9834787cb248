// Back-to-back sums of mid-size columns on ONE GPU — the reference's hot loop, `for _ in 0..N { sum(&arr) }` over an
// IntegerArray / FloatArray (benches/hotloop_benchmark_avg_std.rs:48-62: ITERATIONS passes, an i64 and an f64 sum each; the pass itself: hotloop_benchmark_std.rs:109-127) — as a pipeline. A launch costs ~3.3 us beyond its
// bytes (ramp + hand-off) and one stream starts scan k + 1 only behind the LAST workgroup of scan k:
// 2^24-row sums run at 0.70-0.73 of peak that way, the 125 M-row step of an 8-way partition at 0.885. ma_scan_lanes_* puts
// consecutive fused scans on two streams of the device in turn and starts each one when the scan in front of it has begun
// to DRAIN (its early stamp: FusedArgs::early_word, ma_reduce_fused.hip) — the ramp of scan k + 1 runs under the stragglers
// of scan k, not beside its whole length. It is what ma_group_* does per member under MA_GROUP_SCAN_LANES (ma_group.hip),
// for a host that drives one GPU without a group.
#include <cstring>
#include <memory>
#include <mutex>

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
};

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
    ma_ctx* ctx = lanes->ctx;
    const bool tiny = scan_bytes < kTinyScanBytes;
    const int k = tiny ? 0 : lanes->turn;
    ma_ctx* sc = k == 0 ? ctx : lanes->lane;
    MA_HIP(hipSetDevice(ctx->device));
    if (k == 1) {
        sc->variant = ctx->variant;  // the tuning knobs follow the caller's context
        sc->blocks_per_cu = ctx->blocks_per_cu;
        sc->grid_override = ctx->grid_override;
        // whatever else the host put on ctx's stream since the pipeline last looked (a kernel that WRITES the column this scan
        // reads, say) comes first; lane 0 is that stream itself
        if (ctx->calls.load(std::memory_order_relaxed) != lanes->seen_calls) {
            MA_HIP(hipEventRecord(lanes->ev, ctx->stream));
            MA_HIP(hipStreamWaitEvent(sc->stream, lanes->ev, 0));
        }
    }
    if (lanes->prev >= 0 && lanes->prev != k)
        MA_HIP(hipStreamWaitValue64(sc->stream, lanes->stamp[lanes->prev] + 1, lanes->seq[lanes->prev], hipStreamWaitValueGte, ~(uint64_t)0));
    ma_status st;
    {
        NoSync enqueue_only;  // whatever mode ctx is in: the results are waited for by ma_scan_lanes_synchronize
        st = launch(sc, lanes->stamp[k], lanes->seq[k] + 1, lanes->stamp[k] + 1);
    }
    lanes->seen_calls = ctx->calls.load(std::memory_order_relaxed);
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
    MA_HIP(hipSetDevice(lanes->ctx->device));
    MA_HIP(hipEventRecord(lanes->ev, lanes->lane->stream));
    MA_HIP(hipStreamWaitEvent(lanes->ctx->stream, lanes->ev, 0));
    return MA_OK;
}

ma_status ma_scan_lanes_synchronize(ma_scan_lanes* lanes) {
    MA_REQUIRE(lanes != nullptr, MA_ERR_INVALID_ARGUMENT, "lanes is NULL");
    std::lock_guard<std::mutex> lock(lanes->mu);
    const ma_status second = ma_ctx_synchronize(lanes->lane);  // its scans wait for the first lane's early stamps only
    const ma_status first = ma_ctx_synchronize(lanes->ctx);
    return first != MA_OK ? first : second;
}

uint64_t ma_scan_lanes_scans(ma_scan_lanes* lanes) {
    if (!lanes) return 0;
    std::lock_guard<std::mutex> lock(lanes->mu);
    return lanes->scans;
}

void ma_scan_lanes_destroy(ma_scan_lanes* lanes) {
    if (!lanes) return;
    (void)ma_scan_lanes_synchronize(lanes);
    (void)hipSetDevice(lanes->ctx->device);
    for (uint64_t* s : lanes->stamp)
        if (s) (void)ma_stamp_free(lanes->ctx, s);
    if (lanes->ev) (void)hipEventDestroy(lanes->ev);
    ma_ctx_destroy(lanes->lane);
    delete lanes;
}

}  // extern "C"
