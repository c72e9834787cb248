// Internal header of the single-process multi-GPU group (ma_group.hip: creation, enqueue, exchange; ma_group_guard.hip:
// bounded waits, abort, exchange rebuild, self-test). The C ABI on top of it is ma_group_* (include/minarrow_hip.h).
#pragma once

#include <chrono>
#include <condition_variable>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "ma_rccl.hpp"

struct ma_group {
    std::vector<ma_ctx*> ctxs;
    std::recursive_mutex mu;  // recursive: the one-call forms (ma_group_sum_*) hold it across enqueue + exchange + synchronize
    // ---- per-member issue threads (see the header comment). One job at a time, posted under `mu`.
    bool threads = false;
    std::vector<std::thread> workers;
    const std::function<ma_status(size_t)>* job = nullptr;
    std::atomic<uint64_t> job_seq{0};
    struct alignas(64) Slot {
        std::atomic<uint64_t> done{0};
        ma_status status = MA_OK;
        std::string message;
    };
    std::vector<Slot> slots;
    std::mutex sleep_mu;
    std::condition_variable work_cv, done_cv;
    std::atomic<int> sleepers{0};
    std::atomic<int> caller_waiting{0};
    std::atomic<bool> stop{false};
    // peer[i * G + j]: member i's device can address member j's device memory (same device, or hipDeviceCanAccessPeer and
    // enabled at creation). Probed once; ma_group_consolidate_column refuses pairs that are not.
    std::vector<uint8_t> peer;
    // The HIP device a member's chunks must be resident on: its context's device (ma_group_test_set_member_device
    // overrides it so that the refusal paths can be exercised on a one-GPU box).
    std::vector<int> home;
    bool use_rccl = false;
    uint32_t flags = 0;   // the MA_GROUP_* flags in effect (creation, or the last ma_group_rebuild_exchange)
    // An exchange failed on some member after others had enqueued theirs, or a bounded wait ran out: the communicators were
    // aborted. Every call that would wait on the exchange refuses from then on; ma_group_rebuild_exchange clears it.
    bool broken = false;
    bool drained = true;  // after an abort: did every stream of the group run empty? (a rebuild needs that)
    int fail_member = -1; // ma_group_test_fail_next_exchange: that member's next exchange fails in front of its all-gather
    // ma_group_test_stall_next_exchange: that member's next exchange is held behind a word nobody writes (stall_word[member],
    // hipStreamWaitValue64) until an abort — or the group's destruction — releases it: what a lost peer or a fabric fault looks
    // like to the waiting host. ma_group_test_corrupt_next_exchange: one word of the records that member gathered is flipped
    // in front of its fold: finals that are wrong on one member only.
    int stall_member = -1, corrupt_member = -1;
    bool stall_armed = false;
    uint64_t stall_seq = 0;      // the stall words hold the sequence of the last release; a new stall waits for the next one
    uint64_t stall_release = 0;  // host storage the release copies read from
    std::vector<uint64_t*> stall_word;
    // One stream per member that nothing else is ever enqueued on: the abort path writes the release values of stall words
    // and stamps through it while the member's own streams are stuck.
    std::vector<hipStream_t> rescue;
    int carrier_class = 0;  // the stream priority class of the members' and exchange streams (+1 in a rehearsal through the loopback double)
    int handoff = 0;      // overlapped exchanges: 0 = the scan's stamp when the step was a stamped launch, 1 = always an event
    std::string peer_note;  // the peer-access summary of ma_group_exchange_note (probed once at creation)
    std::vector<ncclComm_t> comms;
    // RCCL: per member a device block of kColumns records (`local`), a device block of G x kColumns gathered records
    // and a pinned host block of kColumns x 4 finals the fold kernel writes. host: `local[i]` points into `host_records`.
    std::vector<uint64_t*> local, gathered, finals;
    // MA_GROUP_EXCHANGE_OVERLAP (RCCL exchange only): a second record set per member ([1]; the vectors above are set 0), an
    // internal context per member whose stream carries the all-gather + fold of the set just filled while the member's own
    // stream already scans into the other set, and per member and set one event each way. `cur` = the set being filled.
    bool overlap = false;
    int cur = 0, last = 0;  // last = the set of the most recent exchange (what ma_group_result reads)
    std::vector<uint64_t*> local1, gathered1, finals1;
    std::vector<ma_ctx*> side;
    std::vector<hipEvent_t> ev_ready[2], ev_done[2];
    bool set_used[2] = {false, false};
    // Which record slots were filled into each set since its last exchange (overlap only): ma_group_result reads the set of
    // the LAST exchange, and a column that was not enqueued in that step would come back from the other set — the value of
    // two steps ago, or zeros — so it is refused instead.
    uint32_t enq_mask[2] = {0, 0}, exchanged_mask[2] = {0, 0};
    // Event-free hand-off to the exchange stream (overlap only): the fused table launch of a step stamps stamp[set][member]
    // with stamp_seq[set] behind its results, and the member's exchange stream waits for that value (hipStreamWaitValue64)
    // instead of an event recorded on the scan stream — which then carries nothing but scans. stamp_ok[set]: every launch
    // into the set since its last exchange was such a stamped one (anything else falls back to the event).
    std::vector<uint64_t*> stamp[2];
    uint64_t stamp_seq[2] = {0, 0};
    bool stamp_ok[2] = {false, false};
    // MA_GROUP_SCAN_LANES (overlapped RCCL exchange only): a SECOND scan context per member. Record set 0 is filled by scans on
    // the member's own context, set 1 by scans on scan2[i] — consecutive steps therefore run on two streams, and a stamped step
    // on one lane is gated on the EARLY stamp of the step before it on the other (stored while that step drains: FusedArgs::early_word,
    // ma_reduce_fused.hip): its ramp runs under that step's stragglers and hand-off instead of behind them (125 M rows per column:
    // 0.2842 -> 0.2737 ms per step, profiles/r05_probe_early_stamp.jsonl), without the two scans running side by side for their
    // whole length (which costs 2-5 % from 10^8 rows on). ev_lane / seen_calls: when the host — or a group call that is not such a
    // step — has put work of its own on a member's context (ma_ctx::calls moved), the next lane-1 launch is ordered behind ALL
    // of it with an event first.
    bool lanes2 = false;
    bool lanes_on = true;  // ma_group_set_scan_lanes: with the lanes set up, whether steps use them (a host measures both and keeps the faster)
    std::vector<ma_ctx*> scan2;
    std::vector<hipEvent_t> ev_lane;
    std::vector<uint64_t> seen_calls;
    // ma_group_mark_next_scan: the next table step records these two timing marks on every member's scanning context, right
    // around the scan launch (behind the lane's waits, in front of the exchange); mark_lane[index] = which lane holds mark index
    int mark_from = -1, mark_to = -1;
    std::vector<uint8_t> mark_lane;
    int prev_set = -1;       // the set the previous call — a stamped table step — filled; -1: anything else
    uint64_t prev_seq = 0;   // ... and the sequence its stamps carry
    uint64_t* host_records = nullptr;  // pinned, G x kColumns records (host exchange)
    uint64_t* host_finals = nullptr;   // pinned (RCCL: G x kColumns x 4) or plain (host: kColumns x 4) finals
    // ma_group_consolidate_column: per destination member a grow-only device arena the chunks' validity bytes are
    // gathered into before the bit-granular join (re-used across calls in stream order)
    std::vector<void*> mask_stage;
    std::vector<size_t> mask_stage_bytes;
    char note[512] = "";
    ma::ExchangeTimer timer;        // member 0's exchange, every 4th call (ma_group_exchange_stats)
    double host_fold_us = 0.0;      // host exchange: wall time of the host fold, summed ...
    int host_fold_samples = 0;      // ... over this many synchronizes
};

namespace ma {
ma_status make_lane(ma_ctx* root, ma_ctx** out, int cls = 0);  // ma_ctx.hip: an internal context of root's device, own stream (in priority class cls) + scratch
ma_status create_ctx_in_class(int32_t device_ordinal, int cls, ma_ctx** out);  // ma_ctx.hip: +1 high / -1 low priority stream
ma_status sum_fused_impl(ma_ctx* ctx, size_t n_cols, const ma_fused_column* cols, uint64_t* stamp, uint64_t stamp_value,
                         bool as_partials = false, uint64_t* early_stamp = nullptr);

namespace grp {

constexpr int kColumns = MA_GROUP_MAX_COLUMNS;
constexpr size_t kBlockWords = (size_t)kColumns * kRecordWords;

// The member-ordered fold of the reduction records on the host: the same arithmetic as the device fold
// (ma_fold_sum_records): wrapping integer adds, error-free two-sum for the (hi, lo) pairs.
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

// The context whose stream fills record set `set` on member i.
inline ma_ctx* scan_ctx(const ma_group* g, int set, size_t i) { return (g->lanes2 && g->lanes_on && set == 1) ? g->scan2[i] : g->ctxs[i]; }

// ma_group.hip
void release_exchange(ma_group* g);
ma_status setup_rccl(ma_group* g, bool overlap, bool lanes);
// Lane 1 of member i behind everything the member's own stream has been given so far (an event), when something other than the
// group's stepping has touched the member's context since the group last looked.
ma_status order_lane_if_foreign(ma_group* g, size_t i, uint64_t now);  // now = the member context's call counter as the caller read it
ma_status setup_host(ma_group* g);
void start_workers(ma_group* g);
void stop_workers(ma_group* g);
ma_status run_on_members(ma_group* g, const std::function<ma_status(size_t)>& fn);
ma_status exchange_locked(ma_group* g);
ma_status synchronize_locked(ma_group* g);
ma_status configure_exchange(ma_group* g, uint32_t flags);
const uint64_t* finals_of(const ma_group* g, size_t member, int32_t column);
// ma_group_guard.hip
extern const char* const kBrokenMessage;
// Waits for every stream of the group for at most timeout_ms (<= 0: without limit). Past the deadline: abort_locked, and
// MA_ERR_DEVICE with the members and phases still pending in the thread's error string.
ma_status synchronize_for_locked(ma_group* g, double timeout_ms);
// Aborts every communicator, releases whatever a stream of the group may be held behind (stall words, stamps), waits a
// bounded time for the streams to run empty (g->drained) and marks the group broken.
void abort_locked(ma_group* g, const char* why);
// Writes the release value into every armed stall word (and, stamps_too, ~0 into every stamp) through the rescue streams.
void release_waits(ma_group* g, bool stamps_too);
// Testing hooks: holds `stream` behind member's stall word until release_waits.
ma_status enqueue_stall(ma_group* g, size_t member, hipStream_t stream);

}  // namespace grp
}  // namespace ma
