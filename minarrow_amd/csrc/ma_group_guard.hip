// First contact with a multi-GPU node, made survivable. The reference's parallel reduction is a plain `main` over a Rayon
// pool (benches/benchmark_parallel_simd.rs:81-125): a worker that panics ends the process with a message. The group's twin of
// that failure is worse — a collective whose peer never arrives, or a stream held behind a value nobody writes, is a host
// blocked in hipStreamSynchronize for good. This file is what a host gets instead:
//
//   ma_group_synchronize_for   ma_group_synchronize with a deadline: polls the members' streams; past the deadline it aborts
//                              every communicator (ncclCommAbort ends the collective kernels in flight), releases whatever a
//                              stream may be held behind, marks the group broken and returns MA_ERR_DEVICE naming the
//                              members and phases that were still pending.
//   ma_group_rebuild_exchange  a fresh exchange (other flags: one notch down) for the same members — their contexts, and every
//                              column allocated from them, stay.
//   ma_group_selftest          rank-tagged records through the group's exchange in each hand-off / issue form, the gathered
//                              blocks compared in rank order and the finals on every member; a peer-copy round trip between
//                              every ordered pair of distinct devices; a stamp written by a kernel on each member and waited on
//                              by that member's exchange stream — each step under the deadline.
//   ma_group_test_stall_next_exchange / _corrupt_next_exchange   the faults themselves, for a one-GPU box.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <memory>
#include <mutex>

#include "minarrow_hip_testing.h"

#include "ma_group.hpp"

using namespace ma;
using namespace ma::grp;

namespace ma {

__global__ void stamp_store_kernel(uint64_t* stamp, uint64_t value) {
    // what the fused scan's final thread does behind its results (ma_reduce_fused.hip)
    __hip_atomic_store(stamp, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

void guard_log(const char* fmt, ...) {
    static const bool on = [] {
        const char* e = getenv("MINARROW_HIP_GUARD_LOG");
        return e && e[0] && e[0] != '0';
    }();
    if (!on) return;
    static const auto t0 = std::chrono::steady_clock::now();
    char buf[4096];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    fprintf(stderr, "[minarrow_hip guard %9.3f ms] %s\n",
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), buf);
    fflush(stderr);
}

double destroy_wait_ms() {
    const char* e = getenv("MINARROW_HIP_DESTROY_WAIT_MS");  // read at every destroy: a host may set it late
    const double v = e ? atof(e) : 0.0;
    return v > 0.0 ? v : 10000.0;
}

bool call_bounded(const std::function<void()>& fn, double timeout_ms) {
    struct State {
        std::mutex mu;
        std::condition_variable cv;
        bool done = false;
    };
    auto st = std::make_shared<State>();
    std::thread([st, fn] {
        fn();
        std::lock_guard<std::mutex> lock(st->mu);
        st->done = true;
        st->cv.notify_all();
    }).detach();
    std::unique_lock<std::mutex> lock(st->mu);
    return st->cv.wait_for(lock, std::chrono::duration<double, std::milli>(timeout_ms), [&] { return st->done; });
}

hipError_t launch_stamp_store(hipStream_t stream, uint64_t* stamp, uint64_t value) {
    hipLaunchKernelGGL(stamp_store_kernel, dim3(1), dim3(1), 0, stream, stamp, value);
    return hipGetLastError();
}

namespace grp {

const char* const kBrokenMessage =
    "an earlier exchange of this group failed or timed out and its communicators were aborted: ma_group_rebuild_exchange, or "
    "destroy the group";

namespace {

using Clock = std::chrono::steady_clock;

double us_since(Clock::time_point t0) { return std::chrono::duration<double, std::micro>(Clock::now() - t0).count(); }

hipStream_t rescue_stream(ma_group* g, size_t i) {
    if (g->rescue.size() < g->ctxs.size()) g->rescue.resize(g->ctxs.size(), nullptr);
    for (size_t j = 0; j < g->ctxs.size() && !g->rescue[i]; ++j)  // one per device: it only ever carries 8-byte writes
        if (g->rescue[j] && g->ctxs[j]->device == g->ctxs[i]->device) g->rescue[i] = g->rescue[j];
    if (!g->rescue[i]) {
        (void)hipSetDevice(g->ctxs[i]->device);
        // the LOW priority class: the runtime keeps a hardware-queue pool per class, so this stream never sits behind an ordinary
        // one that is held (ordinary streams share hardware queues once a process has more than a few)
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) (void)hipGetLastError();
        if (hipStreamCreateWithPriority(&g->rescue[i], hipStreamNonBlocking, least) != hipSuccess) {
            (void)hipGetLastError();
            g->rescue[i] = nullptr;
        }
    }
    return g->rescue[i];
}

// `*word = *value` from the host while the member's own streams may be stuck: a write packet on the rescue stream (the twin
// of the wait the stuck stream sits in), or a small copy there when the runtime has no stream memory operations. `value`
// points at storage that outlives the copy.
void write_word(ma_group* g, size_t i, uint64_t* word, const uint64_t* value) {
    if (stamp_host_store(word, *value)) {  // host memory (signal memory is): no GPU queue involved at all
        guard_log("member %zu: word %p <- %llu by a host store", i, (void*)word, (unsigned long long)*value);
        return;
    }
    hipStream_t s = rescue_stream(g, i);
    if (!s) {  // never the null stream: it would wait for the very streams that are held
        guard_log("member %zu: no rescue stream: word %p is not released", i, (void*)word);
        return;
    }
    (void)hipSetDevice(g->ctxs[i]->device);
    const hipError_t w = hipStreamWriteValue64(s, word, *value, 0);
    guard_log("member %zu: word %p <- %llu through the rescue stream: %s (signal memory: %d)", i, (void*)word,
              (unsigned long long)*value, hipGetErrorString(w), ma_stamp_is_signal(word));
    if (w == hipSuccess) return;
    (void)hipGetLastError();
    if (hipMemcpyAsync(word, value, 8, hipMemcpyHostToDevice, s) != hipSuccess) (void)hipGetLastError();
}

// The value of a device word, read through the rescue stream with a bound of its own (~50 ms); false when it did not arrive.
bool read_word(ma_group* g, size_t i, const uint64_t* word, uint64_t* out) {
    hipStream_t s = rescue_stream(g, i);
    if (!s) return false;
    static thread_local uint64_t* pinned = nullptr;
    if (!pinned && hipHostMalloc((void**)&pinned, 64, hipHostMallocPortable) != hipSuccess) {  // every member's device copies into it
        (void)hipGetLastError();
        pinned = nullptr;
        return false;
    }
    (void)hipSetDevice(g->ctxs[i]->device);
    if (hipMemcpyAsync(pinned, word, 8, hipMemcpyDeviceToHost, s) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    const auto t0 = Clock::now();
    while (hipStreamQuery(s) != hipSuccess) {
        (void)hipGetLastError();
        if (us_since(t0) > 50e3) return false;
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    *out = *pinned;
    return true;
}

struct Pending {
    std::vector<uint8_t> scan, side, lane;  // 1 = that stream of the member has run empty (lane: the second scan lane)
    bool all_done = false;
    hipError_t error = hipSuccess;
    size_t error_member = 0;
};

// One look at every stream. hipStreamQuery also pushes out anything the runtime still holds back.
void look(ma_group* g, Pending& p) {
    const size_t n = g->ctxs.size();
    if (p.scan.empty()) {
        p.scan.assign(n, 0);
        p.side.assign(n, 0);
        p.lane.assign(n, 0);
    }
    bool all = true;
    for (size_t i = 0; i < n; ++i) {
        (void)hipSetDevice(g->ctxs[i]->device);
        hipStream_t streams[3] = {g->ctxs[i]->stream, (g->overlap && i < g->side.size() && g->side[i]) ? g->side[i]->stream : nullptr,
                                  (g->lanes2 && i < g->scan2.size() && g->scan2[i]) ? g->scan2[i]->stream : nullptr};
        uint8_t* done[3] = {&p.scan[i], &p.side[i], &p.lane[i]};
        for (int k = 0; k < 3; ++k) {
            if (*done[k]) continue;
            if (!streams[k]) {
                *done[k] = 1;
                continue;
            }
            const hipError_t q = hipStreamQuery(streams[k]);
            if (q == hipSuccess) {
                *done[k] = 1;
            } else {
                (void)hipGetLastError();
                all = false;
                if (q != hipErrorNotReady && p.error == hipSuccess) {
                    p.error = q;
                    p.error_member = i;
                }
            }
        }
    }
    p.all_done = all;
}

// Polls until every stream has run empty or `timeout_ms` have passed. True when all ran empty.
bool wait_streams(ma_group* g, double timeout_ms, Pending& p) {
    const auto t0 = Clock::now();
    for (;;) {
        look(g, p);
        if (p.all_done || p.error != hipSuccess) return p.all_done;
        const double us = us_since(t0);
        if (us >= timeout_ms * 1e3) return false;
        // a benchmark's timed region ends in this wait: poll back to back for the first 10 ms, then sleep 1/200 of the time
        // waited so far (the overshoot stays under ~1 %)
        if (us < 10e3)
            __builtin_ia32_pause();
        else
            std::this_thread::sleep_for(std::chrono::microseconds((long)std::min(500.0, std::max(20.0, us / 200.0))));
    }
}

// "member 3 (device 3): exchange stream waiting for the scan's stamp (have 41, want 42)" for every stream still pending.
std::string describe(ma_group* g, const Pending& p) {
    std::string text;
    const size_t n = g->ctxs.size();
    int listed = 0;
    for (size_t i = 0; i < n; ++i) {
        if (p.scan[i] && p.side[i] && p.lane[i]) continue;
        if (++listed > 8) {
            text += "; ...";
            break;
        }
        char buf[400];
        std::string what;
        if (!p.scan[i]) what = g->overlap ? "scan stream (scans, or the wait for an earlier exchange of the set being re-filled)" :
                                            (g->use_rccl ? "stream (scan, all-gather or fold)" : "stream (scan)");
        if (!p.lane[i]) what += (what.empty() ? "" : " and ") + std::string("second scan lane (a scan, or its wait for the step before it)");
        if (!p.side[i]) {
            if (!what.empty()) what += " and ";
            // which hand-off the last exchange used, and whether its value has arrived
            const int set = g->last;
            uint64_t have = 0;
            if (g->set_used[set] && i < g->stamp[set].size() && g->stamp[set][i] && read_word(g, i, g->stamp[set][i], &have) &&
                have < g->stamp_seq[set] && have != ~(uint64_t)0) {
                snprintf(buf, sizeof(buf), "exchange stream, possibly still waiting for the scan's stamp (have %llu, want %llu)",
                         (unsigned long long)have, (unsigned long long)g->stamp_seq[set]);
                what += buf;
            } else {
                what += "exchange stream (all-gather or fold in flight)";
            }
        }
        if (g->lanes2 && (!p.scan[i] || !p.lane[i])) {
            // a lane's scan is gated on the EARLY stamp (word 1) of the step before it on the other lane: what the words hold,
            // against the sequence of each set's last stamped step
            for (int k = 0; k < 2; ++k) {
                uint64_t fin = 0, early = 0;
                if (i < g->stamp[k].size() && g->stamp[k][i] && read_word(g, i, g->stamp[k][i], &fin) && read_word(g, i, g->stamp[k][i] + 1, &early)) {
                    snprintf(buf, sizeof(buf), "; set %d stamps: final %lld early %lld of sequence %llu", k, (long long)fin, (long long)early,
                             (unsigned long long)g->stamp_seq[k]);
                    what += buf;
                }
            }
        }
        snprintf(buf, sizeof(buf), "%smember %zu (device %d): %s", text.empty() ? "" : "; ", i, g->ctxs[i]->device, what.c_str());
        text += buf;
    }
    return text;
}

}  // namespace

ma_status enqueue_stall(ma_group* g, size_t member, hipStream_t stream) {
    const size_t n = g->ctxs.size();
    if (g->stall_word.size() < n) g->stall_word.resize(n, nullptr);
    MA_HIP(hipSetDevice(g->ctxs[member]->device));
    if (!g->stall_word[member]) MA_TRY(stamp_alloc_kind(g->ctxs[member], &g->stall_word[member], true));  // host-releasable
    (void)rescue_stream(g, member);  // made now: not while a stream is stuck
    // the word holds the sequence of the last stall that was released; this one waits for the next
    if (hipStreamWaitValue64(stream, g->stall_word[member], g->stall_seq + 1, hipStreamWaitValueGte, ~(uint64_t)0) != hipSuccess) {
        (void)hipGetLastError();
        set_error("this runtime has no stream memory operations: the stall hook cannot hold a stream");
        return MA_ERR_UNSUPPORTED;
    }
    g->stall_armed = true;
    guard_log("member %zu: exchange held behind its stall word %p >= %llu (testing hook; signal memory: %d)", member,
              (void*)g->stall_word[member], (unsigned long long)g->stall_seq + 1, ma_stamp_is_signal(g->stall_word[member]));
    return MA_OK;
}

void release_waits(ma_group* g, bool stamps_too) {
    static const uint64_t kAll = ~(uint64_t)0;
    const size_t n = g->ctxs.size();
    if (g->stall_armed) {
        guard_log("releasing the stall words (sequence %llu)", (unsigned long long)g->stall_seq + 1);
        ++g->stall_seq;
        g->stall_release = g->stall_seq;
        for (size_t i = 0; i < n && i < g->stall_word.size(); ++i)
            if (g->stall_word[i]) write_word(g, i, g->stall_word[i], &g->stall_release);
        g->stall_armed = false;
    }
    if (stamps_too)
        for (int k = 0; k < 2; ++k)
            for (size_t i = 0; i < n && i < g->stamp[k].size(); ++i) {
                if (!g->stamp[k][i]) continue;
                write_word(g, i, g->stamp[k][i], &kAll);
                // the scan lanes' gates wait on word 1 of the line (the early stamp): a lane parked behind a step that will never
                // run would otherwise keep the group from ever running empty
                if (g->lanes2) write_word(g, i, g->stamp[k][i] + 1, &kAll);
            }
}

// The order matters. First everything a stream of the group may be HELD behind is released and the streams get a moment:
// a collective that was queued behind such a wait starts while its communicator is still alive (launched after the abort
// it would run on freed communicator state). Then ncclCommAbort ends the collective kernels that are in flight — the ones
// whose peer never arrived. Then a bounded wait for the streams to run empty.
void abort_locked(ma_group* g, const char* why) {
    guard_log("group abort: %s", why ? why : "");
    release_waits(g, true);
    guard_log("group abort: held streams released; giving queued work 200 ms");
    {
        Pending p;
        const bool idle = wait_streams(g, 200.0, p);
        guard_log("group abort: after 200 ms the streams are %s%s", idle ? "idle" : "still pending: ", idle ? "" : describe(g, p).c_str());
        for (size_t i = 0; i < g->rescue.size(); ++i)
            if (g->rescue[i]) guard_log("group abort: rescue stream of member %zu: %s", i, hipGetErrorString(hipStreamQuery(g->rescue[i])));
        (void)hipGetLastError();
    }
    const RcclApi* api = g->comms.empty() ? nullptr : rccl();
    for (size_t i = 0; i < g->comms.size(); ++i) {
        if (!g->comms[i] || !api || !api->CommAbort) continue;
        (void)hipSetDevice(g->ctxs[i]->device);
        guard_log("group abort: ncclCommAbort(member %zu)", i);
        ncclComm_t comm = g->comms[i];
        const int dev = g->ctxs[i]->device;
        auto abort_fn = api->CommAbort;
        // documented to return, but it waits for the communicator's work inside: a stream that is held for a reason nothing here
        // could release would keep it for good. Bounded; a call that has not returned is left behind (drained stays false).
        if (!call_bounded([comm, dev, abort_fn] { (void)hipSetDevice(dev); (void)abort_fn(comm); }, 5000.0))
            guard_log("group abort: ncclCommAbort(member %zu) has not returned within 5 s: left behind", i);
        g->comms[i] = nullptr;
    }
    guard_log("group abort: communicators aborted; waiting up to 5 s for the streams");
    Pending p;
    g->drained = wait_streams(g, 5000.0, p);
    guard_log("group abort: streams %s", g->drained ? "have run empty" : "are STILL busy");
    g->broken = true;
    g->fail_member = g->stall_member = g->corrupt_member = -1;
}

ma_status synchronize_for_locked(ma_group* g, double timeout_ms) {
    MA_REQUIRE(!g->broken, MA_ERR_DEVICE, "%s", kBrokenMessage);
    if (!(timeout_ms > 0)) return synchronize_locked(g);
    Pending p;
    if (!wait_streams(g, timeout_ms, p)) {
        guard_log("group wait: %.0f ms passed with streams pending", timeout_ms);
        if (p.error != hipSuccess) {
            const hipError_t e = p.error;
            const size_t m = p.error_member;
            abort_locked(g, "a stream reported an error");
            set_error("member %zu's stream failed (%s); the group's communicators were aborted (ma_group_rebuild_exchange, or "
                      "destroy the group)", m, hipGetErrorString(e));
            return MA_ERR_DEVICE;
        }
        const std::string pending = describe(g, p);
        abort_locked(g, pending.c_str());
        set_error("the group did not finish within %.0f ms — still pending: %s. Its communicators were aborted and the streams %s "
                  "(ma_group_rebuild_exchange with other flags, or destroy the group)",
                  timeout_ms, pending.c_str(), g->drained ? "have run empty since" : "are STILL busy: the device may need a reset");
        return MA_ERR_DEVICE;
    }
    return synchronize_locked(g);  // everything has finished: the latched conditions, the host fold
}

}  // namespace grp
}  // namespace ma

// ---- the self-test ------------------------------------------------------------------------------------------------------

namespace {

const char* const kFormNames[MA_SELFTEST_FORMS] = {"in-stream/threads", "in-stream/caller", "overlap-event/threads", "overlap-event/caller",
                                                    "overlap-stamp/threads", "overlap-stamp/caller", "host-fold", "-"};

struct SelfTest {
    ma_group* g;
    double timeout_ms;
    ma_selftest_report* rep;
    std::string text;
    std::vector<std::vector<uint64_t>> blocks;  // per member: kColumns tagged records
    // ... and their pinned twins, the source of the async uploads (a copy from pageable memory blocks inside the runtime behind a
    // stuck stream, in front of every deadline); one block per form and member, kept until the test ends — or, on a group that
    // ended up broken, for good: a copy still queued behind a stuck stream must not read freed memory
    std::vector<uint64_t*> pinned;
    uint64_t round = 0;
    ~SelfTest() {
        if (g->broken) return;
        for (uint64_t* p : pinned) (void)hipHostFree(p);
    }

    void say(const char* fmt, ...) __attribute__((format(printf, 2, 3))) {
        char buf[400];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        if (!text.empty()) text += "; ";
        text += buf;
    }

    void tag_blocks() {
        const size_t n = g->ctxs.size();
        ++round;
        blocks.assign(n, std::vector<uint64_t>(kBlockWords, 0));
        for (size_t i = 0; i < n; ++i)
            for (int c = 0; c < kColumns; ++c) {
                uint64_t* r = &blocks[i][(size_t)c * kRecordWords];
                r[0] = 0x0101010101010101ull * (i + 1) + (uint64_t)c + round * 1000003ull;
                r[1] = i + 1;
                const double hi = (double)(i + 1) * 1e3 + c + 0.5, lo = std::ldexp((double)(i + 1), -70);
                memcpy(&r[2], &hi, 8);
                memcpy(&r[3], &lo, 8);
                r[4] = 3 * (i + 1);
                for (size_t w = 5; w < kRecordWords; ++w) r[w] = 0xA5A5000000000000ull ^ (i << 16) ^ (uint64_t)(c << 8) ^ w ^ (round << 32);
            }
    }

    // One exchange of tagged records in the form (handoff, issue) the group is switched to; `bit` = its MA_SELFTEST_FORM_*.
    ma_status run_form(int bit, int handoff, bool threads) {
        const size_t n = g->ctxs.size();
        rep->forms_tried |= 1u << bit;
        if (threads && !g->threads) start_workers(g);
        if (!threads && g->threads) stop_workers(g);
        g->handoff = handoff;
        tag_blocks();
        const int set = g->overlap ? g->cur : 0;
        const bool want_stamp = bit == MA_SELFTEST_FORM_OVERLAP_STAMP_THREADS || bit == MA_SELFTEST_FORM_OVERLAP_STAMP_CALLER;
        const uint64_t seq = want_stamp ? ++g->stamp_seq[set] : 0;
        for (size_t i = 0; i < n; ++i) {
            uint64_t* local = (g->overlap && set == 1) ? g->local1[i] : g->local[i];
            if (!g->use_rccl) {
                memcpy(local, blocks[i].data(), kBlockWords * 8);  // pinned host records; the streams are idle
                continue;
            }
            MA_HIP(hipSetDevice(g->ctxs[i]->device));
            hipStream_t fill = scan_ctx(g, set, i)->stream;  // the stream that fills this record set (the second lane for set 1)
            uint64_t* stage = nullptr;
            MA_HIP(hipHostMalloc((void**)&stage, kBlockWords * 8, hipHostMallocPortable));
            pinned.push_back(stage);
            memcpy(stage, blocks[i].data(), kBlockWords * 8);
            MA_HIP(hipMemcpyAsync(local, stage, kBlockWords * 8, hipMemcpyHostToDevice, fill));
            if (want_stamp) {
                hipLaunchKernelGGL(stamp_store_kernel, dim3(1), dim3(1), 0, fill, g->stamp[set][i], seq);
                MA_HIP(hipGetLastError());
            }
        }
        g->prev_set = -1;
        g->stamp_ok[set] = want_stamp;
        g->enq_mask[set] = (kColumns >= 32) ? ~0u : ((1u << kColumns) - 1u);
        const auto t0 = std::chrono::steady_clock::now();
        MA_TRY(exchange_locked(g));
        const ma_status st = synchronize_for_locked(g, timeout_ms);
        rep->form_us[bit] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (st != MA_OK) {
            rep->failed_form = bit;
            rep->timed_out = g->broken ? 1 : 0;
            say("%s: %s", kFormNames[bit], ma_last_error_string());
            return st;
        }
        // what every member must hold: the member-ordered fold of the tagged records, and (RCCL) the blocks in rank order
        int bad_member = -1;
        const char* bad_what = "";
        for (size_t m = 0; m < n && bad_member < 0; ++m) {
            for (int c = 0; c < kColumns && bad_member < 0; ++c) {
                HostFoldDD f;
                for (size_t i = 0; i < n; ++i) f.add(&blocks[i][(size_t)c * kRecordWords]);
                const double total = f.total();
                uint64_t want[4] = {f.isum, f.icnt, 0, f.fcnt};
                memcpy(&want[2], &total, 8);
                if (memcmp(finals_of(g, m, c), want, 32) != 0) {
                    bad_member = (int)m;
                    bad_what = "finals differ from the member-ordered fold of the tagged records";
                }
            }
            if (g->use_rccl && bad_member < 0) {
                std::vector<uint64_t> got(n * kBlockWords);
                const uint64_t* gathered = (g->overlap && g->last == 1) ? g->gathered1[m] : g->gathered[m];
                MA_HIP(hipSetDevice(g->ctxs[m]->device));
                MA_HIP(hipMemcpy(got.data(), gathered, got.size() * 8, hipMemcpyDeviceToHost));
                for (size_t i = 0; i < n && bad_member < 0; ++i)
                    if (memcmp(&got[i * kBlockWords], blocks[i].data(), kBlockWords * 8) != 0) {
                        bad_member = (int)m;
                        bad_what = "the gathered blocks are not the members' records in rank order";
                    }
            }
        }
        if (bad_member >= 0) {
            rep->failed_form = bit;
            rep->failed_member = bad_member;
            say("%s: member %d: %s", kFormNames[bit], bad_member, bad_what);
            set_error("self-test, form %s: member %d: %s", kFormNames[bit], bad_member, bad_what);
            return MA_ERR_DEVICE;
        }
        rep->forms_ok |= 1u << bit;
        say("%s ok %.0f us", kFormNames[bit], rep->form_us[bit]);
        return MA_OK;
    }

    // Bounded wait for ONE stream of one member; on expiry the group is aborted.
    ma_status wait_one(size_t member, hipStream_t s, const char* what) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            const hipError_t q = hipStreamQuery(s);
            if (q == hipSuccess) return MA_OK;
            (void)hipGetLastError();
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (q != hipErrorNotReady || us >= timeout_ms * 1e3) {
                abort_locked(g, what);
                rep->failed_member = (int)member;
                rep->timed_out = q == hipErrorNotReady ? 1 : 0;
                set_error("self-test: member %zu (device %d): %s %s; the group's communicators were aborted "
                          "(ma_group_rebuild_exchange, or destroy the group)",
                          member, g->ctxs[member]->device, what,
                          q == hipErrorNotReady ? "did not finish within the deadline" : hipGetErrorString(q));
                say("%s", ma_last_error_string());
                return MA_ERR_DEVICE;
            }
            if (us > 200.0) std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
    }

    ma_status peer_copies() {
        const size_t n = g->ctxs.size();
        constexpr size_t kBytes = (size_t)1 << 20;
        std::vector<void*> a(n, nullptr), b(n, nullptr), c(n, nullptr);
        // pinned staging: an async copy from / into pageable memory blocks INSIDE the runtime until it has run — behind a stuck
        // stream that is for good, in front of the deadline below — and would land in freed memory if it ran after a timeout
        uint64_t *pattern = nullptr, *back = nullptr;
        if (hipHostMalloc((void**)&pattern, kBytes, hipHostMallocPortable) != hipSuccess ||
            hipHostMalloc((void**)&back, kBytes, hipHostMallocPortable) != hipSuccess) {
            if (pattern) (void)hipHostFree(pattern);
            return hip_fail(hipGetLastError(), "self-test staging buffers", __FILE__, __LINE__);
        }
        ma_status st = MA_OK;
        int slow_i = -1, slow_j = -1;
        auto cleanup = [&] {  // never on a broken group: a stuck stream may still use the buffers — they go with the process
            for (size_t i = 0; i < n; ++i) {
                (void)hipSetDevice(g->ctxs[i]->device);
                for (void* p : {a[i], b[i], c[i]})
                    if (p) (void)hipFree(p);
            }
            (void)hipHostFree(pattern);
            (void)hipHostFree(back);
        };
        for (size_t i = 0; i < n && st == MA_OK; ++i)
            for (size_t j = 0; j < n && st == MA_OK; ++j) {
                const int di = g->ctxs[i]->device, dj = g->ctxs[j]->device;
                if (di == dj || !g->peer[i * n + j]) continue;
                ++rep->peer_pairs;
                for (size_t m : {i, j}) {
                    if (a[m]) continue;
                    hipError_t e = hipSetDevice(g->ctxs[m]->device);
                    for (void** p : {&a[m], &b[m], &c[m]})
                        if (e == hipSuccess) e = hipMalloc(p, kBytes);
                    if (e != hipSuccess) {
                        cleanup();
                        return hip_fail(e, "self-test peer buffers", __FILE__, __LINE__);
                    }
                }
                for (size_t w = 0; w < kBytes / 8; ++w) pattern[w] = (w * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)i << 56) ^ ((uint64_t)j << 48);
                hipStream_t s = g->ctxs[i]->stream;
                hipError_t e = hipSetDevice(di);
                const auto t0 = std::chrono::steady_clock::now();
                if (e == hipSuccess) e = hipMemcpyAsync(a[i], pattern, kBytes, hipMemcpyHostToDevice, s);
                if (e == hipSuccess) e = hipMemsetAsync(c[i], 0, kBytes, s);
                if (e == hipSuccess) e = hipMemcpyPeerAsync(b[j], dj, a[i], di, kBytes, s);  // i -> j over the link ...
                if (e == hipSuccess) e = hipMemcpyPeerAsync(c[i], di, b[j], dj, kBytes, s);  // ... and back
                if (e == hipSuccess) e = hipMemcpyAsync(back, c[i], kBytes, hipMemcpyDeviceToHost, s);
                if (e != hipSuccess) {
                    cleanup();
                    return hip_fail(e, "self-test peer copy", __FILE__, __LINE__);
                }
                char what[96];
                snprintf(what, sizeof(what), "the peer-copy round trip %d -> %d -> %d", di, dj, di);
                st = wait_one(i, s, what);
                if (st != MA_OK) break;
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                if (memcmp(back, pattern, kBytes) != 0) {
                    rep->failed_member = (int)i;
                    say("peer copy %d -> %d -> %d returned different bytes", di, dj, di);
                    set_error("self-test: the peer-copy round trip %d -> %d -> %d returned different bytes", di, dj, di);
                    st = MA_ERR_DEVICE;
                    break;
                }
                ++rep->peer_pairs_ok;
                if (us > rep->peer_us_max) {
                    rep->peer_us_max = us;
                    slow_i = di;
                    slow_j = dj;
                }
            }
        if (!g->broken) cleanup();  // a stuck stream may still hold the buffers: they go with the process
        if (st == MA_OK) {
            if (rep->peer_pairs)
                say("peer round trips %d/%d ok (1 MiB each way; slowest %d<->%d %.0f us)", rep->peer_pairs_ok, rep->peer_pairs, slow_i,
                    slow_j, rep->peer_us_max);
            else
                say("peer round trips: n/a (no pair of distinct peer-capable devices)");
        }
        return st;
    }

    ma_status stamp_waits() {
        const size_t n = g->ctxs.size();
        if (!g->overlap) {
            say("stamp waits: n/a (no exchange streams)");
            return MA_OK;
        }
        for (int k = 0; k < 2; ++k) {
            bool have_all = g->stamp[k].size() == n;
            for (size_t i = 0; have_all && i < n; ++i) have_all = g->stamp[k][i] != nullptr;
            if (!have_all) continue;
            const uint64_t seq = ++g->stamp_seq[k];  // one sequence per set, as a stamped step has: every member's word gets it
            for (size_t i = 0; i < n; ++i) {
                ++rep->stamp_waits;
                MA_HIP(hipSetDevice(g->ctxs[i]->device));
                // the order of a real step: the storing kernel is on the scan stream before the exchange stream is made to wait
                // (the other order could sit behind its own wait where the two streams share a hardware queue)
                const auto t0 = std::chrono::steady_clock::now();
                hipLaunchKernelGGL(stamp_store_kernel, dim3(1), dim3(1), 0, g->ctxs[i]->stream, g->stamp[k][i], seq);
                MA_HIP(hipGetLastError());
                MA_HIP(hipStreamWaitValue64(g->side[i]->stream, g->stamp[k][i], seq, hipStreamWaitValueGte, ~(uint64_t)0));
                MA_TRY(wait_one(i, g->side[i]->stream, "the exchange stream's wait for a stamp a kernel on the scan stream stores"));
                MA_TRY(wait_one(i, g->ctxs[i]->stream, "the stamping kernel"));
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                if (us > rep->stamp_us_max) rep->stamp_us_max = us;
                ++rep->stamp_waits_ok;
            }
        }
        if (g->lanes2) {  // the lanes' gate: lane 1 waits on the EARLY word of set 0's stamp, stored by a kernel on lane 0 (and back)
            for (int k = 0; k < 2; ++k) {
                const uint64_t seq = ++g->stamp_seq[k];
                for (size_t i = 0; i < n; ++i) {
                    ++rep->stamp_waits;
                    MA_HIP(hipSetDevice(g->ctxs[i]->device));
                    hipStream_t from = scan_ctx(g, k, i)->stream, to = scan_ctx(g, k ^ 1, i)->stream;
                    hipLaunchKernelGGL(stamp_store_kernel, dim3(1), dim3(1), 0, from, g->stamp[k][i] + 1, seq);
                    MA_HIP(hipGetLastError());
                    hipLaunchKernelGGL(stamp_store_kernel, dim3(1), dim3(1), 0, from, g->stamp[k][i], seq);  // the set's sequence stays level
                    MA_HIP(hipGetLastError());
                    MA_HIP(hipStreamWaitValue64(to, g->stamp[k][i] + 1, seq, hipStreamWaitValueGte, ~(uint64_t)0));
                    MA_TRY(wait_one(i, to, "a scan lane's wait for the early stamp a kernel on the other lane stores"));
                    MA_TRY(wait_one(i, from, "the stamping kernel"));
                    ++rep->stamp_waits_ok;
                }
            }
        }
        if (rep->stamp_waits)
            say("stamp waits %d/%d ok (slowest %.0f us from launch to the exchange stream running on)", rep->stamp_waits_ok,
                rep->stamp_waits, rep->stamp_us_max);
        else
            say("stamp waits: n/a (this runtime gave no waitable words: hand-off by events)");
        return MA_OK;
    }
};

}  // namespace

extern "C" {

ma_status ma_group_synchronize_for(ma_group* group, double timeout_ms) {
    MA_REQUIRE(group != nullptr, MA_ERR_INVALID_ARGUMENT, "group is NULL");
    std::lock_guard<std::recursive_mutex> lock(group->mu);
    return synchronize_for_locked(group, timeout_ms);
}

int32_t ma_group_is_broken(ma_group* group) {
    if (!group) return 0;
    std::lock_guard<std::recursive_mutex> lock(group->mu);
    return group->broken ? (group->drained ? 1 : 2) : 0;
}

uint32_t ma_group_flags(ma_group* group) {
    if (!group) return 0;
    std::lock_guard<std::recursive_mutex> lock(group->mu);
    uint32_t f = group->flags & ~(uint32_t)(MA_GROUP_ISSUE_CALLER | MA_GROUP_EXCHANGE_RCCL | MA_GROUP_EXCHANGE_OVERLAP | MA_GROUP_SCAN_LANES);
    if (group->use_rccl) f |= MA_GROUP_EXCHANGE_RCCL;
    if (group->overlap) f |= MA_GROUP_EXCHANGE_OVERLAP;
    if (group->lanes2 && group->lanes_on) f |= MA_GROUP_SCAN_LANES;
    if (!group->threads) f |= MA_GROUP_ISSUE_CALLER;
    return f;
}

ma_status ma_group_set_handoff(ma_group* group, int32_t kind) {
    MA_REQUIRE(group != nullptr, MA_ERR_INVALID_ARGUMENT, "group is NULL");
    MA_REQUIRE(kind == MA_GROUP_HANDOFF_STAMP || kind == MA_GROUP_HANDOFF_EVENT, MA_ERR_INVALID_ARGUMENT,
               "hand-off kind %d (MA_GROUP_HANDOFF_STAMP or MA_GROUP_HANDOFF_EVENT)", kind);
    std::lock_guard<std::recursive_mutex> lock(group->mu);
    group->handoff = kind;
    return MA_OK;
}

int32_t ma_group_handoff(ma_group* group) {
    if (!group) return -1;
    std::lock_guard<std::recursive_mutex> lock(group->mu);
    if (!group->overlap || !group->use_rccl) return -1;
    if (group->handoff != MA_GROUP_HANDOFF_STAMP || (tuning_variant(group->ctxs[0]) & 4096)) return MA_GROUP_HANDOFF_EVENT;
    for (int k = 0; k < 2; ++k) {
        if (group->stamp[k].size() != group->ctxs.size()) return MA_GROUP_HANDOFF_EVENT;
        for (uint64_t* w : group->stamp[k])
            if (!w) return MA_GROUP_HANDOFF_EVENT;
    }
    return MA_GROUP_HANDOFF_STAMP;
}

ma_status ma_group_set_scan_lanes(ma_group* group, int32_t on) {
    MA_REQUIRE(group != nullptr, MA_ERR_INVALID_ARGUMENT, "group is NULL");
    std::lock_guard<std::recursive_mutex> lock(group->mu);
    MA_REQUIRE(!group->broken, MA_ERR_DEVICE, "%s", kBrokenMessage);
    if (!group->lanes2) {
        MA_REQUIRE(!on, MA_ERR_UNSUPPORTED, "this group has no second scan lanes (create or rebuild it with MA_GROUP_SCAN_LANES)");
        return MA_OK;
    }
    if (on != 2 && group->lanes_on == (on != 0)) return MA_OK;
    MA_TRY(synchronize_locked(group));  // a set is never filled from two streams at once
    if (on == 2) {
        // fresh second lanes: new contexts, i.e. new streams, and the steps start from rest — what a host tries when its trial found
        // the lanes no faster than one stream
        for (size_t i = 0; i < group->ctxs.size(); ++i) {
            ma_ctx* fresh = nullptr;
            MA_TRY(create_ctx_in_class(group->ctxs[i]->ordinal, group->carrier_class, &fresh));
            const ma_status st = ma_ctx_set_async(fresh, 1);
            if (st != MA_OK) {
                ma_ctx_destroy(fresh);
                return st;
            }
            ma_ctx_destroy(group->scan2[i]);
            group->scan2[i] = fresh;
        }
        for (uint8_t& lane : group->mark_lane)  // the marks the old lane contexts recorded went with them
            if (lane == 1) lane = 255;
    }
    group->lanes_on = on != 0;
    group->prev_set = -1;
    return MA_OK;
}

ma_status ma_group_join_lanes(ma_group* group) {
    MA_REQUIRE(group != nullptr, MA_ERR_INVALID_ARGUMENT, "group is NULL");
    std::lock_guard<std::recursive_mutex> lock(group->mu);
    MA_REQUIRE(!group->broken, MA_ERR_DEVICE, "%s", kBrokenMessage);
    if (!group->lanes2 || !group->lanes_on) return MA_OK;
    for (size_t i = 0; i < group->ctxs.size(); ++i) {  // the member's own stream behind everything its second lane has been given
        MA_HIP(hipSetDevice(group->ctxs[i]->device));
        MA_HIP(hipEventRecord(group->ev_lane[i], group->scan2[i]->stream));
        MA_HIP(hipStreamWaitEvent(group->ctxs[i]->stream, group->ev_lane[i], 0));
    }
    group->prev_set = -1;
    return MA_OK;
}

ma_status ma_group_rebuild_exchange(ma_group* group, uint32_t flags) {
    MA_REQUIRE(group != nullptr, MA_ERR_INVALID_ARGUMENT, "group is NULL");
    std::lock_guard<std::recursive_mutex> lock(group->mu);
    if (group->broken) {
        if (!group->drained) {  // one more bounded look: the streams may have run empty since the abort
            grp::release_waits(group, true);
            Pending p;  // scan streams, exchange streams, second scan lanes
            group->drained = wait_streams(group, 2000.0, p);
        }
        MA_REQUIRE(group->drained, MA_ERR_DEVICE,
                   "the group's streams are still busy after its communicators were aborted: it cannot be rebuilt (the device may "
                   "need a reset)");
    } else {
        MA_TRY(synchronize_locked(group));
    }
    guard_log("group rebuild: releasing the old exchange (flags %u -> %u)", group->flags, flags);
    release_exchange(group);  // drains (now idle) streams, frees both record sets, side contexts, stamps, communicators
    guard_log("group rebuild: setting up the new exchange");
    group->drained = true;
    group->handoff = 0;
    const ma_status st = configure_exchange(group, flags);
    // no exchange at all after a failed set-up (its pieces were released): the group stays refused until a rebuild succeeds
    group->broken = st != MA_OK;
    if (st != MA_OK) release_exchange(group);
    return st;
}

ma_status ma_group_test_stall_next_exchange(ma_group* group, int32_t member) {
    MA_REQUIRE(group != nullptr, MA_ERR_INVALID_ARGUMENT, "group is NULL");
    MA_TRY(test_hooks_enabled());
    std::lock_guard<std::recursive_mutex> lock(group->mu);
    MA_REQUIRE(member >= 0 && (size_t)member < group->ctxs.size(), MA_ERR_INVALID_ARGUMENT, "member %d out of range", member);
    group->stall_member = member;
    return MA_OK;
}

ma_status ma_group_test_corrupt_next_exchange(ma_group* group, int32_t member) {
    MA_REQUIRE(group != nullptr, MA_ERR_INVALID_ARGUMENT, "group is NULL");
    MA_TRY(test_hooks_enabled());
    std::lock_guard<std::recursive_mutex> lock(group->mu);
    MA_REQUIRE(member >= 0 && (size_t)member < group->ctxs.size(), MA_ERR_INVALID_ARGUMENT, "member %d out of range", member);
    group->corrupt_member = member;
    return MA_OK;
}

ma_status ma_group_selftest(ma_group* group, uint32_t what, double timeout_ms, ma_selftest_report* out_report) {
    MA_REQUIRE(group != nullptr, MA_ERR_INVALID_ARGUMENT, "group is NULL");
    MA_REQUIRE(timeout_ms > 0, MA_ERR_INVALID_ARGUMENT, "the self-test needs a deadline (timeout_ms > 0)");
    ma_selftest_report local_report;
    ma_selftest_report* rep = out_report ? out_report : &local_report;
    memset(rep, 0, sizeof(*rep));
    rep->struct_bytes = (uint32_t)sizeof(*rep);
    rep->failed_form = rep->failed_member = -1;
    if (what == 0) what = MA_SELFTEST_EXCHANGE | MA_SELFTEST_PEER_COPIES | MA_SELFTEST_STAMPS;
    std::lock_guard<std::recursive_mutex> lock(group->mu);
    ma_group* g = group;
    MA_REQUIRE(!g->broken, MA_ERR_DEVICE, "%s", kBrokenMessage);
    const size_t n = g->ctxs.size();
    rep->n_members = (int32_t)n;
    std::vector<int> devs;
    for (ma_ctx* c : g->ctxs)
        if (std::find(devs.begin(), devs.end(), c->device) == devs.end()) devs.push_back(c->device);
    rep->n_devices = (int32_t)devs.size();
    rep->exchange_kind = g->use_rccl ? 1 : 0;
    if (g->use_rccl) {
        int ranks = 0;
        const RcclApi* api = rccl();
        if (api && api->CommCount && !g->comms.empty() && g->comms[0] && api->CommCount(g->comms[0], &ranks) == ncclSuccess)
            rep->rccl_ranks = ranks;
    }
    SelfTest t{g, timeout_ms, rep, {}, {}, {}, 0};
    t.say("%zu members on %d device(s), %s exchange%s", n, rep->n_devices, g->use_rccl ? "RCCL" : "host-fold",
          g->use_rccl ? (g->overlap ? " overlapped on side streams" : " on the scan streams") : "");
    // whatever the host had in flight first, under the same deadline
    ma_status st = synchronize_for_locked(g, timeout_ms);
    const bool had_threads = g->threads;
    const int had_handoff = g->handoff;
    if (st == MA_OK && (what & (MA_SELFTEST_EXCHANGE | MA_SELFTEST_EXCHANGE_ALL_FORMS))) {
        const bool all = (what & MA_SELFTEST_EXCHANGE_ALL_FORMS) != 0;
        bool stamps = g->overlap && !(tuning_variant(g->ctxs[0]) & 4096);
        for (int k = 0; k < 2 && stamps; ++k) {
            stamps = g->stamp[k].size() == n;
            for (size_t i = 0; stamps && i < n; ++i) stamps = g->stamp[k][i] != nullptr;
        }
        struct Form {
            int bit, handoff;
            bool threads;
        };
        std::vector<Form> forms;
        auto add = [&](int bit_threads, int handoff) {
            if (all || had_threads) forms.push_back({bit_threads, handoff, true});
            if (all || !had_threads) forms.push_back({bit_threads + 1, handoff, false});
        };
        if (!g->use_rccl) {
            forms.push_back({MA_SELFTEST_FORM_HOST_FOLD, 0, had_threads});
        } else if (!g->overlap) {
            add(MA_SELFTEST_FORM_IN_STREAM_THREADS, 0);
        } else {
            const bool configured_stamp = stamps && had_handoff == MA_GROUP_HANDOFF_STAMP;
            if (all || !configured_stamp) add(MA_SELFTEST_FORM_OVERLAP_EVENT_THREADS, MA_GROUP_HANDOFF_EVENT);
            if (stamps && (all || configured_stamp)) add(MA_SELFTEST_FORM_OVERLAP_STAMP_THREADS, MA_GROUP_HANDOFF_STAMP);
        }
        for (const Form& f : forms) {
            // both record sets of an overlapped group go through the form (the sets alternate)
            st = t.run_form(f.bit, f.handoff, f.threads);
            if (st == MA_OK && g->overlap) {
                rep->forms_ok &= ~(1u << f.bit);
                st = t.run_form(f.bit, f.handoff, f.threads);
            }
            if (st != MA_OK) break;
        }
    }
    if (st == MA_OK && (what & MA_SELFTEST_STAMPS)) st = t.stamp_waits();
    if (st == MA_OK && (what & MA_SELFTEST_PEER_COPIES)) st = t.peer_copies();
    // leave the group as it was found: issue form, hand-off, no record of the test's exchanges
    if (!g->broken) {
        if (had_threads && !g->threads) start_workers(g);
        if (!had_threads && g->threads) stop_workers(g);
        g->handoff = had_handoff;
        for (size_t i = 0; i < n; ++i) {
            if (!g->use_rccl) {
                memset(g->local[i], 0, kBlockWords * 8);
                continue;
            }
            (void)hipSetDevice(g->ctxs[i]->device);
            (void)hipMemsetAsync(g->local[i], 0, kBlockWords * 8, g->ctxs[i]->stream);
            if (g->overlap) (void)hipMemsetAsync(g->local1[i], 0, kBlockWords * 8, g->ctxs[i]->stream);
        }
        for (int k = 0; k < 2; ++k) {
            g->enq_mask[k] = g->exchanged_mask[k] = 0;
            g->stamp_ok[k] = false;
        }
        const std::string keep = st != MA_OK ? ma_last_error_string() : "";
        const ma_status drained = synchronize_for_locked(g, timeout_ms);
        if (st == MA_OK) st = drained;
        else set_error("%s", keep.c_str());
        (void)hipSetDevice(g->ctxs[0]->device);
        g->timer.report(nullptr, nullptr, nullptr);  // the test's samples are not the host's
    }
    snprintf(rep->text, sizeof(rep->text), "%s%s", st == MA_OK ? "PASS: " : "FAIL: ", t.text.c_str());
    return st;
}

}  // extern "C"
