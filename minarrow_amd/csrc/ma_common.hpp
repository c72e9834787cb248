// Internal definitions shared by the HIP translation units of libminarrow_hip.so.
// Nothing here is part of the C ABI (include/minarrow_hip.h is).
#pragma once

#include "ma_env.hpp"  // the environment surface, one documented line per variable

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include "minarrow_hip.h"

namespace ma {

constexpr int kBlock = 256;       // threads per workgroup: 4 wave64s, one per SIMD of a CU
constexpr int kWaves = kBlock / 64;
constexpr int kMaxGrid = 16384;   // upper bound on workgroups of a reduction launch (scratch is sized for it)
constexpr int kDefaultBlocksPerCu = 2;  // streaming kernels: two waves per SIMD (profiles/r01_sweep_sum_v2.txt)

// One reduction partial: 32 bytes so that a workgroup's record never straddles a 64-B line with
// more than one neighbour.
struct alignas(32) Partial {
    uint64_t a;    // integer sum, or bit pattern of the double-double high part
    uint64_t b;    // bit pattern of the double-double low part (0 for integers)
    uint64_t cnt;  // number of valid rows seen
    uint64_t pad;
};

// Pinned-host result slot a synchronous call reads after the stream drains.
struct ResultSlot {
    uint64_t a;
    uint64_t b;
    uint64_t cnt;
    uint32_t flags;  // device-detected conditions (bit 0: integer divide by zero)
    uint32_t pad;
};

void set_error(const char* fmt, ...);
// hipMalloc on `device` (already current) that releases the device's parked-block cache and retries once when HBM is full.
hipError_t device_malloc(int device, void** out, size_t bytes);
// The same through the device's block cache; the caller has made the device current and, for a free, has made sure that
// nothing in flight still touches the block.
hipError_t device_block_alloc(int device, void** out, size_t bytes);
hipError_t device_block_free(int device, void* ptr);
ma_status hip_fail(hipError_t e, const char* what, const char* file, int line);

}  // namespace ma

#define MA_HIP(expr)                                                        \
    do {                                                                    \
        hipError_t _e = (expr);                                             \
        if (_e != hipSuccess) return ::ma::hip_fail(_e, #expr, __FILE__, __LINE__); \
    } while (0)

#define MA_TRY(expr)                      \
    do {                                  \
        ma_status _s = (expr);            \
        if (_s != MA_OK) return _s;       \
    } while (0)

#define MA_ENTER(ctx) ::ma::Enter _ma_enter(ctx)
#define MA_ENTER_PRIMARY(ctx) ::ma::Enter _ma_enter(ctx, true)

#define MA_REQUIRE(cond, status, ...)     \
    do {                                  \
        if (!(cond)) {                    \
            ::ma::set_error(__VA_ARGS__); \
            return (status);              \
        }                                 \
    } while (0)

struct ma_ctx {
    int device = 0;    // HIP runtime ordinal (hipSetDevice)
    int ordinal = 0;   // library ordinal: index into the MINARROW_HIP_DEVICES list (what ma_ctx_create took)
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    std::atomic<bool> async{false};      // read without the lock by ma::Enter (may a busy context fan out?)
    std::atomic<uint64_t> calls{0};      // entry points that have entered this context (a group with two scan lanes looks at
                                         // it to learn that the host enqueued work of its own on a member's context)
    int num_cus = 0;
    int blocks_per_cu = 0;  // 0 = each kernel's own default
    int variant = 0;
    int grid_override = 0;  // tuning: absolute workgroup count for the streaming kernels (0 = auto)
    std::mutex mu;

    ma::Partial* partials = nullptr;   // device, kMaxGrid records
    unsigned int* ticket = nullptr;    // device, zero between launches
    ma::ResultSlot* result = nullptr;  // pinned host, device-mapped
    uint32_t* dev_flags = nullptr;     // device word for elementwise kernels (divide-by-zero latch)
    hipEvent_t ev_start = nullptr;
    hipEvent_t ev_stop = nullptr;
    std::vector<hipEvent_t> marks;     // ma_ctx_mark: timing events, created on first use of their index
    bool pending_flags = false;        // async mode: dev_flags must be inspected at the next synchronize
    std::atomic<bool> capturing{false};  // between ma_ctx_capture_begin / _end: calls are recorded into a hipGraph
    bool async_before_capture = false;
    void* scratch = nullptr;           // grow-only device scratch for descriptor tables / per-segment partials
    size_t scratch_bytes = 0;          //   (one user at a time: callers hold `mu` and order their use on `stream`)
    void* pipe = nullptr;              // staging ring of the tiled host-operand path (ma_pipeline.hip), made on first use
    size_t staging_tile_bytes = (size_t)32 << 20;  // bytes of one operand per tile; 0 = stage whole operands
    // Lanes: a synchronous call holds its context for the whole call (enqueue + wait). So that host threads sharing
    // one context do not serialise on that wait (the reference's kernels are re-entrant: src/kernels/arithmetic/mod.rs:
    // 29-31), a call that finds the context busy runs on a LANE instead: an internal context of the same device with its
    // own stream and reduction scratch, created on demand (ma::Enter, ma_ctx.hip). Only for synchronous calls on a
    // context that owns its stream: async / capturing / borrowed-stream contexts promise ordering on ONE stream.
    ma_ctx* parent = nullptr;          // set on a lane
    std::vector<ma_ctx*> lanes;        // owned; guarded by lanes_mu
    std::mutex lanes_mu;
    int max_lanes = 4;                 // the context itself included (MINARROW_HIP_LANES)
    uint64_t result_seq = 0;           // stamps of the polled synchronous reductions (ma_reduce.hip)
    long poll_us = 60;                 // MINARROW_HIP_POLL_US: how long such a call polls before it blocks (0 = never poll)
    bool fenced_reduce = false;        // MINARROW_HIP_FENCED_REDUCE=1: the round-1 release/acquire publish in the sum kernels
    // Pinned staging buffers (kTableSlots, used in turn) for small host tables on their way to the device (ma::upload_table): descriptor
    // tables live in the caller's frame, and a pageable source would force a stream drain per call.
    // 16: a segmented chunk list is up to 7 tables per call (122 000 chunks), and a host that streams calls must not find a
    // slot still read by a kernel of the call before — with 4 slots the host waited for the GPU at every other segment of
    // back-to-back calls and the kernels ran at 0.87 of the copy rate instead of 0.99 (profiles/r04_ab_chunked.jsonl)
    static constexpr int kTableSlots = 16;
    void* table_stage[kTableSlots] = {};
    size_t table_stage_bytes[kTableSlots] = {};
    hipEvent_t table_ev[kTableSlots] = {};
    bool table_busy[kTableSlots] = {};
    bool table_mapped[kTableSlots] = {};  // committed in place (table_commit_mapped) and not yet released: its event says nothing yet
    size_t table_high_water = 0;          // largest table this context has staged (new slots are sized for it)
    std::vector<void*> table_garbage;     // outgrown staging buffers: freed with the context (hipHostFree drains the device)
    int table_next = 0;
    int table_cur = -1;                   // the slot table_begin handed out, until it is committed
    // Device copies of tables that go up BESIDE the stream's work (ma::TableUpload): a stream of its own for the copies, a few
    // device buffers used in turn, per buffer the event of its copy and the event behind its last reader's launch.
    static constexpr int kDevTables = 4;
    hipStream_t upload_stream = nullptr;  // made on first use
    void* dev_table[kDevTables] = {};
    size_t dev_table_bytes[kDevTables] = {};
    hipEvent_t dev_table_up[kDevTables] = {};    // recorded on upload_stream behind the buffer's copy
    hipEvent_t dev_table_read[kDevTables] = {};  // recorded on `stream` behind the last launch that reads the buffer
    bool dev_table_has_reader[kDevTables] = {};
    std::vector<void*> dev_table_garbage;        // outgrown buffers: a launched kernel may still read them — freed with the context
    int dev_table_next = 0;
};

// Entry points that must talk to the host (a result copied back, a staging copy, an allocation) cannot be recorded.
#define MA_NO_CAPTURE(ctx, what)                                                                        \
    MA_REQUIRE(!(ctx)->capturing, MA_ERR_INVALID_ARGUMENT,                                              \
               "%s cannot be recorded into a graph: it synchronises with the host (use it outside capture)", what)

namespace ma {

// Entry-point guard. `Enter e(ctx)` locks the context for the duration of the call — or, when another thread holds it
// and the context may fan out (see ma_ctx::lanes), a free lane — and re-points `ctx` at whichever was locked. A call
// made by a thread that already holds (a lane of) the same context — an entry point composed of others — re-uses
// that lane without locking. primary_only: the call is tied to the context's own stream (synchronize, capture,
// timers, collectives).
class Enter {
  public:
    explicit Enter(ma_ctx*& ctx, bool primary_only = false);
    ~Enter();
    Enter(const Enter&) = delete;
    Enter& operator=(const Enter&) = delete;

  private:
    ma_ctx* locked_ = nullptr;
};
// While one is alive on a thread, the entry points that thread calls only enqueue (as in async mode) — the composed
// entry point synchronises once at its end. Never changes the context's user-visible mode.
struct NoSync {
    NoSync();
    ~NoSync();
};
bool nosync_active();

// ---- kernel-form selectors (ma_ctx_set_variant) -------------------------------------------------------------------------------
// Two kinds of bits share ma_ctx::variant. FORM bits force one of two PRODUCT paths that the library otherwise picks by size or
// shape, so that a test can reach the path a 10^9-row input takes with an input a CPU check can follow: they are live in every
// build (kFormBits, six of them; include/minarrow_hip.h lists them). TUNING bits select forms that were measured against the
// defaults and not kept — older launch shapes, other unroll depths and load pacings, the fenced publish, the piece-interleaved
// mapping, the stamp's trigger: they exist only in a library built with -DMA_TUNING=1 (make -C minarrow_amd/csrc TUNING=1 -> build/
// tuning/libminarrow_hip.so, what tools/ sweeps load through MINARROW_HIP_LIB). In the shipped build tuning_variant() is the
// constant 0: the branches fold away, their kernels are never instantiated, and ma_ctx_set_variant refuses the bits
// (MA_ERR_UNSUPPORTED) instead of letting a host select a slower kernel silently.
#ifndef MA_TUNING
#define MA_TUNING 0
#endif
constexpr int kFormBits = 16 | 32 | 128 | 256 | 16384 | 65536;
// Environment variables that only ever served an A/B (MINARROW_HIP_STAMP_SIGNAL, _SCAN_LANE_CLASS, _STREAM_PRIORITY, the tunables of
// the opt-in output search): read in the tuning build only, through this.
inline const char* tuning_env(const char* name) { return MA_TUNING ? getenv(name) : nullptr; }
inline int form_variant(const ma_ctx* ctx) { return ctx->variant & kFormBits; }
#if MA_TUNING
inline int tuning_variant(const ma_ctx* ctx) { return ctx->variant; }
#else
constexpr int tuning_variant(const ma_ctx*) { return 0; }
#endif
// ma_testhooks.hip: MA_OK when the fault hooks are live in this process (MINARROW_HIP_TEST_HOOKS=1 at load), else
// MA_ERR_UNSUPPORTED with the reason in the thread's error string. Every ma_*_test_* entry point starts with it.
ma_status test_hooks_enabled();
// How many entry points THIS thread has entered so far (every context): the difference around a call a pipeline makes itself is
// what that call added to ma_ctx::calls, so that a snapshot of the counter can be told from foreign work without a lock.
uint64_t entries_by_this_thread();
inline bool is_async(const ma_ctx* ctx) { return ctx->async || nosync_active(); }
// Waits for the context's stream and turns a latched device condition (dense integer divide by zero) into its status.
ma_status sync_and_check(ma_ctx* ctx);
// Waits for everything enqueued on the context's stream: polls a stream-written completion word first (small calls),
// then blocks (ma_ctx.hip).
ma_status stream_wait(ma_ctx* ctx);

enum PtrKind : int32_t { kPageable = 0, kPinned = 1, kDevice = 2, kManaged = 3 };
PtrKind pointer_kind(const void* p);
// Which operands of a large elementwise call go through the staging ring (ma_pipeline.hip): pageable host memory
// always; pinned host memory — which kernels COULD address in place — in synchronous mode, because the copy engines
// fill both directions of the link where a kernel reading and writing over it does not (a (+) scalar at 2^28 rows:
// 45.8 vs 49.6 ms, profiles/r01_pcie_tiled.json). An async context keeps pinned operands in place: the call must
// return before the work is done.
inline bool crosses_in_tiles(const ma_ctx* ctx, PtrKind k) { return k == kPageable || (k == kPinned && !is_async(ctx)); }

// The device allocation a pointer was found in during this call (pointer_kind's per-call range cache): lets a loop over
// thousands of chunk pointers test "same allocation as the previous one" with two compares. Only valid while the
// CallScope of the call is alive.
bool known_device_range(const void* p, uintptr_t* lo, uintptr_t* hi);
struct DeviceRange {
    uintptr_t lo = 1, hi = 0;  // empty
    bool holds(const void* p) const { return (uintptr_t)p >= lo && (uintptr_t)p < hi; }
    void learn(const void* p) {
        uintptr_t l, h;
        if (known_device_range(p, &l, &h)) {
            lo = l;
            hi = h;
        }
    }
};

// Makes every buffer of one ABI call device-reachable. Pageable host inputs are copied into temporary
// device buffers; pageable host outputs get a temporary that is copied back by finish(). Using any
// temporary forces the call to be synchronous.
class CallScope {
  public:
    explicit CallScope(ma_ctx* ctx);  // pointer classifications are remembered until the scope ends
    ~CallScope();
    CallScope(const CallScope&) = delete;
    CallScope& operator=(const CallScope&) = delete;

    // Device-reachable alias of `bytes` bytes at `host_or_dev` (copied in when pageable).
    ma_status in(const void* host_or_dev, size_t bytes, const void** out);
    // Device-reachable destination for `bytes` bytes; copied back to `host_or_dev` by finish() when pageable.
    ma_status out(void* host_or_dev, size_t bytes, void** out);
    // Validity bitmap window -> (8-byte aligned word pointer, bit offset of the window's first bit).
    // Reads whole u64 words like the reference (src/structs/bitmask.rs:266-268).
    ma_status in_mask(const uint8_t* bits, size_t bit_offset, size_t len_bits, const uint64_t** out_words,
                      size_t* out_bit_offset);
    // Output bitmap of len_bits bits starting at bit 0: 8*ceil(len_bits/64) bytes are written.
    ma_status out_mask(uint8_t* bits, size_t len_bits, uint64_t** out_words);
    // Synchronises when needed and copies temporaries back.
    ma_status finish();
    bool staged() const { return !temps_.empty(); }

  private:
    struct Temp {
        void* dev;
        void* host_dst;  // nullptr for inputs
        size_t bytes;
    };
    // Temporaries are carved out of a few slabs: a chunked column hands over thousands of small host buffers, and a
    // hipMalloc per buffer would cost more than the copies.
    ma_status carve(size_t bytes, void** out);
    ma_ctx* ctx_;
    std::vector<Temp> temps_;
    std::vector<void*> slabs_;
    bool finished_ = false;  // finish() ran to completion: nothing in flight touches the slabs any more
    char* slab_cur_ = nullptr;
    size_t slab_left_ = 0, slab_next_ = (size_t)1 << 20;
};

inline int grid_for(const ma_ctx* ctx, size_t work_items, int blocks_per_cu = 0) {
    if (ctx->grid_override > 0) {
        size_t g = (size_t)ctx->grid_override < (size_t)kMaxGrid ? (size_t)ctx->grid_override : (size_t)kMaxGrid;
        if (work_items < 1) work_items = 1;
        return (int)(work_items < g ? work_items : g);
    }
    if (blocks_per_cu <= 0) blocks_per_cu = ctx->blocks_per_cu > 0 ? ctx->blocks_per_cu : kDefaultBlocksPerCu;
    size_t cap = (size_t)ctx->num_cus * (size_t)blocks_per_cu;
    if (cap > (size_t)kMaxGrid) cap = kMaxGrid;
    if (work_items < 1) work_items = 1;
    return (int)(work_items < cap ? work_items : cap);
}

// ma_sum_chunks with the float total as a (hi, lo) pair (ma_reduce_batch.hip); any output may be NULL.
ma_status sum_chunks_dd(ma_ctx* ctx, int32_t format_code, size_t n_chunks, const void* const* chunk_data, const size_t* chunk_lens,
                        const uint8_t* const* chunk_masks, const size_t* chunk_mask_offsets, double* out_hi, double* out_lo,
                        int64_t* out_sum_i64, uint64_t* out_valid_count);

// All chunk pairs of a SuperArray (op) SuperArray in one launch (ma_superarray.hip). smode 1 / 2: the left / right operand
// is the scalar whose bits are the low bytes of `sbits` (that side's tables may be NULL).
ma_status route_batched(ma_ctx* ctx, int32_t format_code, int32_t op, size_t n_chunks, const void* const* lhs_data,
                        const size_t* lens, const uint8_t* const* lhs_masks, const void* const* rhs_data,
                        const uint8_t* const* rhs_masks, const uint8_t* override_mask, void* const* out_data,
                        uint8_t* const* out_masks, int32_t* out_has_mask, int smode = 0, uint64_t sbits = 0);

// Chunk-pipelined staging (ma_pipeline.hip). An entry point whose rows are independent describes its operands and
// hands over a function that enqueues the kernels of one tile on ctx->stream; run_tiled moves the pageable operands
// through a ring of device buffers (H2D of tile k+1, kernels of tile k and D2H of tile k-1 overlap) and returns when
// every result has landed. ptrs[i] is operand i's device-reachable address of the tile's first row.
constexpr int kMaxPipeOperands = 4;
struct PipeOperand {
    const void* in;     // source rows (inputs), or nullptr
    void* out;          // destination rows (outputs), or nullptr
    size_t elem_bytes;
    bool staged;        // pageable host memory: goes through the ring; otherwise used in place
};
typedef ma_status (*TileFn)(void* user, size_t row0, size_t rows, void* const* ptrs);
ma_status run_tiled(ma_ctx* ctx, size_t n_rows, size_t tile_rows, const PipeOperand* ops, int n_ops, TileFn fn,
                    void* user);
void pipe_destroy(ma_ctx* ctx);

// `bytes` of device scratch owned by the context (256-byte aligned). Valid until the next ctx_scratch call on this
// context; the caller holds ctx->mu and enqueues every use on ctx->stream, so successive users are stream-ordered.
ma_status ctx_scratch(ma_ctx* ctx, size_t bytes, void** out);
// Enqueues the copy of a small host table (descriptors living in the caller's frame) to `dev_dst` on ctx->stream
// WITHOUT waiting for the stream: the bytes are taken into one of the context's pinned staging buffers (four, used in turn) before the
// call returns. (A buffer is re-used only after the copy — or the kernel that read it in place — issued from it four uploads ago has finished.) The caller holds
// the context.
ma_status upload_table(ma_ctx* ctx, const void* src, size_t bytes, void* dev_dst);
// The same in two steps for tables large enough that the extra copy shows (100 000 chunk descriptors): table_begin hands
// out the pinned staging buffer to BUILD the table in, table_commit enqueues its copy. Nothing else may stage a table on
// this context in between; a begin without a commit is harmless.
ma_status table_begin(ma_ctx* ctx, size_t bytes, void** out_host);
ma_status table_commit(ma_ctx* ctx, const void* host, size_t bytes, void* dev_dst);
// Instead of table_commit: no copy at all — kernels read the table where it was built, in the pinned staging buffer
// (*out_dev_alias is its device-side address). For tables a kernel walks ONCE with wave-uniform loads issued a chunk
// ahead of their use (one PCIe read per descriptor, hidden behind the chunk's rows), where the copy in front of the
// launch — 0.1 ms for 60 000 descriptors, on the stream — costs more than it saves. table_release(slot) after the last
// launch that reads the table: the staging buffer is re-used only once that launch has finished.
ma_status table_commit_mapped(ma_ctx* ctx, const void* host, const void** out_dev_alias, int* out_slot);
ma_status table_release(ma_ctx* ctx, int slot);
// Releases a mapped table's slot on EVERY way out of the scope that launched on it (an early MA_TRY / MA_HIP return past
// table_commit_mapped would otherwise leave the slot marked free while a launched kernel may still read it).
struct TableSlotGuard {
    ma_ctx* ctx;
    int slot = -1;
    explicit TableSlotGuard(ma_ctx* c) : ctx(c) {}
    ~TableSlotGuard() {
        if (slot >= 0) (void)table_release(ctx, slot);
    }
    TableSlotGuard(const TableSlotGuard&) = delete;
    TableSlotGuard& operator=(const TableSlotGuard&) = delete;
};

// Instead of table_commit / table_commit_mapped, for a table that a SHORT kernel walks (60 000 chunk descriptors under a 0.29-ms
// scan): the table built by table_begin goes to a device buffer of the context on the context's UPLOAD stream — beside whatever
// ctx->stream is still running, and in pieces while the host is still writing the rest; the host waits for the last piece
// (microseconds) and launches. Read in place, every descriptor is a PCIe read of its own (200 M/s for 32-KB chunks) whose latency depends
// on where the pinned buffer sits: the same kernel took 288 us on one staging slot and 310-370 on the next
// (profiles/r06_column_waves.md); copied on ctx->stream, the 40-us copy sits between the kernels.
//   TableUpload up(ctx); up.begin(host, bytes); [up.push(bytes_final_so_far) ...]; up.finish(&dev); launches on ctx->stream;
// the destructor records the readers' event (the buffer is overwritten only behind it) on every way out.
struct TableUpload {
    ma_ctx* ctx;
    const char* host = nullptr;
    size_t total = 0, sent = 0;
    int dslot = -1;
    bool handed_over = false;
    explicit TableUpload(ma_ctx* c) : ctx(c) {}
    ~TableUpload();
    TableUpload(const TableUpload&) = delete;
    TableUpload& operator=(const TableUpload&) = delete;
    ma_status begin(const void* host_table, size_t bytes);  // host_table: what table_begin handed out
    ma_status push(size_t upto);                            // bytes [sent, upto) are final: their copy starts now
    ma_status finish(const void** out_dev);                 // the rest, and waits for it: launches may follow at once
};

// Host side, used where small Arrow batches are gathered into pinned tiles (ma_stream.hip, ma_stream_op.hip).
// n bits of `src` starting at bit `s` (a buffer of `src_bytes` bytes; nullptr = all ones) appended to `dst` at bit `p`;
// the words of `dst` from bit p on are zero.
inline void append_bits(uint64_t* dst, size_t p, const uint8_t* src, size_t src_bytes, size_t s, size_t n) {
    for (size_t i = 0; i < n; i += 64) {
        const size_t take = n - i < 64 ? n - i : 64;
        uint64_t w = ~(uint64_t)0;
        if (src) {
            const size_t b = (s + i) >> 3;
            const unsigned sh = (unsigned)((s + i) & 7);
            uint64_t lo = 0;
            const size_t avail = src_bytes > b ? src_bytes - b : 0;
            memcpy(&lo, src + b, avail < 8 ? avail : 8);
            w = lo >> sh;
            if (sh && avail > 8) w |= (uint64_t)src[b + 8] << (64 - sh);
        }
        if (take < 64) w &= (((uint64_t)1) << take) - 1;
        const size_t q = p + i, wi = q >> 6;
        const unsigned ps = (unsigned)(q & 63);
        dst[wi] |= w << ps;
        if (ps && take > 64 - ps) dst[wi + 1] |= w >> (64 - ps);
    }
}

// A reduction record as it is exchanged between GPUs: 8 x u64 (64 bytes) — [0] integer sum, [1] integer valid count,
// [2] f64 hi bits, [3] f64 lo bits, [4] float valid count, [5..7] unused.
constexpr size_t kRecordWords = 8;
// Enqueues the ordered fold of gathered records on ctx->stream (ma_reduce_batch.hip): column c = records at
// rec + c * kRecordWords + r * stride_words for r in [0, n_records); out = 4 x u64 per column. Device-reachable buffers;
// the caller holds ctx->mu.
ma_status enqueue_fold_columns(ma_ctx* ctx, const uint64_t* rec, size_t n_records, size_t stride_words, size_t n_columns,
                               uint64_t* out);

// Completes a call: in sync mode waits for the stream. Returns MA_ERR_DEVICE on failure.
ma_status end_call(ma_ctx* ctx, CallScope& scope);

}  // namespace ma
