// Context, memory and error plumbing of libminarrow_hip.so, plus the synthetic-input generators.
// C ABI: include/minarrow_hip.h.
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <map>
#include <unordered_set>
#include <unordered_map>
#include <utility>
#include <vector>

#include <dlfcn.h>

#include "ma_common.hpp"

namespace ma {

static thread_local char g_err[4096] = "";

// Pinned blocks of 1 MiB and more are recycled: hipHostMalloc pins pages at a few GB/s (a 256-MiB Vec64 costs ~40 ms to
// allocate and as much to free), which would make an allocator built on it 50x slower than malloc for exactly the
// columns the device path is for. Sizes are rounded up to one of eight steps per power of two (at most 12.5 % of slack);
// freed blocks wait in per-size lists until the cache holds more than the limit. Smaller blocks go straight to
// hipHostMalloc / hipHostFree.
namespace {
struct PinnedPool {
    std::mutex mu;
    std::map<size_t, std::vector<void*>> parked;   // rounded size -> free blocks of that size
    std::unordered_set<void*> parked_set;          // the same blocks by address: a second free of one is refused
    std::unordered_map<void*, size_t> live;        // blocks handed out by the pool -> their rounded size
    std::unordered_map<void*, float> write_gbps;   // device pools: measured write rate of a block (ma_dev_alloc_output)
    size_t cached_bytes = 0;
    size_t limit_bytes = (size_t)2 << 30;
};
constexpr size_t kPoolMinBytes = (size_t)1 << 20;
size_t pool_size_of(size_t bytes, size_t min_bytes = kPoolMinBytes) {  // bytes >= min_bytes
    size_t top = min_bytes;
    while ((top << 1) != 0 && (top << 1) <= bytes) top <<= 1;  // largest power of two <= bytes
    const size_t step = top >> 3;
    return ((bytes + step - 1) / step) * step;
}
// Pinned HOST blocks are recycled from 4 KiB on: a result slab of an 8192-row record batch is ~130 KiB, and a
// hipHostMalloc + hipHostFree pair per batch (~250 us) was most of the 366 us a batch cost the stream operator.
constexpr size_t kPinnedPoolMinBytes = (size_t)4 << 10;
// MINARROW_HIP_PINNED_POOL_BYTES / MINARROW_HIP_DEV_POOL_BYTES: cache limits, read once (0 = no caching).
size_t env_bytes(const char* name, size_t fallback) {
    const char* v = getenv(name);
    if (!v || !*v) return fallback;
    char* end = nullptr;
    unsigned long long x = strtoull(v, &end, 10);
    return end && *end == '\0' ? (size_t)x : fallback;
}
PinnedPool& pinned_pool() {
    static PinnedPool* pool = [] {  // intentionally leaked: frees may arrive during process teardown
        PinnedPool* p = new PinnedPool();
        p->limit_bytes = env_bytes("MINARROW_HIP_PINNED_POOL_BYTES", p->limit_bytes);
        return p;
    }();
    return *pool;
}
// The same cache for device blocks, one per device: hipFree synchronises the WHOLE device (~160 us, and every other
// stream stalls with it), so a chain of resident results that frees its temporaries would serialise all contexts.
constexpr int kMaxPooledDevices = 64;
PinnedPool& device_pool(int device) {
    static PinnedPool* pools = [] {
        PinnedPool* p = new PinnedPool[kMaxPooledDevices];
        const size_t limit = env_bytes("MINARROW_HIP_DEV_POOL_BYTES", (size_t)16 << 30);
        for (int i = 0; i < kMaxPooledDevices; ++i) p[i].limit_bytes = limit;
        return p;
    }();
    return pools[device];
}

// Releases every parked block of `device` back to the runtime (the device must be current). Returns the bytes freed.
size_t dev_pool_flush(int device) {
    if (device < 0 || device >= kMaxPooledDevices) return 0;
    PinnedPool& pool = device_pool(device);
    std::vector<void*> victims;
    size_t bytes = 0;
    {
        std::lock_guard<std::mutex> plock(pool.mu);
        for (auto& kv : pool.parked) {
            victims.insert(victims.end(), kv.second.begin(), kv.second.end());
            kv.second.clear();
        }
        bytes = pool.cached_bytes;
        pool.cached_bytes = 0;
        pool.parked_set.clear();
        for (void* v : victims) pool.write_gbps.erase(v);
    }
    for (void* v : victims) (void)hipFree(v);
    return bytes;
}

// hipMalloc that, when HBM is full of parked blocks, releases them and tries once more. Every internal device
// allocation goes through here (scratch, staging rings, context state), not only ma_dev_alloc.
hipError_t dev_malloc_retry(int device, void** out, size_t bytes) {
    hipError_t e = hipMalloc(out, bytes);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        if (dev_pool_flush(device) > 0) e = hipMalloc(out, bytes);
    }
    return e;
}

// hipMalloc / hipFree through the device's block cache (the caller has made the device current and, for a free, has
// made sure nothing in flight still touches the block).
hipError_t dev_block_alloc(int device, size_t bytes, void** out) {
    *out = nullptr;
    if (bytes < kPoolMinBytes || device < 0 || device >= kMaxPooledDevices) return dev_malloc_retry(device, out, bytes == 0 ? 64 : bytes);
    PinnedPool& pool = device_pool(device);
    const size_t rounded = pool_size_of(bytes);
    {
        std::lock_guard<std::mutex> plock(pool.mu);
        auto it = pool.parked.find(rounded);
        if (it != pool.parked.end() && !it->second.empty()) {
            *out = it->second.back();
            it->second.pop_back();
            pool.parked_set.erase(*out);
            pool.cached_bytes -= rounded;
            pool.live.emplace(*out, rounded);
            return hipSuccess;
        }
    }
    hipError_t e = dev_malloc_retry(device, out, rounded);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> plock(pool.mu);
    pool.live.emplace(*out, rounded);
    return hipSuccess;
}

hipError_t dev_block_free(int device, void* ptr) {
    if (device >= 0 && device < kMaxPooledDevices) {
        PinnedPool& pool = device_pool(device);
        std::lock_guard<std::mutex> plock(pool.mu);
        if (pool.parked_set.count(ptr)) return hipErrorInvalidValue;  // already parked: a double free
        auto it = pool.live.find(ptr);
        if (it != pool.live.end()) {
            const size_t rounded = it->second;
            pool.live.erase(it);
            if (pool.cached_bytes + rounded <= pool.limit_bytes) {
                pool.parked[rounded].push_back(ptr);
                pool.parked_set.insert(ptr);
                pool.cached_bytes += rounded;
                return hipSuccess;
            }
        }
        pool.write_gbps.erase(ptr);
    }
    return hipFree(ptr);
}

// Takes parked blocks out of `pool` until at most keep_bytes stay, largest first.
std::vector<void*> pool_take_victims(PinnedPool& pool, size_t keep_bytes) {
    std::vector<void*> victims;
    std::lock_guard<std::mutex> plock(pool.mu);
    pool.limit_bytes = keep_bytes;
    for (auto it = pool.parked.rbegin(); it != pool.parked.rend() && pool.cached_bytes > keep_bytes; ++it)
        while (!it->second.empty() && pool.cached_bytes > keep_bytes) {
            victims.push_back(it->second.back());
            pool.parked_set.erase(it->second.back());
            pool.write_gbps.erase(it->second.back());
            it->second.pop_back();
            pool.cached_bytes -= it->first;
        }
    return victims;
}
}  // namespace

hipError_t device_malloc(int device, void** out, size_t bytes) { return dev_malloc_retry(device, out, bytes); }
// Through the device's block cache (blocks of 1 MiB and more are parked on free instead of stalling the device in hipFree).
hipError_t device_block_alloc(int device, void** out, size_t bytes) { return dev_block_alloc(device, bytes, out); }
hipError_t device_block_free(int device, void* ptr) { return dev_block_free(device, ptr); }

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

ma_status hip_fail(hipError_t e, const char* what, const char* file, int line) {
    set_error("HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    // A failed launch leaves a sticky "last error"; clear it so later calls report their own.
    (void)hipGetLastError();
    return e == hipErrorNoDevice ? MA_ERR_NO_DEVICE : MA_ERR_DEVICE;
}

// A chunked column hands over thousands of pointers into a few allocations (122 071 chunk pairs for 10^9 rows at the
// reference's default 8192-row chunking), so the address ranges of the device allocations seen are remembered — but
// only for the duration of ONE ABI call (the lifetime of its CallScope): while a call runs its caller cannot free
// anything, whereas a range remembered across calls could have been freed behind the library's back (a torch tensor,
// a hipFree by the host) and handed out again as host memory, which a kernel must never be pointed at.
namespace {
constexpr int kRanges = 8;
// One thread-local object (one TLS lookup per call in a shared library, not one per variable); the range that matched
// last is tried first — a chunked column's pointers arrive in runs from the same few allocations.
struct RangeCache {
    uintptr_t lo[kRanges] = {0}, hi[kRanges] = {0};
    unsigned next_slot = 0;
    unsigned last_hit = 0;
    int scope_depth = 0;
};
thread_local RangeCache t_ranges;

void forget_ranges() {
    RangeCache& c = t_ranges;
    for (int i = 0; i < kRanges; ++i) c.lo[i] = c.hi[i] = 0;
    c.next_slot = 0;
    c.last_hit = 0;
}
}  // namespace

PtrKind pointer_kind(const void* p) {
    RangeCache& c = t_ranges;
    const bool remember = c.scope_depth > 0;
    const uintptr_t addr = (uintptr_t)p;
    if (remember) {
        if (addr >= c.lo[c.last_hit] && addr < c.hi[c.last_hit]) return kDevice;
        for (unsigned i = 0; i < (unsigned)kRanges; ++i)
            if (addr >= c.lo[i] && addr < c.hi[i]) {
                c.last_hit = i;
                return kDevice;
            }
    }
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return kPageable;
    }
    switch (attr.type) {
        case hipMemoryTypeDevice: {
            hipDeviceptr_t base = nullptr;
            size_t size = 0;
            if (remember) {
                if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)p) == hipSuccess && size) {
                    c.lo[c.next_slot] = (uintptr_t)base;
                    c.hi[c.next_slot] = (uintptr_t)base + size;
                    c.last_hit = c.next_slot;
                    c.next_slot = (c.next_slot + 1) % kRanges;
                } else {
                    (void)hipGetLastError();
                }
            }
            return kDevice;
        }
        case hipMemoryTypeArray:
            return kDevice;
        case hipMemoryTypeHost:
            return kPinned;
        case hipMemoryTypeManaged:
        case hipMemoryTypeUnified:
            return kManaged;
        default:
            return kPageable;
    }
}

bool known_device_range(const void* p, uintptr_t* lo, uintptr_t* hi) {
    RangeCache& c = t_ranges;
    const uintptr_t addr = (uintptr_t)p;
    for (int i = 0; i < kRanges; ++i)
        if (addr >= c.lo[i] && addr < c.hi[i]) {
            *lo = c.lo[i];
            *hi = c.hi[i];
            return true;
        }
    return false;
}

CallScope::CallScope(ma_ctx* ctx) : ctx_(ctx) {
    if (t_ranges.scope_depth++ == 0) forget_ranges();
}

CallScope::~CallScope() {
    // A parked slab can be handed to another context at once. finish() leaves the stream drained; an early error return
    // (a later operand failing to stage, a validation failure after kernels were enqueued) does not: wait here, so that
    // nothing in flight still reads or writes the slabs and the caller gets its "synchronous" call back quiescent.
    if (!slabs_.empty() && !finished_) (void)hipStreamSynchronize(ctx_->stream);
    for (void* slab : slabs_) (void)dev_block_free(ctx_->device, slab);
    if (--t_ranges.scope_depth == 0 || !temps_.empty()) forget_ranges();
}

ma_status CallScope::carve(size_t bytes, void** out) {
    const size_t need = (bytes + 255) & ~(size_t)255;  // every temporary starts on a 256-byte boundary
    if (need > slab_left_) {
        size_t want = need > slab_next_ ? need : slab_next_;
        void* slab = nullptr;
        MA_HIP(dev_block_alloc(ctx_->device, want, &slab));
        slabs_.push_back(slab);
        if (need >= slab_next_) {  // a large operand gets its own allocation; the current slab stays open
            *out = slab;
            return MA_OK;
        }
        slab_cur_ = (char*)slab;
        slab_left_ = want;
        if (slab_next_ < ((size_t)64 << 20)) slab_next_ *= 4;
    }
    *out = slab_cur_;
    slab_cur_ += need;
    slab_left_ -= need;
    return MA_OK;
}

ma_status CallScope::in(const void* p, size_t bytes, const void** out) {
    *out = p;
    if (bytes == 0 || p == nullptr) return MA_OK;
    if (pointer_kind(p) != kPageable) return MA_OK;
    MA_NO_CAPTURE(ctx_, "staging a pageable host input");
    void* d = nullptr;
    MA_TRY(carve(bytes, &d));
    temps_.push_back({d, nullptr, bytes});
    MA_HIP(hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, ctx_->stream));
    *out = d;
    return MA_OK;
}

ma_status CallScope::out(void* p, size_t bytes, void** out) {
    *out = p;
    if (bytes == 0 || p == nullptr) return MA_OK;
    if (pointer_kind(p) != kPageable) return MA_OK;
    MA_NO_CAPTURE(ctx_, "staging a pageable host output");
    void* d = nullptr;
    MA_TRY(carve(bytes, &d));
    temps_.push_back({d, p, bytes});
    *out = d;
    return MA_OK;
}

ma_status CallScope::in_mask(const uint8_t* bits, size_t bit_offset, size_t len_bits, const uint64_t** out_words,
                             size_t* out_bit_offset) {
    *out_words = nullptr;
    *out_bit_offset = 0;
    if (bits == nullptr || len_bits == 0) return MA_OK;
    if (pointer_kind(bits) != kPageable) {
        // Re-base onto the enclosing 8-byte aligned word; the few bytes in front belong to the same
        // aligned word and are never interpreted.
        uintptr_t addr = (uintptr_t)bits;
        uintptr_t base = addr & ~(uintptr_t)7;
        *out_words = (const uint64_t*)base;
        *out_bit_offset = bit_offset + (size_t)(addr - base) * 8;
        return MA_OK;
    }
    // Pageable: copy exactly the bytes that hold the window into a zero-padded word buffer.
    MA_NO_CAPTURE(ctx_, "staging a pageable host bitmap");
    size_t first_byte = bit_offset >> 3;
    size_t end_byte = (bit_offset + len_bits + 7) >> 3;
    size_t nbytes = end_byte - first_byte;
    size_t padded = ((nbytes + 7) & ~(size_t)7) + 8;
    void* d = nullptr;
    MA_TRY(carve(padded, &d));
    temps_.push_back({d, nullptr, padded});
    MA_HIP(hipMemsetAsync(d, 0, padded, ctx_->stream));
    MA_HIP(hipMemcpyAsync(d, bits + first_byte, nbytes, hipMemcpyHostToDevice, ctx_->stream));
    *out_words = (const uint64_t*)d;
    *out_bit_offset = bit_offset & 7;
    return MA_OK;
}

ma_status CallScope::out_mask(uint8_t* bits, size_t len_bits, uint64_t** out_words) {
    *out_words = nullptr;
    if (bits == nullptr || len_bits == 0) return MA_OK;
    MA_REQUIRE(((uintptr_t)bits & 7) == 0, MA_ERR_INVALID_ARGUMENT,
               "output bitmap must be 8-byte aligned (got %p)", (const void*)bits);
    size_t bytes = ((len_bits + 63) / 64) * 8;
    void* d = nullptr;
    MA_TRY(out(bits, bytes, &d));
    *out_words = (uint64_t*)d;
    return MA_OK;
}

// The wait at the end of a synchronous call. hipStreamSynchronize's wake-up costs ~10 us — most of a small call — so the
// stream is first asked to stamp a pinned word when it has finished everything before (hipStreamWriteValue64: a
// command-processor write, no kernel) and the host polls that word for up to poll_us; a long call falls through to the
// blocking wait, and so does a faulted one (whose stamp never comes), which then reports its error.
ma_status stream_wait(ma_ctx* ctx) {
    if (ctx->poll_us > 0 && !ctx->capturing && ctx->result && !(tuning_variant(ctx) & 128)) {
        volatile uint64_t* done = (volatile uint64_t*)&ctx->result[2].cnt;
        const uint64_t seq = ++ctx->result_seq;
        if (hipStreamWriteValue64(ctx->stream, (void*)done, seq, 0) == hipSuccess) {
            const auto t0 = std::chrono::steady_clock::now();
            for (unsigned spins = 0;; ++spins) {
                if (*done == seq) {
                    std::atomic_thread_fence(std::memory_order_acquire);
                    return MA_OK;
                }
                if ((spins & 63) == 63 &&
                    std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() >= ctx->poll_us)
                    break;
                __builtin_ia32_pause();
            }
        } else {
            (void)hipGetLastError();
        }
    }
    MA_HIP(hipStreamSynchronize(ctx->stream));
    return MA_OK;
}

ma_status CallScope::finish() {
    bool need_sync = !is_async(ctx_) || !temps_.empty();
    bool any_out = false;
    for (auto& t : temps_) {
        if (t.host_dst) {
            MA_HIP(hipMemcpyAsync(t.host_dst, t.dev, t.bytes, hipMemcpyDeviceToHost, ctx_->stream));
            any_out = true;
        }
    }
    (void)any_out;
    if (need_sync) {
        MA_TRY(stream_wait(ctx_));
        finished_ = true;
    }
    return MA_OK;
}

ma_status ctx_scratch(ma_ctx* ctx, size_t bytes, void** out) {
    *out = nullptr;
    if (bytes > ctx->scratch_bytes) {
        if (ctx->scratch) {
            MA_HIP(hipStreamSynchronize(ctx->stream));  // the previous user may still be running
            forget_ranges();
            MA_HIP(hipFree(ctx->scratch));
            ctx->scratch = nullptr;
            ctx->scratch_bytes = 0;
        }
        size_t want = bytes + bytes / 2;
        want = (want + 4095) & ~(size_t)4095;
        MA_HIP(dev_malloc_retry(ctx->device, &ctx->scratch, want));
        ctx->scratch_bytes = want;
    }
    *out = ctx->scratch;
    return MA_OK;
}

ma_status table_begin(ma_ctx* ctx, size_t bytes, void** out_host) {
    *out_host = nullptr;
    if (bytes == 0) return MA_OK;
    // Which slot: the first one (from the rotation point) whose last user has finished AND that is large enough; else any
    // finished one (never-used slots first: they cost one allocation, an outgrown one costs a replacement); only when all 16
    // are still being read does the host wait. A host whose GPU keeps up therefore cycles through two or three slots and
    // the others are never allocated.
    int pick = -1, spare = -1;
    for (int i = 0; i < ma_ctx::kTableSlots && pick < 0; ++i) {
        const int k = (ctx->table_next + i) % ma_ctx::kTableSlots;
        if (ctx->table_mapped[k]) continue;  // a launched kernel reads it in place; its event is only recorded by table_release
        if (ctx->table_busy[k]) {
            const hipError_t q = hipEventQuery(ctx->table_ev[k]);
            if (q == hipSuccess) {
                ctx->table_busy[k] = false;
            } else {
                (void)hipGetLastError();
                continue;
            }
        }
        if (ctx->table_stage_bytes[k] >= bytes) pick = k;
        else if (spare < 0 || (ctx->table_stage[spare] != nullptr && ctx->table_stage[k] == nullptr)) spare = k;
    }
    if (pick < 0) pick = spare;
    if (pick < 0) {
        for (int i = 0; i < ma_ctx::kTableSlots && pick < 0; ++i) {  // the oldest slot whose event HAS been recorded
            const int k = (ctx->table_next + i) % ma_ctx::kTableSlots;
            if (!ctx->table_mapped[k]) pick = k;
        }
        MA_REQUIRE(pick >= 0, MA_ERR_DEVICE, "every staging slot is held by a table read in place (table_commit_mapped without table_release)");
        MA_HIP(hipEventSynchronize(ctx->table_ev[pick]));
        ctx->table_busy[pick] = false;
    }
    const int k = pick;
    ctx->table_cur = k;
    if (!ctx->table_ev[k]) MA_HIP(hipEventCreateWithFlags(&ctx->table_ev[k], hipEventDisableTiming));
    if (bytes > ctx->table_stage_bytes[k]) {
        // A slot that is too small is replaced, not freed: hipHostFree drains the whole device, and a context that streams
        // chunk lists of mixed sizes through its slots would pay that in the middle of its pipeline (the old buffers go when
        // the context does). New slots are sized for the largest table the context has seen so far.
        if (ctx->table_stage[k]) ctx->table_garbage.push_back(ctx->table_stage[k]);
        ctx->table_stage[k] = nullptr;
        ctx->table_stage_bytes[k] = 0;
        if (bytes > ctx->table_high_water) ctx->table_high_water = bytes;
        const size_t base = ctx->table_high_water;
        const size_t want = (base + base / 2 + 4095) & ~(size_t)4095;
        // Mapped: kernels read some tables in place (table_commit_mapped) — asked for, not left to the runtime's implicit mapping
        MA_HIP(hipHostMalloc(&ctx->table_stage[k], want, hipHostMallocPortable | hipHostMallocMapped));
        ctx->table_stage_bytes[k] = want;
    }
    *out_host = ctx->table_stage[k];
    return MA_OK;
}

ma_status table_commit(ma_ctx* ctx, const void* host, size_t bytes, void* dev_dst) {
    if (bytes == 0) return MA_OK;
    const int k = ctx->table_cur;
    MA_REQUIRE(k >= 0 && host == ctx->table_stage[k], MA_ERR_INVALID_ARGUMENT, "table_commit without table_begin");
    ctx->table_next = (k + 1) % ma_ctx::kTableSlots;
    ctx->table_cur = -1;
    MA_HIP(hipMemcpyAsync(dev_dst, host, bytes, hipMemcpyHostToDevice, ctx->stream));
    MA_HIP(hipEventRecord(ctx->table_ev[k], ctx->stream));
    ctx->table_busy[k] = true;
    return MA_OK;
}

ma_status table_commit_mapped(ma_ctx* ctx, const void* host, const void** out_dev_alias, int* out_slot) {
    const int k = ctx->table_cur;
    MA_REQUIRE(k >= 0 && host == ctx->table_stage[k], MA_ERR_INVALID_ARGUMENT, "table_commit_mapped without table_begin");
    void* alias = nullptr;
    MA_HIP(hipHostGetDevicePointer(&alias, ctx->table_stage[k], 0));
    ctx->table_next = (k + 1) % ma_ctx::kTableSlots;
    ctx->table_cur = -1;
    ctx->table_busy[k] = true;
    ctx->table_mapped[k] = true;  // until table_release records the event: table_begin skips the slot without asking the event
    *out_dev_alias = alias;
    *out_slot = k;
    return MA_OK;
}

ma_status table_release(ma_ctx* ctx, int slot) {
    const hipError_t e = hipEventRecord(ctx->table_ev[slot], ctx->stream);
    ctx->table_mapped[slot] = false;  // recorded or not, the reader has been launched: the event (or a drain) covers it from here
    ctx->table_busy[slot] = true;
    if (e != hipSuccess) return hip_fail(e, "hipEventRecord(table_release)", __FILE__, __LINE__);
    return MA_OK;
}

ma_status TableUpload::begin(const void* host_table, size_t bytes) {
    const int k = ctx->table_cur;
    MA_REQUIRE(k >= 0 && host_table == ctx->table_stage[k], MA_ERR_INVALID_ARGUMENT, "TableUpload::begin without table_begin");
    if (!ctx->upload_stream) MA_HIP(hipStreamCreateWithFlags(&ctx->upload_stream, hipStreamNonBlocking));
    const int j = ctx->dev_table_next;
    ctx->dev_table_next = (j + 1) % ma_ctx::kDevTables;
    if (!ctx->dev_table_up[j]) MA_HIP(hipEventCreateWithFlags(&ctx->dev_table_up[j], hipEventDisableTiming));
    if (!ctx->dev_table_read[j]) MA_HIP(hipEventCreateWithFlags(&ctx->dev_table_read[j], hipEventDisableTiming));
    if (bytes > ctx->dev_table_bytes[j]) {
        // replaced, not freed: a kernel launched four calls ago may still read the old one, and hipFree drains the device
        if (ctx->dev_table[j]) ctx->dev_table_garbage.push_back(ctx->dev_table[j]);
        ctx->dev_table[j] = nullptr;
        ctx->dev_table_bytes[j] = 0;
        ctx->dev_table_has_reader[j] = false;
        const size_t base = bytes > ctx->table_high_water ? bytes : ctx->table_high_water;
        const size_t want = (base + base / 2 + 4095) & ~(size_t)4095;
        MA_HIP(dev_malloc_retry(ctx->device, &ctx->dev_table[j], want));
        ctx->dev_table_bytes[j] = want;
    }
    // The buffer's last readers were launched kDevTables calls ago. The HOST waits for them (normally they are long done): that
    // is also what keeps a host that describes tables faster than the GPU scans them from running ahead without bound — left
    // to a device-side wait it queued ten calls' copies, events and waits and then stood still for 8 ms at a time while the
    // GPU ran dry (profiles/r06_column_waves.md).
    if (ctx->dev_table_has_reader[j]) MA_HIP(hipEventSynchronize(ctx->dev_table_read[j]));
    host = (const char*)host_table;
    total = bytes;
    sent = 0;
    dslot = j;
    return MA_OK;
}

ma_status TableUpload::push(size_t upto) {
    if (dslot < 0 || upto <= sent) return MA_OK;
    if (upto > total) upto = total;
    MA_HIP(hipMemcpyAsync((char*)ctx->dev_table[dslot] + sent, host + sent, upto - sent, hipMemcpyHostToDevice, ctx->upload_stream));
    sent = upto;
    return MA_OK;
}

ma_status TableUpload::finish(const void** out_dev) {
    MA_REQUIRE(dslot >= 0, MA_ERR_INVALID_ARGUMENT, "TableUpload::finish without begin");
    MA_TRY(push(total));
    const int k = ctx->table_cur;
    // The HOST waits for the last piece (~15 us; the earlier ones ran while the table was still being written), not ctx->stream:
    // a cross-stream wait in front of every launch is a barrier packet the queue works through only after the kernel before it
    // has ended — 10-12 us between back-to-back scans where a plain launch leaves 4 (profiles/r06_column_waves.md). With the
    // GPU still busy with the call before, the wait costs the pipeline nothing.
    MA_HIP(hipEventRecord(ctx->dev_table_up[dslot], ctx->upload_stream));
    MA_HIP(hipEventSynchronize(ctx->dev_table_up[dslot]));
    ctx->table_busy[k] = false;  // copied: the pinned staging slot is free
    ctx->table_next = (k + 1) % ma_ctx::kTableSlots;
    ctx->table_cur = -1;
    handed_over = true;
    *out_dev = ctx->dev_table[dslot];
    return MA_OK;
}

TableUpload::~TableUpload() {
    if (dslot < 0) return;
    if (!handed_over) {
        // begun, never finished (an error on the way): pieces may be in flight out of the pinned slot — wait for them
        if (sent > 0 && hipStreamSynchronize(ctx->upload_stream) != hipSuccess) (void)hipGetLastError();
        return;
    }
    // behind the last launch that reads the buffer (launched or not: the event covers whatever the stream holds by now)
    if (hipEventRecord(ctx->dev_table_read[dslot], ctx->stream) == hipSuccess) ctx->dev_table_has_reader[dslot] = true;
    else (void)hipGetLastError();
}

ma_status upload_table(ma_ctx* ctx, const void* src, size_t bytes, void* dev_dst) {
    if (bytes == 0) return MA_OK;
    void* host = nullptr;
    MA_TRY(table_begin(ctx, bytes, &host));
    memcpy(host, src, bytes);
    return table_commit(ctx, host, bytes, dev_dst);
}

ma_status end_call(ma_ctx* ctx, CallScope& scope) {
    (void)ctx;
    return scope.finish();
}

// ---------------------------------------------------------------------------------------------
// Entry-point guard: the context, or a lane of it (ma_common.hpp)
// ---------------------------------------------------------------------------------------------
namespace {
struct Held {
    ma_ctx* root;
    ma_ctx* lane;
};
constexpr int kMaxHeld = 8;
thread_local Held t_held[kMaxHeld];
thread_local int t_n_held = 0;
thread_local int t_nosync = 0;
thread_local uint64_t t_entries = 0;
}  // namespace

uint64_t entries_by_this_thread() { return t_entries; }

NoSync::NoSync() { ++t_nosync; }
NoSync::~NoSync() { --t_nosync; }
bool nosync_active() { return t_nosync > 0; }

ma_status make_lane(ma_ctx* root, ma_ctx** out, int cls = 0);  // defined with the context constructor below

Enter::Enter(ma_ctx*& ctx, bool primary_only) {
    if (ctx == nullptr) return;  // the entry point reports the NULL itself
    ma_ctx* root = ctx->parent ? ctx->parent : ctx;
    root->calls.fetch_add(1, std::memory_order_relaxed);
    ++t_entries;
    for (int i = 0; i < t_n_held; ++i)
        if (t_held[i].root == root) {  // a composed entry point calling its parts: same lane, no second lock
            ctx = t_held[i].lane;
            return;
        }
    ma_ctx* lane = root;
    if (!root->mu.try_lock()) {
        // Busy in another thread. Ordering on the context's one stream is part of the contract for async, capturing and
        // borrowed-stream contexts (and for calls tied to that stream): wait. Otherwise run beside it on a lane.
        const bool may_fan_out = !primary_only && !root->async && !root->capturing && root->owns_stream && root->max_lanes > 1;
        lane = nullptr;
        if (may_fan_out) {
            std::lock_guard<std::mutex> ll(root->lanes_mu);
            for (ma_ctx* x : root->lanes)
                if (x->mu.try_lock()) {
                    lane = x;
                    break;
                }
            if (!lane && (int)root->lanes.size() + 1 < root->max_lanes) {
                ma_ctx* fresh = nullptr;
                if (make_lane(root, &fresh) == MA_OK) {
                    fresh->mu.lock();
                    root->lanes.push_back(fresh);
                    lane = fresh;
                }
            }
        }
        if (!lane) {
            root->mu.lock();
            lane = root;
        } else {  // tuning knobs follow the context (plain ints; a concurrent setter is the caller's race, as before)
            lane->variant = root->variant;
            lane->blocks_per_cu = root->blocks_per_cu;
            lane->grid_override = root->grid_override;
            lane->staging_tile_bytes = root->staging_tile_bytes;
            lane->fenced_reduce = root->fenced_reduce;
            lane->poll_us = root->poll_us;
        }
    }
    locked_ = lane;
    ctx = lane;
    if (t_n_held < kMaxHeld) t_held[t_n_held++] = {root, lane};
}

Enter::~Enter() {
    if (!locked_) return;
    for (int i = t_n_held - 1; i >= 0; --i)
        if (t_held[i].lane == locked_) {
            for (int j = i; j + 1 < t_n_held; ++j) t_held[j] = t_held[j + 1];
            --t_n_held;
            break;
        }
    locked_->mu.unlock();
}

ma_status sync_and_check(ma_ctx* ctx) {
    MA_TRY(stream_wait(ctx));
    if (ctx->pending_flags) {
        ctx->pending_flags = false;
        uint32_t flags = 0;
        MA_HIP(hipMemcpy(&flags, ctx->dev_flags, sizeof(flags), hipMemcpyDeviceToHost));
        if (flags) {
            MA_HIP(hipMemset(ctx->dev_flags, 0, sizeof(flags)));
            if (flags & 1u) {
                set_error("integer division by zero in a dense kernel enqueued before this synchronize");
                return MA_ERR_DIVIDE_BY_ZERO;
            }
        }
    }
    return MA_OK;
}

// ---------------------------------------------------------------------------------------------
// Synthetic inputs
// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    // Public-domain SplitMix64 finaliser (Steele, Lea, Flood 2014); `x` is the already-advanced state.
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// Write-only: the whole grid advances as ONE 1-MiB front — one workgroup per CU, one 16-byte store per lane (1 KiB per
// wave) per step — the pattern that writes 6.3-6.5 TB/s wherever the driver placed the block (the tiled
// bursts of the read+write kernels write 5.4-5.7 TB/s on three blocks in four). `head` rows in front of the first
// 16-byte boundary and the < 16 bytes behind the last whole vector go one row per lane.
template <typename T, typename F>
__global__ __launch_bounds__(kBlock) void fill_kernel(T* __restrict__ dst, size_t n, size_t head, F f) {
    constexpr int R = 16 / (int)sizeof(T);
    typedef T V __attribute__((ext_vector_type(R)));
    const size_t tid = (size_t)blockIdx.x * kBlock + threadIdx.x, stride = (size_t)gridDim.x * kBlock;
    for (size_t i = tid; i < head; i += stride) dst[i] = f(i);
    const size_t n_vec = (n - head) / R;
    V* __restrict__ out = (V*)(dst + head);
    for (size_t v = tid; v < n_vec; v += stride) {
        V x;
#pragma unroll
        for (int k = 0; k < R; ++k) x[k] = f(head + v * R + k);
        __builtin_nontemporal_store(x, out + v);
    }
    for (size_t i = head + n_vec * R + tid; i < n; i += stride) dst[i] = f(i);
}

__global__ __launch_bounds__(kBlock) void validity_kernel(uint64_t* __restrict__ words, size_t n_bits, uint64_t seed,
                                                         uint64_t first_index, uint32_t null_every) {
    // One wave64 produces one u64 word per step: lane k owns bit k.
    size_t n_words = (n_bits + 63) / 64;
    size_t wave = ((size_t)blockIdx.x * kBlock + threadIdx.x) >> 6;
    size_t n_waves = ((size_t)gridDim.x * kBlock) >> 6;
    unsigned lane = threadIdx.x & 63;
    for (size_t w = wave; w < n_words; w += n_waves) {
        size_t i = w * 64 + lane;
        bool valid = i < n_bits && (splitmix64(seed + first_index + i) % null_every) != 0;
        unsigned long long word = __ballot(valid);
        if (lane == 0) words[w] = word;
    }
}

template <typename T, typename F>
static ma_status launch_fill(ma_ctx* ctx, T* dst, size_t n, F f) {
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    if (n == 0) return MA_OK;
    MA_REQUIRE(dst != nullptr, MA_ERR_INVALID_ARGUMENT, "dst is NULL");
    MA_REQUIRE(pointer_kind(dst) != kPageable, MA_ERR_INVALID_ARGUMENT,
               "synthetic generators need a device-reachable destination");
    MA_ENTER(ctx);
    MA_HIP(hipSetDevice(ctx->device));
    const uintptr_t mis = (uintptr_t)dst & 15;
    size_t head = mis ? (16 - mis) / sizeof(T) : 0;
    if (head > n) head = n;
    const size_t work = (n * sizeof(T) / 16 + kBlock - 1) / kBlock;
    const int grid = (int)(work < (size_t)ctx->num_cus ? (work ? work : 1) : (size_t)ctx->num_cus);  // one workgroup per CU
    hipLaunchKernelGGL((fill_kernel<T, F>), dim3(grid), dim3(kBlock), 0, ctx->stream, dst, n, head, f);
    MA_HIP(hipGetLastError());
    if (!is_async(ctx)) MA_TRY(stream_wait(ctx));
    return MA_OK;
}

struct IotaI64 {
    int64_t start;
    __device__ int64_t operator()(size_t i) const { return (int64_t)((uint64_t)start + (uint64_t)i); }
};
struct IotaF64 {
    int64_t start;
    __device__ double operator()(size_t i) const { return (double)(int64_t)((uint64_t)start + (uint64_t)i); }
};
struct IotaI32 {
    int32_t start;
    __device__ int32_t operator()(size_t i) const { return (int32_t)((uint32_t)start + (uint32_t)i); }
};
struct IotaF32 {
    int32_t start;
    __device__ float operator()(size_t i) const { return (float)(int32_t)((uint32_t)start + (uint32_t)i); }
};
struct SplitI64 {
    uint64_t seed, first;
    __device__ int64_t operator()(size_t i) const { return (int64_t)splitmix64(seed + first + i); }
};
struct SplitF64 {
    uint64_t seed, first;
    __device__ double operator()(size_t i) const {
        return (double)(splitmix64(seed + first + i) >> 11) * 0x1.0p-52 - 1.0;
    }
};

}  // namespace ma

using namespace ma;

extern "C" {

int32_t ma_abi_version(void) { return MA_ABI_VERSION; }

const char* ma_hip_runtime_path(void) {
    static char path[512] = "";
    if (!path[0]) {
        Dl_info info;
        if (dladdr((void*)&hipGetDeviceCount, &info) && info.dli_fname) snprintf(path, sizeof(path), "%s", info.dli_fname);
    }
    return path;
}

static int physical_device_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

// MINARROW_HIP_DEVICES = "2,3,5": the library's device ordinal i is HIP device list[i] (a per-library
// HIP_VISIBLE_DEVICES that leaves the rest of the process alone). Unset or empty: the identity over all devices.
static const std::vector<int>& device_map() {
    static const std::vector<int>* map = [] {
        auto* m = new std::vector<int>();
        const int n = physical_device_count();
        const char* v = getenv("MINARROW_HIP_DEVICES");
        if (v && *v) {
            const char* p = v;
            while (*p) {
                char* end = nullptr;
                long d = strtol(p, &end, 10);
                if (end == p) break;
                if (d >= 0 && d < n) m->push_back((int)d);
                p = (*end == ',') ? end + 1 : end;
                if (*end != ',' && *end != '\0') break;
            }
        }
        if (m->empty())
            for (int i = 0; i < n; ++i) m->push_back(i);
        return m;
    }();
    return *map;
}

int32_t ma_device_count(void) { return physical_device_count() > 0 ? (int32_t)device_map().size() : 0; }

// The column length below which a host wrapper should keep the reference's CPU kernels. Derived from what was measured
// (INTEGRATION.md §5 has the table): a synchronous call on a resident column costs 11.4-12.1 us up to 65 536 rows and
// 13-15 us at 2^20 (profiles/r02_sync_latency_stream_stamp.json); ONE host thread sums 18.8 i64 rows per ns while the
// column is cache-resident (BENCH_r02 cpu_baseline.config0_1m_rows: 10^6 rows in 53 us) and adds two f64 columns at
// 1.9 rows per ns from DRAM, ~4 from cache (other_configs_single_thread). Crossovers: a sum 12 us x 18.8 = 2.2 x 10^5 rows
// (-> 2^18); an elementwise a (+) b 12 us x 3 = 3.6 x 10^4 rows (-> 2^15); a bitmap scan (popcount, ~150 bits per ns on the
// host) 10.5 us x 150 = 1.6 x 10^6 bits (-> 2^21). Many small columns: one ma_sum_columns call (0.24 us per column) or a
// replayed hipGraph instead of a call each. The library itself never computes on the CPU; this is advice the host reads.
int64_t ma_min_device_rows_for(int32_t kind) {
    static const int64_t reduce = (int64_t)env_bytes("MINARROW_HIP_MIN_ROWS", (size_t)1 << 18);
    static const int64_t elementwise = (int64_t)env_bytes("MINARROW_HIP_MIN_ROWS_ELEMENTWISE", (size_t)1 << 15);
    static const int64_t scan_bits = (int64_t)env_bytes("MINARROW_HIP_MIN_BITS_SCAN", (size_t)1 << 21);
    switch (kind) {
        case MA_KIND_ELEMENTWISE: return elementwise;
        case MA_KIND_BITMASK_SCAN: return scan_bits;
        default: return reduce;
    }
}

int64_t ma_min_device_rows(void) { return ma_min_device_rows_for(MA_KIND_REDUCTION); }

const char* ma_last_error_string(void) { return g_err; }

const char* ma_status_name(ma_status s) {
    switch (s) {
        case MA_OK: return "MA_OK";
        case MA_ERR_LENGTH_MISMATCH: return "MA_ERR_LENGTH_MISMATCH";
        case MA_ERR_DIVIDE_BY_ZERO: return "MA_ERR_DIVIDE_BY_ZERO";
        case MA_ERR_UNSUPPORTED: return "MA_ERR_UNSUPPORTED";
        case MA_ERR_INVALID_ARGUMENT: return "MA_ERR_INVALID_ARGUMENT";
        case MA_ERR_DEVICE: return "MA_ERR_DEVICE";
        case MA_ERR_NO_DEVICE: return "MA_ERR_NO_DEVICE";
        default: return "MA_ERR_UNKNOWN";
    }
}

// +1 / -1: the next context this thread creates gets its stream in the high / low priority class (ma::create_ctx_in_class)
static thread_local int t_stream_class = 0;

static ma_status ctx_create_impl(int32_t device, void* stream, bool borrow, ma_ctx** out_ctx, bool map_ordinal = true) {
    MA_REQUIRE(out_ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "out_ctx is NULL");
    *out_ctx = nullptr;
    int n = ma_device_count();
    if (n <= 0) {
        set_error("no HIP device is visible; libminarrow_hip has no CPU fallback");
        return MA_ERR_NO_DEVICE;
    }
    int ordinal = device;
    if (map_ordinal) {
        MA_REQUIRE(device >= 0 && device < n, MA_ERR_INVALID_ARGUMENT, "device ordinal %d out of range [0,%d)", device, n);
        device = device_map()[(size_t)device];  // MINARROW_HIP_DEVICES
    } else {
        ordinal = -1;  // a lane: inherits its root's ordinal (make_lane)
    }
    if (borrow && stream != nullptr) {
        // a borrowed stream belongs to whatever device it was made on: it must be the one the ordinal names
        hipDevice_t sdev = -1;
        if (hipStreamGetDevice((hipStream_t)stream, &sdev) == hipSuccess)
            MA_REQUIRE((int)sdev == device, MA_ERR_INVALID_ARGUMENT,
                       "the stream lives on HIP device %d but device ordinal %d is HIP device %d (MINARROW_HIP_DEVICES remaps "
                       "ordinals)", (int)sdev, ordinal, device);
        else
            (void)hipGetLastError();
    }
    MA_HIP(hipSetDevice(device));
    ma_ctx* c = new ma_ctx();
    c->device = device;
    c->ordinal = ordinal;
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) {
        delete c;
        return hip_fail(e, "hipGetDeviceProperties", __FILE__, __LINE__);
    }
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (borrow) {
        c->stream = (hipStream_t)stream;
        c->owns_stream = false;
    } else {
        // (tuning build only) MINARROW_HIP_STREAM_PRIORITY=high|low (read at every creation): the context's stream in that priority class — the
        // runtime keeps a hardware-queue pool per class, so a "high" context never shares a hardware queue with ordinary ones
        const char* prio = tuning_env("MINARROW_HIP_STREAM_PRIORITY");
        const int cls = t_stream_class ? t_stream_class : (prio && prio[0] == 'h') ? 1 : (prio && prio[0] == 'l') ? -1 : 0;
        int least = 0, greatest = 0;
        if (cls != 0 && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess)
            e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, cls > 0 ? greatest : least);
        else
            e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete c;
            return hip_fail(e, "hipStreamCreateWithFlags", __FILE__, __LINE__);
        }
        c->owns_stream = true;
    }
    auto fail = [&](hipError_t err, const char* what) {
        ma_status s = hip_fail(err, what, __FILE__, __LINE__);
        ma_ctx_destroy(c);
        return s;
    };
    if ((e = device_malloc(device, (void**)&c->partials, sizeof(Partial) * kMaxGrid)) != hipSuccess) return fail(e, "hipMalloc(partials)");
    // one zeroed block: [0] reduction ticket, [64 B] dev_flags, [128 B] bitmask scan accumulator, [256 B ...] 8 ticket shards
    if ((e = device_malloc(device, (void**)&c->ticket, 1024)) != hipSuccess) return fail(e, "hipMalloc(ticket)");
    if ((e = hipMemset(c->ticket, 0, 1024)) != hipSuccess) return fail(e, "hipMemset(ticket)");
    c->dev_flags = c->ticket + 16;  // same zeroed allocation, a different 64-B line
    if ((e = hipHostMalloc((void**)&c->result, sizeof(ResultSlot) * 4, hipHostMallocDefault)) != hipSuccess)
        return fail(e, "hipHostMalloc(result)");
    memset(c->result, 0, sizeof(ResultSlot) * 4);
    if ((e = hipEventCreate(&c->ev_start)) != hipSuccess) return fail(e, "hipEventCreate");
    if ((e = hipEventCreate(&c->ev_stop)) != hipSuccess) return fail(e, "hipEventCreate");
    // Environment surface (read at context creation): staging tile of host-resident operands and
    // how many lanes concurrent synchronous calls may fan out over.
    c->staging_tile_bytes = env_bytes("MINARROW_HIP_STAGING_TILE", c->staging_tile_bytes);
    if (c->staging_tile_bytes != 0 && c->staging_tile_bytes < ((size_t)1 << 16)) c->staging_tile_bytes = (size_t)1 << 16;
    const size_t lanes = env_bytes("MINARROW_HIP_LANES", (size_t)c->max_lanes);
    c->max_lanes = lanes < 1 ? 1 : (lanes > 16 ? 16 : (int)lanes);
    c->fenced_reduce = env_bytes("MINARROW_HIP_FENCED_REDUCE", 0) != 0;
    c->poll_us = (long)env_bytes("MINARROW_HIP_POLL_US", (size_t)c->poll_us);
    *out_ctx = c;
    return MA_OK;
}

}  // extern "C"

namespace ma {
// An independent context (own stream, partials, tickets) whose stream is in another priority class than ordinary contexts'.
// The runtime keeps a pool of hardware queues per class: two streams of one class may be mapped onto the same hardware queue —
// where a wait on one holds up the other — two streams of different classes never are.
ma_status create_ctx_in_class(int32_t device_ordinal, int cls, ma_ctx** out) {
    t_stream_class = cls;
    const ma_status st = ma_ctx_create(device_ordinal, out);
    t_stream_class = 0;
    return st;
}

ma_status make_lane(ma_ctx* root, ma_ctx** out, int cls) {
    int prev = 0;
    (void)hipGetDevice(&prev);
    t_stream_class = cls;
    ma_status st = ctx_create_impl(root->device, nullptr, false, out, false);  // root->device is already a HIP ordinal
    t_stream_class = 0;
    if (st == MA_OK) {
        (*out)->parent = root;
        (*out)->ordinal = root->ordinal;
        (*out)->max_lanes = 1;
    }
    (void)hipSetDevice(prev);
    return st;
}
}  // namespace ma

extern "C" {

ma_status ma_ctx_create(int32_t device_ordinal, ma_ctx** out_ctx) {
    return ctx_create_impl(device_ordinal, nullptr, false, out_ctx);
}

ma_status ma_ctx_create_on_stream(int32_t device_ordinal, void* hip_stream, ma_ctx** out_ctx) {
    return ctx_create_impl(device_ordinal, hip_stream, true, out_ctx);
}

void ma_ctx_destroy(ma_ctx* ctx) {
    if (!ctx) return;
    for (ma_ctx* lane : ctx->lanes) ma_ctx_destroy(lane);
    ctx->lanes.clear();
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->partials) (void)hipFree(ctx->partials);
    if (ctx->ticket) (void)hipFree(ctx->ticket);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->upload_stream) {
        (void)hipStreamSynchronize(ctx->upload_stream);
        (void)hipStreamDestroy(ctx->upload_stream);
    }
    for (int j = 0; j < ma_ctx::kDevTables; ++j) {
        if (ctx->dev_table[j]) (void)hipFree(ctx->dev_table[j]);
        if (ctx->dev_table_up[j]) (void)hipEventDestroy(ctx->dev_table_up[j]);
        if (ctx->dev_table_read[j]) (void)hipEventDestroy(ctx->dev_table_read[j]);
    }
    for (void* g : ctx->dev_table_garbage) (void)hipFree(g);
    ctx->dev_table_garbage.clear();
    for (void* g : ctx->table_garbage) (void)hipHostFree(g);
    ctx->table_garbage.clear();
    for (int k = 0; k < ma_ctx::kTableSlots; ++k) {
        if (ctx->table_stage[k]) (void)hipHostFree(ctx->table_stage[k]);
        if (ctx->table_ev[k]) (void)hipEventDestroy(ctx->table_ev[k]);
    }
    ma::pipe_destroy(ctx);
    if (ctx->result) (void)hipHostFree(ctx->result);
    if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
    if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
    for (hipEvent_t ev : ctx->marks)
        if (ev) (void)hipEventDestroy(ev);
    if (ctx->owns_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

ma_status ma_ctx_synchronize(ma_ctx* ctx) {
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    MA_ENTER_PRIMARY(ctx);
    MA_NO_CAPTURE(ctx, "ma_ctx_synchronize");
    MA_HIP(hipSetDevice(ctx->device));
    return sync_and_check(ctx);
}

ma_status ma_ctx_set_async(ma_ctx* ctx, int32_t enabled) {
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    MA_ENTER_PRIMARY(ctx);
    if (ctx->capturing) {  // takes effect when the capture ends
        ctx->async_before_capture = enabled != 0;
        return MA_OK;
    }
    ctx->async = enabled != 0;
    return MA_OK;
}

// ---- hipGraph capture -----------------------------------------------------------------------------
}  // extern "C"

struct ma_graph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    int device = 0;
    size_t nodes = 0;
    bool may_latch = false;  // a dense integer Div/Rem/FloorDiv was recorded: dev_flags must be inspected after a replay
};

extern "C" {

ma_status ma_ctx_capture_begin(ma_ctx* ctx) {
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    MA_ENTER_PRIMARY(ctx);
    MA_REQUIRE(!ctx->capturing, MA_ERR_INVALID_ARGUMENT, "a capture is already in progress on this context");
    MA_HIP(hipSetDevice(ctx->device));
    MA_HIP(hipStreamSynchronize(ctx->stream));
    // Relaxed: the library's own pointer classification (hipPointerGetAttributes) stays legal while recording.
    MA_HIP(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed));
    ctx->capturing = true;
    ctx->async_before_capture = ctx->async;
    ctx->async = true;
    ctx->pending_flags = false;
    return MA_OK;
}

ma_status ma_ctx_capture_end(ma_ctx* ctx, ma_graph** out_graph) {
    MA_REQUIRE(ctx != nullptr && out_graph != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx or out_graph is NULL");
    *out_graph = nullptr;
    MA_ENTER_PRIMARY(ctx);
    MA_REQUIRE(ctx->capturing, MA_ERR_INVALID_ARGUMENT, "no capture in progress on this context");
    MA_HIP(hipSetDevice(ctx->device));
    ctx->capturing = false;
    ctx->async = ctx->async_before_capture;
    const bool may_latch = ctx->pending_flags;
    ctx->pending_flags = false;  // nothing has executed yet
    hipGraph_t g = nullptr;
    MA_HIP(hipStreamEndCapture(ctx->stream, &g));
    MA_REQUIRE(g != nullptr, MA_ERR_DEVICE, "hipStreamEndCapture returned no graph (the capture was invalidated)");
    ma_graph* out = new ma_graph();
    out->graph = g;
    out->device = ctx->device;
    out->may_latch = may_latch;
    hipError_t e = hipGraphGetNodes(g, nullptr, &out->nodes);
    if (e == hipSuccess) e = hipGraphInstantiate(&out->exec, g, nullptr, nullptr, 0);
    if (e != hipSuccess) {
        (void)hipGraphDestroy(g);
        delete out;
        return hip_fail(e, "hipGraphInstantiate", __FILE__, __LINE__);
    }
    *out_graph = out;
    return MA_OK;
}

ma_status ma_graph_launch(ma_ctx* ctx, ma_graph* graph) {
    MA_REQUIRE(ctx != nullptr && graph != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx or graph is NULL");
    MA_REQUIRE(graph->device == ctx->device, MA_ERR_INVALID_ARGUMENT, "the graph was recorded on device %d, this context is on %d",
               graph->device, ctx->device);
    MA_ENTER_PRIMARY(ctx);
    MA_NO_CAPTURE(ctx, "ma_graph_launch");
    MA_HIP(hipSetDevice(ctx->device));
    MA_HIP(hipGraphLaunch(graph->exec, ctx->stream));
    if (graph->may_latch) ctx->pending_flags = true;
    return is_async(ctx) ? MA_OK : sync_and_check(ctx);
}

ma_status ma_graph_node_count(const ma_graph* graph, size_t* out_nodes) {
    MA_REQUIRE(graph != nullptr && out_nodes != nullptr, MA_ERR_INVALID_ARGUMENT, "graph or out_nodes is NULL");
    *out_nodes = graph->nodes;
    return MA_OK;
}

void ma_graph_destroy(ma_graph* graph) {
    if (!graph) return;
    (void)hipSetDevice(graph->device);
    if (graph->exec) (void)hipGraphExecDestroy(graph->exec);
    if (graph->graph) (void)hipGraphDestroy(graph->graph);
    delete graph;
}

void* ma_ctx_stream(ma_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }
int32_t ma_ctx_device(ma_ctx* ctx) { return ctx ? ctx->ordinal : -1; }
int32_t ma_ctx_hip_device(ma_ctx* ctx) { return ctx ? ctx->device : -1; }
int32_t ma_ctx_compute_units(ma_ctx* ctx) { return ctx ? ctx->num_cus : 0; }
int32_t ma_ctx_lane_count(ma_ctx* ctx) {
    if (!ctx) return 0;
    std::lock_guard<std::mutex> ll(ctx->lanes_mu);
    return 1 + (int32_t)ctx->lanes.size();
}

ma_status ma_ctx_set_blocks_per_cu(ma_ctx* ctx, int32_t blocks_per_cu) {
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    MA_REQUIRE(blocks_per_cu >= 0 && blocks_per_cu <= 64, MA_ERR_INVALID_ARGUMENT, "blocks_per_cu %d out of range",
               blocks_per_cu);
    MA_ENTER_PRIMARY(ctx);
    ctx->blocks_per_cu = blocks_per_cu;
    return MA_OK;
}

ma_status ma_ctx_set_grid(ma_ctx* ctx, int32_t workgroups) {
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    MA_REQUIRE(workgroups >= 0, MA_ERR_INVALID_ARGUMENT, "workgroups must be >= 0");
    MA_ENTER_PRIMARY(ctx);
    ctx->grid_override = workgroups;
    return MA_OK;
}

ma_status ma_ctx_set_variant(ma_ctx* ctx, int32_t variant) {
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    MA_REQUIRE(MA_TUNING || (variant & ~kFormBits) == 0, MA_ERR_UNSUPPORTED,
               "variant %d has tuning bits (%d): this build of the library keeps the form bits %d only; the forms those bits select live "
               "in a library built with TUNING=1 (make -C minarrow_amd/csrc TUNING=1)", variant, variant & ~kFormBits, kFormBits);
    MA_ENTER_PRIMARY(ctx);
    ctx->variant = variant;
    return MA_OK;
}

ma_status ma_ctx_timer_start(ma_ctx* ctx) {
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    MA_ENTER_PRIMARY(ctx);
    MA_NO_CAPTURE(ctx, "ma_ctx_timer_start");
    MA_HIP(hipSetDevice(ctx->device));
    MA_HIP(hipEventRecord(ctx->ev_start, ctx->stream));
    return MA_OK;
}

ma_status ma_ctx_timer_stop(ma_ctx* ctx) {
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    MA_ENTER_PRIMARY(ctx);
    MA_NO_CAPTURE(ctx, "ma_ctx_timer_stop");
    MA_HIP(hipSetDevice(ctx->device));
    MA_HIP(hipEventRecord(ctx->ev_stop, ctx->stream));
    return MA_OK;
}

ma_status ma_ctx_timer_elapsed_ms(ma_ctx* ctx, float* out_ms) {
    MA_REQUIRE(ctx != nullptr && out_ms != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx or out_ms is NULL");
    MA_ENTER_PRIMARY(ctx);
    MA_NO_CAPTURE(ctx, "ma_ctx_timer_elapsed_ms");
    MA_HIP(hipSetDevice(ctx->device));
    MA_HIP(hipEventSynchronize(ctx->ev_stop));
    MA_HIP(hipEventElapsedTime(out_ms, ctx->ev_start, ctx->ev_stop));
    return MA_OK;
}

ma_status ma_ctx_mark(ma_ctx* ctx, int32_t index) {
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    MA_REQUIRE(index >= 0 && index < MA_CTX_MAX_MARKS, MA_ERR_INVALID_ARGUMENT, "mark %d out of range [0,%d)", index, MA_CTX_MAX_MARKS);
    MA_ENTER_PRIMARY(ctx);
    MA_NO_CAPTURE(ctx, "ma_ctx_mark");
    MA_HIP(hipSetDevice(ctx->device));
    if (ctx->marks.size() <= (size_t)index) ctx->marks.resize((size_t)index + 1, nullptr);
    if (!ctx->marks[(size_t)index]) MA_HIP(hipEventCreate(&ctx->marks[(size_t)index]));
    MA_HIP(hipEventRecord(ctx->marks[(size_t)index], ctx->stream));
    return MA_OK;
}

ma_status ma_ctx_mark_elapsed_ms(ma_ctx* ctx, int32_t from_index, int32_t to_index, float* out_ms) {
    MA_REQUIRE(ctx != nullptr && out_ms != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx or out_ms is NULL");
    MA_ENTER_PRIMARY(ctx);
    MA_NO_CAPTURE(ctx, "ma_ctx_mark_elapsed_ms");
    for (int32_t i : {from_index, to_index})
        MA_REQUIRE(i >= 0 && (size_t)i < ctx->marks.size() && ctx->marks[(size_t)i] != nullptr, MA_ERR_INVALID_ARGUMENT,
                   "mark %d was never recorded on this context", i);
    MA_HIP(hipSetDevice(ctx->device));
    MA_HIP(hipEventSynchronize(ctx->marks[(size_t)to_index]));
    MA_HIP(hipEventElapsedTime(out_ms, ctx->marks[(size_t)from_index], ctx->marks[(size_t)to_index]));
    return MA_OK;
}

// ---- memory ---------------------------------------------------------------------------------

ma_status ma_alloc64_pinned(size_t bytes, void** out_ptr) {
    MA_REQUIRE(out_ptr != nullptr, MA_ERR_INVALID_ARGUMENT, "out_ptr is NULL");
    *out_ptr = nullptr;
    if (ma_device_count() <= 0) {
        set_error("no HIP device is visible; pinned allocation needs the HIP runtime");
        return MA_ERR_NO_DEVICE;
    }
    // hipHostMalloc returns page-aligned memory, which satisfies Vec64's 64-byte contract.
    void* p = nullptr;
    MA_REQUIRE(bytes < ((size_t)1 << 46), MA_ERR_INVALID_ARGUMENT, "pinned allocation of %zu bytes is too large", bytes);
    PinnedPool& pool = pinned_pool();
    const size_t rounded = pool_size_of(bytes < kPinnedPoolMinBytes ? kPinnedPoolMinBytes : bytes, kPinnedPoolMinBytes);
    {
        std::lock_guard<std::mutex> lock(pool.mu);
        auto it = pool.parked.find(rounded);
        if (it != pool.parked.end() && !it->second.empty()) {
            p = it->second.back();
            it->second.pop_back();
            pool.parked_set.erase(p);
            pool.cached_bytes -= rounded;
            pool.live.emplace(p, rounded);
            *out_ptr = p;
            return MA_OK;
        }
    }
    MA_HIP(hipHostMalloc(&p, rounded, hipHostMallocPortable | hipHostMallocMapped));
    std::lock_guard<std::mutex> lock(pool.mu);
    pool.live.emplace(p, rounded);
    *out_ptr = p;
    return MA_OK;
}

ma_status ma_free_pinned(void* ptr) {
    if (!ptr) return MA_OK;
    PinnedPool& pool = pinned_pool();
    {
        std::lock_guard<std::mutex> lock(pool.mu);
        MA_REQUIRE(pool.parked_set.count(ptr) == 0, MA_ERR_INVALID_ARGUMENT,
                   "ma_free_pinned(%p): the block is already free (double free)", ptr);
        auto it = pool.live.find(ptr);
        if (it != pool.live.end()) {
            const size_t rounded = it->second;
            pool.live.erase(it);
            if (pool.cached_bytes + rounded <= pool.limit_bytes) {
                pool.parked[rounded].push_back(ptr);
                pool.parked_set.insert(ptr);
                pool.cached_bytes += rounded;
                return MA_OK;
            }
            // the cache is full: give the block back to the runtime
        }
    }
    MA_HIP(hipHostFree(ptr));
    return MA_OK;
}

ma_status ma_pinned_pool_trim(size_t keep_bytes) {
    hipError_t first = hipSuccess;
    for (void* v : pool_take_victims(pinned_pool(), keep_bytes)) {
        hipError_t e = hipHostFree(v);
        if (e != hipSuccess && first == hipSuccess) first = e;
    }
    MA_HIP(first);
    return MA_OK;
}

ma_status ma_pinned_pool_set_limit(size_t limit_bytes) {
    PinnedPool& pool = pinned_pool();
    std::lock_guard<std::mutex> lock(pool.mu);
    pool.limit_bytes = limit_bytes;
    return MA_OK;
}

ma_status ma_host_register(void* ptr, size_t bytes) {
    MA_REQUIRE(ptr != nullptr && bytes > 0, MA_ERR_INVALID_ARGUMENT, "nothing to register");
    if (ma_device_count() <= 0) {
        set_error("no HIP device is visible; pinning host memory needs the HIP runtime");
        return MA_ERR_NO_DEVICE;
    }
    MA_HIP(hipHostRegister(ptr, bytes, hipHostRegisterPortable | hipHostRegisterMapped));
    return MA_OK;
}

ma_status ma_host_unregister(void* ptr) {
    if (!ptr) return MA_OK;
    MA_HIP(hipHostUnregister(ptr));
    return MA_OK;
}

}  // extern "C"

namespace ma {
// The store pattern of the elementwise kernels (a wave owns 8 KiB per step, six workgroups per CU, non-temporal 16-byte
// stores): the pattern whose rate tells a slow-writing region of HBM from a fast one (a tight 1-MiB
// front would write 6.3-6.5 TB/s on both and tell nothing).
typedef unsigned int ProbeVec __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(kBlock) void probe_write_kernel(ProbeVec* __restrict__ out, size_t n_tiles) {
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr size_t WAVE_VECS = 64 * 8, TILE_VECS = WAVE_VECS * kWaves;
    for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        ProbeVec* p = out + t * TILE_VECS + wave * WAVE_VECS + lane;
#pragma unroll
        for (int u = 0; u < 8; ++u) __builtin_nontemporal_store(ProbeVec{0u, 0u, 0u, 0u}, p + (size_t)u * 64);
    }
}

// The placement-INDEPENDENT write pattern: the whole grid advances as one 1-MiB front — one workgroup per
// CU, one 1-KiB store per wave per step. 6.3-6.5 TB/s on every block measured, fast or slow: the device's write ceiling,
// which is what the output allocator calibrates its "good block" threshold against (once per device).
__global__ __launch_bounds__(kBlock) void front_write_kernel(ProbeVec* __restrict__ out, size_t n_vecs) {
    const size_t stride = (size_t)gridDim.x * kBlock;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n_vecs; i += stride)
        __builtin_nontemporal_store(ProbeVec{0u, 0u, 0u, 0u}, out + i);
}

// Write rate of a device block in GB/s: three launches of probe_write_kernel, the faster of the last two counts (one
// 1.3-ms sample is at the mercy of whatever else the device was finishing). The block's contents are overwritten with
// zeros. The caller holds the context.
static ma_status measure_write_gbps(ma_ctx* ctx, void* block, size_t bytes, float* out_gbps, bool tight_front = false) {
    const size_t n_tiles = bytes / (16 * 64 * 8 * kWaves);
    *out_gbps = 0.f;
    if (n_tiles == 0) return MA_OK;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    hipError_t e = hipSuccess;
    for (int i = 0; i < 3 && e == hipSuccess; ++i) e = hipEventCreate(&ev[i]);
    const size_t cap = (size_t)ctx->num_cus * 6;
    const int grid = (int)(n_tiles < cap ? n_tiles : cap);
    const size_t n_vecs = n_tiles * 64 * 8 * kWaves;
    auto launch = [&] {
        if (tight_front)
            hipLaunchKernelGGL(front_write_kernel, dim3(ctx->num_cus), dim3(kBlock), 0, ctx->stream, (ProbeVec*)block, n_vecs);
        else
            hipLaunchKernelGGL(probe_write_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, (ProbeVec*)block, n_tiles);
    };
    float ms = 0.f;
    if (e == hipSuccess) {
        launch();
        for (int i = 0; i < 3; ++i) {
            (void)hipEventRecord(ev[i], ctx->stream);
            if (i < 2) launch();
        }
        e = hipEventSynchronize(ev[2]);
        float m1 = 0.f, m2 = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&m1, ev[0], ev[1]);
        if (e == hipSuccess) e = hipEventElapsedTime(&m2, ev[1], ev[2]);
        ms = m1 < m2 ? m1 : m2;
    }
    for (int i = 0; i < 3; ++i)
        if (ev[i]) (void)hipEventDestroy(ev[i]);
    if (e != hipSuccess) return hip_fail(e, "write-rate probe", __FILE__, __LINE__);
    if (ms > 0.f) *out_gbps = (float)((double)n_tiles * 16 * 64 * 8 * kWaves / (ms * 1e-3) / 1e9);
    return MA_OK;
}
}  // namespace ma

extern "C" {

// What the last ma_dev_alloc_output search on this thread cost (ma_dev_alloc_output_stats).
static thread_local double t_out_ms = 0.0;
static thread_local size_t t_out_held = 0;
static thread_local int32_t t_out_measured = 0, t_out_considered = 0;
static thread_local float t_out_good = 0.f;

// The placement search is OPT-IN since round 4: on the driver's box of round 3 it bought nothing (searched block 0.761 of peak
// vs plain block 0.763 for 7.7 ms and 8 GB held), on others +5-10 % — a lottery ticket, not an abstraction. Off: the plain
// allocator. MINARROW_HIP_OUTPUT_SEARCH=1 or ma_dev_output_search(1) turn it on for the process.
static std::atomic<int> g_output_search{-1};  // -1: not decided yet (the environment decides on first use)
static bool output_search_enabled() {
    int v = g_output_search.load();
    if (v < 0) {
        v = env_bytes("MINARROW_HIP_OUTPUT_SEARCH", 0) != 0 ? 1 : 0;
        g_output_search.store(v);
    }
    return v != 0;
}

int32_t ma_dev_output_search(int32_t enabled) {
    const int32_t before = output_search_enabled() ? 1 : 0;
    if (enabled >= 0) g_output_search.store(enabled ? 1 : 0);
    return before;
}

ma_status ma_dev_alloc_output(ma_ctx* ctx, size_t bytes, void** out_dev_ptr, float* out_write_gbps) {
    MA_REQUIRE(ctx != nullptr && out_dev_ptr != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx or out pointer is NULL");
    *out_dev_ptr = nullptr;
    if (out_write_gbps) *out_write_gbps = 0.f;
    t_out_ms = 0.0;
    t_out_held = 0;
    t_out_measured = t_out_considered = 0;
    MA_REQUIRE(bytes < ((size_t)1 << 46), MA_ERR_INVALID_ARGUMENT, "device allocation of %zu bytes is too large", bytes);
    MA_ENTER_PRIMARY(ctx);
    MA_NO_CAPTURE(ctx, "ma_dev_alloc_output");
    MA_HIP(hipSetDevice(ctx->device));
    // Small blocks, or the feature switched off: the plain allocator.
    // (the four tunables below are read in the tuning build only; the shipped library keeps their defaults)
    static const size_t kMinBytes = MA_TUNING ? env_bytes("MINARROW_HIP_OUTPUT_MIN_BYTES", (size_t)256 << 20) : (size_t)256 << 20;
    static const size_t kCandidates = MA_TUNING ? env_bytes("MINARROW_HIP_OUTPUT_CANDIDATES", 6) : 6;
    static const float kGoodEnv = MA_TUNING ? (float)env_bytes("MINARROW_HIP_OUTPUT_GOOD_GBPS", 0) : 0.0f;  // 0 = calibrate (below)
    static const size_t kHoldPercent = MA_TUNING ? env_bytes("MINARROW_HIP_OUTPUT_HOLD_PERCENT", 25) : 25;
    const int dev = ctx->device;
    if (!output_search_enabled() || bytes < kMinBytes || kCandidates <= 1 || dev < 0 || dev >= kMaxPooledDevices) {
        MA_HIP(dev_block_alloc(dev, bytes, out_dev_ptr));
        return MA_OK;
    }
    // The search holds its candidates while it runs — a block given back would be handed out again by the very next
    // allocation, and nothing new would be explored — so it is BOUNDED: the candidates alive at one time never exceed
    // MINARROW_HIP_OUTPUT_HOLD_PERCENT (25) of the HBM that is free when the call starts, the search stops when that many
    // are held, and an output too large for two candidates to fit the bound (a 64-GB consolidated column) takes the plain
    // path without measuring anything.
    size_t free_b = 0, total_b = 0;
    MA_HIP(hipMemGetInfo(&free_b, &total_b));
    const size_t rounded = pool_size_of(bytes);
    PinnedPool& pool = device_pool(dev);
    size_t parked_same = 0;  // parked blocks of this size class are candidates that take no new memory
    {
        std::lock_guard<std::mutex> plock(pool.mu);
        auto it = pool.parked.find(rounded);
        if (it != pool.parked.end()) parked_same = it->second.size();
    }
    const size_t budget = free_b / 100 * kHoldPercent;
    const size_t max_live = budget / rounded + parked_same;
    if (max_live < 2) {
        MA_HIP(dev_block_alloc(dev, bytes, out_dev_ptr));
        return MA_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    // Candidates come from the block cache first (parked blocks of this size class cost nothing to try), then from
    // the runtime; the search stops at the first block that writes at the fast regions' rate.
    std::vector<std::pair<void*, float>> tried;
    void* best = nullptr;
    float best_rate = -1.f;
    size_t measured = 0;
    // "Good" = 0.97 of the device's own write ceiling — the rate of the placement-independent tight-front pattern, measured
    // once per device on the first candidate (6.3-6.5 TB/s on the MI355X boxes seen, so ~6.2 TB/s; fast regions reach
    // 6.6-6.8 with the kernels' pattern, slow ones 5.4-5.8). MINARROW_HIP_OUTPUT_GOOD_GBPS overrides.
    static std::mutex cal_mu;
    static float cal_good[kMaxPooledDevices] = {};
    float good = kGoodEnv;
    // Parked blocks whose rate is already known cost nothing to consider: only fresh measurements count against the
    // limit. Most regions are slow ones (about three in four on the boxes measured): while nothing has reached the good
    // rate the search goes on once more, to twice the limit.
    while (measured < 2 * kCandidates && tried.size() < 2 * kCandidates + 8 && tried.size() < max_live) {
        if (measured >= kCandidates && best_rate >= 0.97f * good) break;
        void* blk = nullptr;
        hipError_t e = dev_block_alloc(dev, bytes, &blk);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            if (tried.empty()) return hip_fail(e, "hipMalloc", __FILE__, __LINE__);
            break;  // HBM is full: keep the best of what there is
        }
        if (good <= 0.f) {
            std::lock_guard<std::mutex> cl(cal_mu);
            if (cal_good[dev] <= 0.f) {
                float ceiling = 0.f;
                ma_status st = measure_write_gbps(ctx, blk, bytes, &ceiling, true);
                if (st != MA_OK) {
                    (void)dev_block_free(dev, blk);
                    return st;
                }
                cal_good[dev] = ceiling > 0.f ? 0.97f * ceiling : 6200.f;
            }
            good = cal_good[dev];
        }
        float rate = 0.f;
        bool known = false;
        {
            std::lock_guard<std::mutex> plock(pool.mu);
            auto it = pool.write_gbps.find(blk);
            if (it != pool.write_gbps.end()) {
                rate = it->second;
                known = true;
            }
        }
        if (!known) {
            ma_status st = measure_write_gbps(ctx, blk, bytes, &rate);
            if (st != MA_OK) {
                (void)dev_block_free(dev, blk);
                for (auto& t : tried) (void)dev_block_free(dev, t.first);
                return st;
            }
            std::lock_guard<std::mutex> plock(pool.mu);
            pool.write_gbps[blk] = rate;
            ++measured;
        }
        tried.emplace_back(blk, rate);
        if (rate > best_rate) {
            best_rate = rate;
            best = blk;
        }
        if (rate >= good) break;
    }
    // The others go (back) to the block cache: they read as fast as any block and serve as inputs.
    for (auto& t : tried)
        if (t.first != best) (void)dev_block_free(dev, t.first);
    t_out_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    t_out_held = tried.size() * rounded;
    t_out_measured = (int32_t)measured;
    t_out_considered = (int32_t)tried.size();
    t_out_good = good;
    *out_dev_ptr = best;
    if (out_write_gbps) *out_write_gbps = best_rate;
    return MA_OK;
}

ma_status ma_dev_alloc_output_stats(double* out_search_ms, size_t* out_held_peak_bytes, int32_t* out_blocks_measured,
                                    int32_t* out_blocks_considered, float* out_good_gbps) {
    if (out_search_ms) *out_search_ms = t_out_ms;
    if (out_held_peak_bytes) *out_held_peak_bytes = t_out_held;
    if (out_blocks_measured) *out_blocks_measured = t_out_measured;
    if (out_blocks_considered) *out_blocks_considered = t_out_considered;
    if (out_good_gbps) *out_good_gbps = t_out_good;
    return MA_OK;
}

// Live stamps and what each is made of: 1 = the runtime's signal memory, 0 = a plain device word; +2 = the runtime reports
// the word as HOST memory (signal memory is: profiles/r05_probe_signal.txt), i.e. the host may store to it directly.
static std::mutex g_stamp_mu;
static std::vector<std::pair<const uint64_t*, int>> g_stamps;

}  // extern "C"
namespace ma {
bool stamp_host_store(uint64_t* stamp, uint64_t value);
bool stamp_default_is_signal();
ma_status stamp_alloc_kind(ma_ctx* ctx, uint64_t** out_stamp, bool want_signal);
}  // namespace ma
extern "C" {

ma_status ma_stamp_alloc(ma_ctx* ctx, uint64_t** out_stamp) {
    return ma::stamp_alloc_kind(ctx, out_stamp, ma::stamp_default_is_signal());
}
}  // extern "C"

// Which memory a hand-off stamp lives in. A stream waits on either kind (hipStreamWaitValue64) and a kernel's system-scope
// store reaches either, but the wait is not the same thing: on the runtime's SIGNAL memory (one 8-byte value — any other size
// is refused — and host memory: profiles/r05_probe_signal.txt) it is a barrier-value packet the command processor polls, on
// a plain device word a one-wave kernel that spins. Measured on the overlapped step of the 8-way share (125 M rows per
// column, one fused scan per step, profiles/r05_share_1gpu.txt): with the stamp in signal memory the SCAN kernel between
// its timing marks takes 0.342-0.360 ms instead of 0.288 — the packet the exchange stream's queue sits in holds up the scan
// stream's dispatches — so hand-off stamps are device words (rounds 3-4's de-facto form: their 64-byte signal request was
// always refused), MINARROW_HIP_STAMP_SIGNAL=1 asks for signal memory (A/B). Signal memory is what the group's stall words
// use (testing hooks): a host store releases them with no GPU queue involved.
bool ma::stamp_default_is_signal() {
    static const bool on = [] {
        const char* e = tuning_env("MINARROW_HIP_STAMP_SIGNAL");
        return e && e[0] && e[0] != '0';
    }();
    return on;
}

ma_status ma::stamp_alloc_kind(ma_ctx* ctx, uint64_t** out_stamp, bool want_signal) {
    MA_REQUIRE(ctx != nullptr && out_stamp != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx or out_stamp is NULL");
    *out_stamp = nullptr;
    MA_ENTER_PRIMARY(ctx);
    MA_HIP(hipSetDevice(ctx->device));
    void* p = nullptr;
    int kind = 1;
    if (!want_signal || hipExtMallocWithFlags(&p, 8, hipMallocSignalMemory) != hipSuccess || hipMemset(p, 0, 8) != hipSuccess) {
        (void)hipGetLastError();
        if (p) (void)hipFree(p);
        p = nullptr;
        kind = 0;
        MA_HIP(hipMalloc(&p, 64));
        MA_HIP(hipMemset(p, 0, 64));
    }
    if (kind == 1) {
        hipPointerAttribute_t attr{};
        if (hipPointerGetAttributes(&attr, p) == hipSuccess && attr.type == hipMemoryTypeHost && attr.hostPointer == p) kind |= 2;
        (void)hipGetLastError();
    }
    {
        std::lock_guard<std::mutex> lock(g_stamp_mu);
        g_stamps.push_back({(const uint64_t*)p, kind});
    }
    *out_stamp = (uint64_t*)p;
    return MA_OK;
}

// `*stamp = value` by a plain host store when the word is host memory: the one release that needs no queue of the GPU (a
// write packet can end up behind the very wait it is meant to end when streams share a hardware queue).
bool ma::stamp_host_store(uint64_t* stamp, uint64_t value) {
    {
        std::lock_guard<std::mutex> lock(g_stamp_mu);
        bool host = false;
        for (const auto& e : g_stamps)
            if (e.first == stamp) host = (e.second & 2) != 0;
        if (!host) return false;
    }
    __atomic_store_n(stamp, value, __ATOMIC_RELEASE);
    return true;
}
extern "C" {

ma_status ma_stamp_free(ma_ctx* ctx, uint64_t* stamp) {
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    if (!stamp) return MA_OK;
    MA_ENTER_PRIMARY(ctx);
    MA_HIP(hipSetDevice(ctx->device));
    {
        std::lock_guard<std::mutex> lock(g_stamp_mu);
        for (size_t i = 0; i < g_stamps.size(); ++i)
            if (g_stamps[i].first == stamp) {
                g_stamps.erase(g_stamps.begin() + (long)i);
                break;
            }
    }
    MA_HIP(hipFree(stamp));
    return MA_OK;
}

ma_status ma_ctx_wait_value(ma_ctx* ctx, const uint64_t* word, uint64_t value) {
    MA_REQUIRE(ctx != nullptr && word != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx or word is NULL");
    MA_REQUIRE(((uintptr_t)word & 7) == 0 && pointer_kind(word) != kPageable, MA_ERR_INVALID_ARGUMENT,
               "the word must be 8-byte aligned, device-reachable memory (ma_stamp_alloc)");
    MA_ENTER_PRIMARY(ctx);
    MA_NO_CAPTURE(ctx, "ma_ctx_wait_value");
    MA_HIP(hipSetDevice(ctx->device));
    const hipError_t e = hipStreamWaitValue64(ctx->stream, (void*)word, value, hipStreamWaitValueGte, ~(uint64_t)0);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("this runtime has no stream memory operations (hipStreamWaitValue64: %s): order the streams with events", hipGetErrorString(e));
        return MA_ERR_UNSUPPORTED;
    }
    return MA_OK;
}

int32_t ma_stamp_is_signal(const uint64_t* stamp) {
    std::lock_guard<std::mutex> lock(g_stamp_mu);
    for (const auto& e : g_stamps)
        if (e.first == stamp) return e.second & 1;
    return -1;
}

ma_status ma_dev_alloc(ma_ctx* ctx, size_t bytes, void** out_dev_ptr) {
    MA_REQUIRE(ctx != nullptr && out_dev_ptr != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx or out pointer is NULL");
    *out_dev_ptr = nullptr;
    MA_REQUIRE(bytes < ((size_t)1 << 46), MA_ERR_INVALID_ARGUMENT, "device allocation of %zu bytes is too large", bytes);
    MA_ENTER_PRIMARY(ctx);
    MA_NO_CAPTURE(ctx, "ma_dev_alloc");
    MA_HIP(hipSetDevice(ctx->device));
    MA_HIP(dev_block_alloc(ctx->device, bytes, out_dev_ptr));
    return MA_OK;
}

ma_status ma_dev_free(ma_ctx* ctx, void* dev_ptr) {
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    if (!dev_ptr) return MA_OK;
    MA_ENTER_PRIMARY(ctx);
    MA_NO_CAPTURE(ctx, "ma_dev_free");
    MA_HIP(hipSetDevice(ctx->device));
    MA_HIP(hipStreamSynchronize(ctx->stream));  // nothing enqueued through this context still touches the block
    hipError_t fe = dev_block_free(ctx->device, dev_ptr);
    MA_REQUIRE(fe != hipErrorInvalidValue, MA_ERR_INVALID_ARGUMENT, "ma_dev_free(%p): the block is already free (double free)", dev_ptr);
    MA_HIP(fe);
    return MA_OK;
}

ma_status ma_dev_pool_trim(ma_ctx* ctx, size_t keep_bytes) {
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    MA_REQUIRE(ctx->device >= 0 && ctx->device < kMaxPooledDevices, MA_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    MA_ENTER_PRIMARY(ctx);
    MA_HIP(hipSetDevice(ctx->device));
    hipError_t first = hipSuccess;
    for (void* v : pool_take_victims(device_pool(ctx->device), keep_bytes)) {
        hipError_t e = hipFree(v);
        if (e != hipSuccess && first == hipSuccess) first = e;
    }
    MA_HIP(first);
    return MA_OK;
}

ma_status ma_dev_pool_set_limit(ma_ctx* ctx, size_t limit_bytes) {
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    MA_REQUIRE(ctx->device >= 0 && ctx->device < kMaxPooledDevices, MA_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    PinnedPool& pool = device_pool(ctx->device);
    std::lock_guard<std::mutex> lock(pool.mu);
    pool.limit_bytes = limit_bytes;
    return MA_OK;
}

ma_status ma_dev_upload(ma_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes) {
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    if (bytes == 0) return MA_OK;
    MA_REQUIRE(dst_dev && src_host, MA_ERR_INVALID_ARGUMENT, "NULL buffer");
    MA_ENTER_PRIMARY(ctx);
    MA_NO_CAPTURE(ctx, "ma_dev_upload");
    MA_HIP(hipSetDevice(ctx->device));
    MA_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    MA_TRY(stream_wait(ctx));
    return MA_OK;
}

ma_status ma_dev_download(ma_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes) {
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    if (bytes == 0) return MA_OK;
    MA_REQUIRE(dst_host && src_dev, MA_ERR_INVALID_ARGUMENT, "NULL buffer");
    MA_ENTER_PRIMARY(ctx);
    MA_NO_CAPTURE(ctx, "ma_dev_download");
    MA_HIP(hipSetDevice(ctx->device));
    MA_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    MA_TRY(stream_wait(ctx));
    return MA_OK;
}

ma_status ma_dev_memset(ma_ctx* ctx, void* dst_dev, int32_t byte_value, size_t bytes) {
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    if (bytes == 0) return MA_OK;
    MA_REQUIRE(dst_dev != nullptr, MA_ERR_INVALID_ARGUMENT, "NULL buffer");
    MA_ENTER_PRIMARY(ctx);
    MA_HIP(hipSetDevice(ctx->device));
    MA_HIP(hipMemsetAsync(dst_dev, byte_value, bytes, ctx->stream));
    if (!is_async(ctx)) MA_TRY(stream_wait(ctx));
    return MA_OK;
}

ma_status ma_dev_copy(ma_ctx* ctx, void* dst_dev, const void* src_dev, size_t bytes) {
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    if (bytes == 0) return MA_OK;
    MA_REQUIRE(dst_dev && src_dev, MA_ERR_INVALID_ARGUMENT, "NULL buffer");
    MA_ENTER_PRIMARY(ctx);
    MA_HIP(hipSetDevice(ctx->device));
    MA_HIP(hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    if (!is_async(ctx)) MA_TRY(stream_wait(ctx));
    return MA_OK;
}

int32_t ma_pointer_kind(const void* ptr) {
    if (ma_device_count() <= 0) return kPageable;
    return (int32_t)pointer_kind(ptr);
}

int32_t ma_pointer_device(const void* ptr) {
    if (ptr == nullptr || ma_device_count() <= 0) return -1;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, ptr) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return attr.type == hipMemoryTypeDevice ? attr.device : -1;
}

// ---- synthetic inputs -------------------------------------------------------------------------

ma_status ma_synth_iota_i64(ma_ctx* ctx, int64_t* dst, size_t n, int64_t start) {
    return launch_fill(ctx, dst, n, IotaI64{start});
}
ma_status ma_synth_iota_f64(ma_ctx* ctx, double* dst, size_t n, int64_t start) {
    return launch_fill(ctx, dst, n, IotaF64{start});
}
ma_status ma_synth_iota_i32(ma_ctx* ctx, int32_t* dst, size_t n, int32_t start) {
    return launch_fill(ctx, dst, n, IotaI32{start});
}
ma_status ma_synth_iota_f32(ma_ctx* ctx, float* dst, size_t n, int32_t start) {
    return launch_fill(ctx, dst, n, IotaF32{start});
}
ma_status ma_synth_splitmix_i64(ma_ctx* ctx, int64_t* dst, size_t n, uint64_t seed, uint64_t first_index) {
    return launch_fill(ctx, dst, n, SplitI64{seed, first_index});
}
ma_status ma_synth_splitmix_f64(ma_ctx* ctx, double* dst, size_t n, uint64_t seed, uint64_t first_index) {
    return launch_fill(ctx, dst, n, SplitF64{seed, first_index});
}

ma_status ma_synth_validity(ma_ctx* ctx, uint8_t* dst_bits, size_t n_bits, uint64_t seed, uint64_t first_index,
                            uint32_t null_every) {
    MA_REQUIRE(ctx != nullptr, MA_ERR_INVALID_ARGUMENT, "ctx is NULL");
    if (n_bits == 0) return MA_OK;
    MA_REQUIRE(dst_bits != nullptr, MA_ERR_INVALID_ARGUMENT, "dst_bits is NULL");
    MA_REQUIRE(null_every >= 1, MA_ERR_INVALID_ARGUMENT, "null_every must be >= 1");
    MA_REQUIRE(((uintptr_t)dst_bits & 7) == 0, MA_ERR_INVALID_ARGUMENT, "validity bitmap must be 8-byte aligned");
    MA_REQUIRE(pointer_kind(dst_bits) != kPageable, MA_ERR_INVALID_ARGUMENT,
               "synthetic generators need a device-reachable destination");
    MA_ENTER(ctx);
    MA_HIP(hipSetDevice(ctx->device));
    size_t n_words = (n_bits + 63) / 64;
    int grid = grid_for(ctx, (n_words + kWaves - 1) / kWaves);
    hipLaunchKernelGGL(validity_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, (uint64_t*)dst_bits, n_bits, seed,
                       first_index, null_every);
    MA_HIP(hipGetLastError());
    if (!is_async(ctx)) MA_TRY(stream_wait(ctx));
    return MA_OK;
}

}  // extern "C"
