// minarrow_hip_parallel.hpp — the reference's `parallel_proc` path as a typed C++17 mirror over the C ABI's group API.
//
// The only place the reference parallelises the hot path is its bench harness:
//     fn rayon_simd_sum_i64(slice: &[i64]) -> i64 { slice.par_chunks(1 << 20).map(simd_sum_i64::<4>).sum() }
// (benches/benchmark_parallel_simd.rs:81-98: Rayon workers on all cores, per-chunk partials, one `.sum()`). With the
// column resident in HBM the chunks are one per GPU — 64-row-aligned, so that a validity word never straddles two devices
// — every member of a `Group` scans its chunk on its own device (issued by its own thread, like a Rayon worker), and the
// `.sum()` of the partials is ONE exchange of 64-byte records (RCCL all-gather + rank-ordered fold inside the library, or the
// host fold of pinned records): wrapping adds for integers, error-free double-double folding for floats, so the f64 result
// stays within 1 ULP of the exactly rounded sum whatever the number of GPUs.
//
//     ma::Group g({0, 1, 2, 3, 4, 5, 6, 7});
//     auto col = g.scatter(host_ptr, n);              // ShardedColumn<int64_t>: chunk i in member i's HBM
//     int64_t s = g.rayon_simd_sum_i64(col);           // == the reference's rayon_simd_sum_i64(&host[..n])
//
// Header-only; needs minarrow_hip.hpp (KernelError, check). Nothing here computes on the CPU.
#pragma once

#include <cstring>
#include <memory>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "minarrow_hip.hpp"

namespace ma {

// Row ranges [lo, hi) of an n-row column for `parts` GPUs; interior boundaries are multiples of 64 rows (the 6-line rule
// of minarrow_amd/parallel.py::row_chunks: what `par_chunks` becomes when a chunk must start on a validity word).
inline std::vector<std::pair<size_t, size_t>> row_chunks(size_t n, size_t parts) {
    std::vector<std::pair<size_t, size_t>> out;
    const size_t units = (n + 63) / 64;
    size_t lo = 0;
    for (size_t r = 0; r < parts; ++r) {
        size_t hi = r + 1 == parts ? n : std::min(n, (units * (r + 1) / parts) * 64);
        if (hi < lo) hi = lo;
        out.emplace_back(lo, hi);
        lo = hi;
    }
    return out;
}

// A column whose row chunks live on the members of a group: chunk i is device memory of member i (freed with the column).
template <typename T>
struct ShardedColumn {
    std::vector<std::shared_ptr<void>> owners;  // data (and validity) allocations, one or two per member
    std::vector<const T*> chunks;
    std::vector<size_t> lens;
    std::vector<const uint8_t*> masks;          // empty: no validity. Else one bitmap per member, bit 0 = the chunk's row 0
    std::vector<size_t> mask_offsets;
    size_t rows = 0;
    bool has_mask() const { return !masks.empty(); }
};

class Group {
  public:
    // flags: MA_GROUP_EXCHANGE_* / MA_GROUP_ISSUE_CALLER (minarrow_hip.h). The default asks for the RCCL exchange and
    // accepts the host fold where RCCL cannot be set up (ma_group_exchange_note says why).
    explicit Group(const std::vector<int32_t>& devices,
                   uint32_t flags = MA_GROUP_EXCHANGE_RCCL | MA_GROUP_EXCHANGE_FALLBACK_HOST) {
        check(ma_group_create_ex(devices.data(), (int32_t)devices.size(), flags, &g_));
    }
    ~Group() { ma_group_destroy(g_); }
    Group(const Group&) = delete;
    Group& operator=(const Group&) = delete;
    ma_group* get() const { return g_; }
    size_t size() const { return (size_t)ma_group_size(g_); }
    ma_ctx* ctx(size_t member) const { return ma_group_ctx(g_, (int32_t)member); }
    bool rccl() const { return ma_group_exchange_kind(g_) == 1; }
    bool issue_threads() const { return ma_group_issue_kind(g_) == 1; }
    bool peer_access(size_t from, size_t to) const { return ma_group_peer_access(g_, (int32_t)from, (int32_t)to) == 1; }
    std::string note() const { return ma_group_exchange_note(g_); }

    // Uploads chunk i of the host column (and of its validity, if any: `mask_bits` is the Arrow bitmap of the WHOLE column,
    // bit i = row i) into member i's HBM. Chunk boundaries are multiples of 64 rows = 8 bytes of bitmap, so every member's
    // bitmap copy starts on a byte and its bit offset is 0.
    template <typename T>
    ShardedColumn<T> scatter(const T* host, size_t n, const uint8_t* mask_bits = nullptr) const {
        ShardedColumn<T> col;
        col.rows = n;
        const auto chunks = row_chunks(n, size());
        for (size_t i = 0; i < chunks.size(); ++i) {
            const size_t lo = chunks[i].first, len = chunks[i].second - chunks[i].first;
            ma_ctx* c = ctx(i);
            void* p = nullptr;
            check(ma_dev_alloc(c, len * sizeof(T) + 64, &p));
            col.owners.emplace_back(p, [c](void* q) { (void)ma_dev_free(c, q); });
            if (len) check(ma_dev_upload(c, p, host + lo, len * sizeof(T)));
            col.chunks.push_back(static_cast<const T*>(p));
            col.lens.push_back(len);
            if (mask_bits) {
                const size_t nbytes = (len + 7) / 8;
                void* m = nullptr;
                check(ma_dev_alloc(c, ((nbytes + 7) / 8) * 8 + 64, &m));  // whole u64 words readable, like the reference
                col.owners.emplace_back(m, [c](void* q) { (void)ma_dev_free(c, q); });
                check(ma_dev_memset(c, m, 0, ((nbytes + 7) / 8) * 8 + 64));
                if (nbytes) check(ma_dev_upload(c, m, mask_bits + lo / 8, nbytes));
                col.masks.push_back(static_cast<const uint8_t*>(m));
                col.mask_offsets.push_back(0);
            }
        }
        return col;
    }

    // rayon_simd_sum_i64 — benches/benchmark_parallel_simd.rs:81-88. Wrapping sum of the valid rows (all rows without a
    // bitmap); *valid_count receives their number. A chunk on the wrong member's device is KernelError::InvalidArguments.
    int64_t rayon_simd_sum_i64(const ShardedColumn<int64_t>& col, uint64_t* valid_count = nullptr) const {
        require_shape(col.chunks.size());
        int64_t s = 0;
        uint64_t c = 0;
        check(ma_group_sum_i64(g_, col.chunks.data(), col.lens.data(), col.has_mask() ? col.masks.data() : nullptr,
                               col.has_mask() ? col.mask_offsets.data() : nullptr, &s, &c));
        if (valid_count) *valid_count = c;
        return s;
    }
    // rayon_simd_sum_f64 — :91-98. Within 1 ULP of the exactly rounded sum, bit-reproducible for a given group size.
    double rayon_simd_sum_f64(const ShardedColumn<double>& col, uint64_t* valid_count = nullptr) const {
        require_shape(col.chunks.size());
        double s = 0;
        uint64_t c = 0;
        check(ma_group_sum_f64(g_, col.chunks.data(), col.lens.data(), col.has_mask() ? col.masks.data() : nullptr,
                               col.has_mask() ? col.mask_offsets.data() : nullptr, &s, &c));
        if (valid_count) *valid_count = c;
        return s;
    }
    // Both columns of the reference's bench in ONE step — two scans per GPU, one exchange — enqueue-only: call wait().
    // `column` (0 .. MA_GROUP_MAX_COLUMNS-1) names the record set the results are read from.
    void enqueue_sums(int32_t column, const ShardedColumn<int64_t>& ints, const ShardedColumn<double>& floats) const {
        require_shape(ints.chunks.size());
        require_shape(floats.chunks.size());
        check(ma_group_enqueue_sum_i64(g_, column, ints.chunks.data(), ints.lens.data(), ints.has_mask() ? ints.masks.data() : nullptr,
                                       ints.has_mask() ? ints.mask_offsets.data() : nullptr));
        check(ma_group_enqueue_sum_f64(g_, column, floats.chunks.data(), floats.lens.data(),
                                       floats.has_mask() ? floats.masks.data() : nullptr,
                                       floats.has_mask() ? floats.mask_offsets.data() : nullptr));
        check(ma_group_exchange(g_));
    }
    // ONE column held as many chunks spread over the group — a SuperArray, or one column of a SuperTable's batches (config 5
    // at the reference's batch sizes): chunk i lives on member i % size(). Enqueue-only (ma_group_enqueue_sum_chunks + the
    // exchange): wait(), then sums(column). T: int64_t ('l'), int32_t ('i'), double ('g'), float ('f').
    template <typename T>
    void enqueue_sum_chunks(int32_t column, const std::vector<const T*>& chunks, const std::vector<size_t>& lens,
                            const std::vector<const uint8_t*>* masks = nullptr,
                            const std::vector<size_t>* mask_offsets = nullptr) const {
        static_assert(std::is_same<T, int64_t>::value || std::is_same<T, int32_t>::value || std::is_same<T, double>::value ||
                          std::is_same<T, float>::value,
                      "i64, i32, f64 or f32 chunks");
        if (chunks.size() != lens.size() || (masks && masks->size() != chunks.size()))
            throw KernelError(KernelError::InvalidArguments, "chunk tables of different lengths");
        const int32_t fmt = std::is_same<T, int64_t>::value ? 'l' : std::is_same<T, int32_t>::value ? 'i' : std::is_same<T, double>::value ? 'g' : 'f';
        check(ma_group_enqueue_sum_chunks(g_, column, fmt, chunks.size(), reinterpret_cast<const void* const*>(chunks.data()),
                                          lens.data(), masks ? masks->data() : nullptr,
                                          mask_offsets ? mask_offsets->data() : nullptr));
        check(ma_group_exchange(g_));
    }
    void wait() const { check(ma_group_synchronize(g_)); }
    // wait() with a deadline (ma_group_synchronize_for): past it the group's communicators are aborted and KernelError names
    // the members and phases still pending — a Rayon worker's panic, instead of a host blocked for good. broken() then says
    // whether rebuild() can give the same members a fresh exchange (1) or a stream never ran empty (2).
    void wait_for(double timeout_ms) const { check(ma_group_synchronize_for(g_, timeout_ms)); }
    int broken() const { return ma_group_is_broken(g_); }
    void rebuild(uint32_t flags) const { check(ma_group_rebuild_exchange(g_, flags)); }
    uint32_t flags() const { return ma_group_flags(g_); }
    // MA_GROUP_SCAN_LANES (with RCCL | OVERLAP): consecutive enqueue_table steps alternate between two scan streams per GPU, each
    // gated on the early stamp of the one before. set_scan_lanes switches them on / off without a rebuild (a host measures both);
    // join_lanes puts every member's own stream behind its second lane (before the host enqueues work of its own there).
    void set_scan_lanes(bool on) const { check(ma_group_set_scan_lanes(g_, on ? 1 : 0)); }
    void join_lanes() const { check(ma_group_join_lanes(g_)); }
    // Proves the exchange (rank-tagged records, in the group's own form), the stamp hand-off and every peer link before a job
    // is trusted to them, each step under the deadline (ma_group_selftest). Returns the report; throws when something failed.
    ma_selftest_report selftest(double timeout_ms, uint32_t what = 0) const {
        ma_selftest_report rep;
        check(ma_group_selftest(g_, what, timeout_ms, &rep));
        return rep;
    }
    std::pair<int64_t, double> sums(int32_t column) const {
        int64_t i = 0;
        double f = 0;
        check(ma_group_result(g_, column, &i, nullptr, &f, nullptr));
        return {i, f};
    }

  private:
    void require_shape(size_t n_chunks) const {
        if (n_chunks != size())
            throw KernelError(KernelError::InvalidArguments, "a sharded column needs one chunk per group member");
    }
    ma_group* g_ = nullptr;
};

// Back-to-back sums on ONE GPU as a pipeline (ma_scan_lanes_*): the reference's hot loop of sums
// (benches/hotloop_benchmark_avg_std.rs:48-62: ITERATIONS passes, an i64 and an f64 sum each; the pass itself: hotloop_benchmark_std.rs:109-127) with consecutive fused scans on two streams of the context's device, each started
// when the one before it has begun to drain. enqueue() only enqueues, whatever mode the context is in; every scan in flight
// writes its OWN records; join() before anything else on the context that reads them or overwrites the columns.
class ScanLanes {
  public:
    explicit ScanLanes(ma_ctx* ctx) { check(ma_scan_lanes_create(ctx, &l_)); }
    ~ScanLanes() { ma_scan_lanes_destroy(l_); }
    ScanLanes(const ScanLanes&) = delete;
    ScanLanes& operator=(const ScanLanes&) = delete;
    void enqueue(const std::vector<ma_fused_column>& cols) const { check(ma_scan_lanes_sum_fused(l_, cols.size(), cols.data())); }
    // one i64 and one f64 column into the integer and the float slots of one 64-byte record ({sum, count, hi, lo, count})
    void enqueue_pair(const int64_t* ints, const double* floats, size_t n, uint64_t* record) const {
        ma_fused_column c[2] = {};
        c[0].data = ints, c[0].n = n, c[0].null_count = -1, c[0].format_code = 'l', c[0].out = record;
        c[1].data = floats, c[1].n = n, c[1].null_count = -1, c[1].format_code = 'g', c[1].out = record + 2;
        check(ma_scan_lanes_sum_fused(l_, 2, c));
    }
    // ONE column of any numeric type (Arrow format character) through the single-column kernels: out_sum = the wrapping 64-bit
    // sum or the f64 sum (with out_lo the double-double pair); every output device-reachable
    void enqueue_sum(char format, const void* data, size_t n, void* out_sum, uint64_t* out_count = nullptr, double* out_lo = nullptr,
                     const uint8_t* mask_bits = nullptr, size_t mask_bit_offset = 0, int64_t null_count = -1) const {
        check(ma_scan_lanes_sum(l_, format, data, n, mask_bits, mask_bit_offset, null_count, out_sum, out_lo, out_count));
    }
    void join() const { check(ma_scan_lanes_join(l_)); }
    void synchronize() const { check(ma_scan_lanes_synchronize(l_)); }
    // the form for a host's loop: past the deadline the gates are released, the pipeline is broken and the error names the lane
    void synchronize_for(double timeout_ms) const { check(ma_scan_lanes_synchronize_for(l_, timeout_ms)); }
    int is_broken() const { return ma_scan_lanes_is_broken(l_); }
    uint64_t scans() const { return ma_scan_lanes_scans(l_); }

  private:
    ma_scan_lanes* l_ = nullptr;
};

}  // namespace ma
