// Single-process, multi-GPU row-chunk reduction: what the reference's Rayon path
// (`slice.par_chunks(1 << 20).map(simd_sum).sum()`, benches/benchmark_parallel_simd.rs:81-98) becomes for a host
// that drives all GPUs of a node from one process (e.g. the Rust library itself). One ma_ctx per device; every
// device scans its own row chunk concurrently (async launches on independent streams) and writes {sum | hi, lo,
// count} into a pinned record; the host folds the G records in device order (wrapping add; double-double for floats,
// so the f64 total stays within 1 ULP). G x 24 bytes: no collective is needed inside one process. The multi-PROCESS
// form of the same exchange is one RCCL all-gather (minarrow_amd/parallel.py, bench.py).
#include <vector>

#include "ma_common.hpp"

struct ma_group {
    std::vector<ma_ctx*> ctxs;
    uint64_t* records = nullptr;  // pinned, 4 x u64 per member
};

using namespace ma;

namespace {

inline void two_sum_acc(double& hi, double& lo, double h, double l) {
    double t = hi + h;
    double bp = t - hi;
    double e = (hi - (t - bp)) + (h - bp);
    hi = t;
    lo += e + l;
}

template <typename Launch>
ma_status group_run(ma_group* g, const size_t* lens, Launch launch) {
    const size_t n = g->ctxs.size();
    std::vector<int> was_async(n);
    ma_status st = MA_OK;
    for (size_t i = 0; i < n; ++i) MA_NO_CAPTURE(g->ctxs[i], "a group reduction");
    for (size_t i = 0; i < n; ++i) {
        {
            std::lock_guard<std::mutex> lock(g->ctxs[i]->mu);
            was_async[i] = g->ctxs[i]->async ? 1 : 0;
        }
        (void)ma_ctx_set_async(g->ctxs[i], 1);
    }
    for (size_t i = 0; i < n && st == MA_OK; ++i) {
        uint64_t* rec = g->records + 4 * i;
        rec[0] = rec[1] = rec[2] = 0;
        if (lens[i]) st = launch(i, rec);  // enqueue only: all devices run concurrently
    }
    for (size_t i = 0; i < n; ++i) {
        ma_status s = ma_ctx_synchronize(g->ctxs[i]);
        if (st == MA_OK) st = s;
        (void)ma_ctx_set_async(g->ctxs[i], was_async[i]);
    }
    return st;
}

}  // namespace

extern "C" {

ma_status ma_group_create(const int32_t* device_ordinals, int32_t n_members, ma_group** out_group) {
    MA_REQUIRE(out_group != nullptr, MA_ERR_INVALID_ARGUMENT, "out_group is NULL");
    *out_group = nullptr;
    MA_REQUIRE(device_ordinals != nullptr && n_members > 0 && n_members <= 1024, MA_ERR_INVALID_ARGUMENT,
               "a group needs 1..1024 members");
    ma_group* g = new ma_group();
    for (int32_t i = 0; i < n_members; ++i) {
        ma_ctx* c = nullptr;
        ma_status st = ma_ctx_create(device_ordinals[i], &c);
        if (st != MA_OK) {
            for (ma_ctx* x : g->ctxs) ma_ctx_destroy(x);
            delete g;
            return st;
        }
        g->ctxs.push_back(c);
    }
    hipError_t e = hipHostMalloc((void**)&g->records, sizeof(uint64_t) * 4 * (size_t)n_members,
                                 hipHostMallocPortable | hipHostMallocMapped);
    if (e != hipSuccess) {
        for (ma_ctx* x : g->ctxs) ma_ctx_destroy(x);
        delete g;
        return hip_fail(e, "hipHostMalloc(group records)", __FILE__, __LINE__);
    }
    *out_group = g;
    return MA_OK;
}

void ma_group_destroy(ma_group* group) {
    if (!group) return;
    for (ma_ctx* c : group->ctxs) ma_ctx_destroy(c);
    if (group->records) (void)hipHostFree(group->records);
    delete group;
}

int32_t ma_group_size(ma_group* group) { return group ? (int32_t)group->ctxs.size() : 0; }

ma_ctx* ma_group_ctx(ma_group* group, int32_t index) {
    if (!group || index < 0 || (size_t)index >= group->ctxs.size()) return nullptr;
    return group->ctxs[(size_t)index];
}

ma_status ma_group_sum_i64(ma_group* group, const int64_t* const* chunk_data, const size_t* chunk_lens,
                           const uint8_t* const* chunk_masks, const size_t* chunk_mask_offsets, int64_t* out_sum,
                           uint64_t* out_valid_count) {
    MA_REQUIRE(group && chunk_data && chunk_lens, MA_ERR_INVALID_ARGUMENT, "NULL argument");
    ma_status st = group_run(group, chunk_lens, [&](size_t i, uint64_t* rec) {
        return ma_i64_sum(group->ctxs[i], chunk_data[i], chunk_lens[i], chunk_masks ? chunk_masks[i] : nullptr,
                          chunk_mask_offsets ? chunk_mask_offsets[i] : 0, -1, (int64_t*)&rec[0], &rec[2]);
    });
    MA_TRY(st);
    uint64_t sum = 0, cnt = 0;
    for (size_t i = 0; i < group->ctxs.size(); ++i) {
        sum += group->records[4 * i];
        cnt += group->records[4 * i + 2];
    }
    if (out_sum) *out_sum = (int64_t)sum;
    if (out_valid_count) *out_valid_count = cnt;
    return MA_OK;
}

ma_status ma_group_sum_f64(ma_group* group, const double* const* chunk_data, const size_t* chunk_lens,
                           const uint8_t* const* chunk_masks, const size_t* chunk_mask_offsets, double* out_sum,
                           uint64_t* out_valid_count) {
    MA_REQUIRE(group && chunk_data && chunk_lens, MA_ERR_INVALID_ARGUMENT, "NULL argument");
    ma_status st = group_run(group, chunk_lens, [&](size_t i, uint64_t* rec) {
        return ma_f64_sum_dd(group->ctxs[i], chunk_data[i], chunk_lens[i], chunk_masks ? chunk_masks[i] : nullptr,
                             chunk_mask_offsets ? chunk_mask_offsets[i] : 0, -1, (double*)&rec[0], (double*)&rec[1], &rec[2]);
    });
    MA_TRY(st);
    double hi = 0.0, lo = 0.0;
    uint64_t cnt = 0;
    for (size_t i = 0; i < group->ctxs.size(); ++i) {
        double h, l;
        memcpy(&h, &group->records[4 * i], 8);
        memcpy(&l, &group->records[4 * i + 1], 8);
        two_sum_acc(hi, lo, h, l);
        cnt += group->records[4 * i + 2];
    }
    const bool finite = (hi - hi == 0.0) && (lo - lo == 0.0);
    if (out_sum) *out_sum = finite ? hi + lo : hi;
    if (out_valid_count) *out_valid_count = cnt;
    return MA_OK;
}

}  // extern "C"
