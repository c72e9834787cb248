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

#include "ma_common.hpp"

namespace ma {

struct RcclApi {
    ncclResult_t (*GetVersion)(int*) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    const char* path = "";  // which library name resolved
};

// The process-wide RCCL entry points, or nullptr (with the thread's error string set) when the library cannot be
// opened. Thread safe; the outcome of the first attempt is kept.
const RcclApi* rccl();

ma_status rccl_fail(ncclResult_t r, const char* what, const char* file, int line);

}  // namespace ma

#define MA_NCCL(api, call)                                                          \
    do {                                                                            \
        ncclResult_t _r = (api)->call;                                              \
        if (_r != ncclSuccess) return ::ma::rccl_fail(_r, #call, __FILE__, __LINE__); \
    } while (0)
