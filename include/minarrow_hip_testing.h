/* include/minarrow_hip_testing.h — the FAULT HOOKS of libminarrow_hip.so, apart from the product's ABI (include/minarrow_hip.h).
 *
 * What they are for: a one-GPU box has no lost peer, no fabric fault, no second device. These entry points make the library
 * produce, on purpose, the failures that the bounded waits, the abort / rebuild paths and the residency checks exist for —
 * the multi-GPU twin of a Rayon worker that panics in the reference's partitioned reduction
 * (benches/benchmark_parallel_simd.rs:81-125). tests/ and bench.py's MA_BENCH_FAULT use them; no binding should
 * (bindings/minarrow_hip_sys.rs does not declare them).
 *
 * They are INERT by default: every function below returns MA_ERR_UNSUPPORTED ("test hooks are disabled") unless the environment
 * held MINARROW_HIP_TEST_HOOKS=1 when the library was loaded — a host cannot corrupt or stall its own exchange through a public
 * symbol by accident. The symbols are always exported (one library, one ABI version). */
#ifndef MINARROW_HIP_TESTING_H
#define MINARROW_HIP_TESTING_H

#include "minarrow_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* 1 when the hooks are live in this process (MINARROW_HIP_TEST_HOOKS=1 at load), else 0. */
int32_t ma_test_hooks_enabled(void);

/* Makes the residency and peer checks treat `member` as if its device were HIP device `hip_device`, with
 * (peer_capable != 0) or without peer access between it and every other member; nothing is launched differently. Lets a
 * one-GPU box exercise the refusals a multi-GPU node produces (a chunk resident on the wrong GPU, an owner without a
 * link to the destination). */
ma_status ma_group_test_set_member_device(ma_group* group, int32_t member, int32_t hip_device, int32_t peer_capable);
/* The next ma_group_exchange fails on `member` in front of its all-gather, as a lost device would make it: with
 * per-member issue threads the other members have enqueued their collectives by then, so the group aborts every
 * communicator (ncclCommAbort), marks itself broken — ma_group_exchange / ma_group_synchronize return MA_ERR_DEVICE from
 * then on instead of blocking on a collective that cannot complete — until ma_group_rebuild_exchange or its destruction. */
ma_status ma_group_test_fail_next_exchange(ma_group* group, int32_t member);
/* The next ma_group_exchange holds `member`'s exchange in front of its all-gather behind a word nobody writes
 * (hipStreamWaitValue64) — what a lost peer or a fabric fault looks like to the waiting host: ma_group_synchronize would
 * block for good, ma_group_synchronize_for returns after its deadline. The abort path (and ma_group_destroy) releases the
 * word, so the stream runs empty. With the host fold the member's scan stream is held. MA_ERR_UNSUPPORTED (at the exchange)
 * on a runtime without stream memory operations. */
ma_status ma_group_test_stall_next_exchange(ma_group* group, int32_t member);
/* The next ma_group_exchange flips one word of the records `member` gathered, in front of its fold: finals
 * that are wrong on that member only and no error anywhere — what a host's own check of a set-up step is for. (Host fold: the
 * job's integer finals are flipped.) */
ma_status ma_group_test_corrupt_next_exchange(ma_group* group, int32_t member);
/* As ma_group_test_stall_next_exchange / _corrupt_next_exchange, for this rank's next ma_comm_sum_exchange* call. */
ma_status ma_comm_test_stall_next_exchange(ma_comm* comm);
ma_status ma_comm_test_corrupt_next_exchange(ma_comm* comm);
/* The next scan of the pipeline (ma_scan_lanes_sum_fused / ma_scan_lanes_sum) is enqueued behind a word nobody writes: a gate that
 * never opens, and every later scan is gated on a stamp that scan never stores. ma_scan_lanes_synchronize would block for good;
 * ma_scan_lanes_synchronize_for returns after its deadline, having released the word. */
ma_status ma_scan_lanes_test_hold_next_scan(ma_scan_lanes* lanes);

/* ------------------------------------------------------------------------------------------------
 * Not a fault hook, but internal all the same:
 * ma_test_pow_series evaluates, element by element, the series float Power is built from — the device stand-ins for the
 * host libm calls of `(rhs * lhs.ln()).exp()` (src/kernels/arithmetic/std.rs:153, simd.rs:570,585):
 *   which 0: in = f64 x,  out[i] = ln x           as the f64 Power path computes it (pow_f64_ln)
 *   which 1: in = f32 a,  out[i] = ln a  in f64   as the f32 Power path computes it before rounding to f32 (pow_f32_ln)
 *   which 2: in = f32 y,  out[i] = exp y in f64   as the f32 Power path computes it before rounding to f32 (pow_f32_exp;
 *                                                 |y| <= 150)
 * tests/test_gpu_pow_series.py holds them to tests/golden/pow_series_kat.npz (tools/check_pow_series.py: 265-bit decimal
 * arithmetic). Buffers may be host or device memory.
 * ---------------------------------------------------------------------------------------------- */
ma_status ma_test_pow_series(ma_ctx* ctx, int32_t which, const void* in, double* out, size_t n);

#ifdef __cplusplus
} /* extern "C" */
#endif
#endif /* MINARROW_HIP_TESTING_H */
