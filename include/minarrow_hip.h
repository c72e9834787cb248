/*
 * minarrow_hip.h — C ABI of the MI355X (gfx950) kernel layer for Minarrow's src/kernels hot path.
 *
 * This is the drop-in boundary. Every entry point replaces one slice-in / slice-out Rust function of the
 * reference (cited as `file:line`, relative to the reference repository) or one of the hand-written
 * reductions in its bench binaries. Plain pointers and sizes only; no C++/torch types.
 *
 * Conventions
 * -----------
 *  - Every compute call takes a `ma_ctx*` (one device + one HIP stream) and returns `ma_status`.
 *  - Buffers may live anywhere the device can reach:
 *      * device memory (hipMalloc / ma_dev_alloc / a torch CUDA tensor's data_ptr)  → used in place (fast path),
 *      * pinned host memory (ma_alloc64_pinned — the Vec64 stand-in)                → read/written in place over PCIe,
 *      * ordinary pageable host memory (a Rust `&[T]`, a numpy array)               → staged through device scratch.
 *    The library classifies each pointer with hipPointerGetAttributes; there is NO CPU compute fallback.
 *  - The caller allocates every output (reference: `Vec64::with_capacity(len); set_len(len)`,
 *    src/kernels/arithmetic/dispatch.rs:88-89; `Bitmask::new_set_all(len,true)`, :92).
 *  - Validity bitmaps are Arrow layout (src/structs/bitmask.rs:66-71): bit i = byte i>>3, bit i&7, 1 = valid.
 *    They are addressed as (bits pointer, bit offset, bit length). Output bitmaps must have room for
 *    8*ceil(len/64) bytes (the reference also reads/writes them as whole u64 words,
 *    src/structs/bitmask.rs:266-277); bits >= len are written as 0.
 *  - Synchronous by default: a call returns after its result is available to the host. With
 *    ma_ctx_set_async(ctx,1) calls only enqueue work on the context's stream; scalar outputs must then be
 *    device-reachable memory and errors detected on the device are reported by ma_ctx_synchronize().
 *  - Thread safety: a ctx may be shared between host threads (the reference's kernels are re-entrant,
 *    src/kernels/arithmetic/mod.rs:29-31). A synchronous call that finds the context busy in another thread runs on
 *    one of the context's internal lanes (own stream + reduction scratch, up to MINARROW_HIP_LANES = 4 per context,
 *    created on demand), so threads overlap instead of queueing behind each other's wait. Calls in async mode, during
 *    capture, or on a caller-owned stream (ma_ctx_create_on_stream) serialise on the context's one stream — ordering
 *    is the contract there. Independent ctxs are fully independent. No hidden global scratch.
 *  - Environment (read once): MINARROW_HIP_DEVICES ("2,3": the library's ordinal i is HIP device
 *    list[i]), MINARROW_HIP_MIN_ROWS (ma_min_device_rows), MINARROW_HIP_STAGING_TILE (bytes, ma_ctx_set_staging_tile's
 *    default), MINARROW_HIP_LANES, MINARROW_HIP_GROUP_EXCHANGE ("rccl"), MINARROW_HIP_FENCED_REDUCE, MINARROW_HIP_POLL_US,
 *    MINARROW_HIP_PINNED_POOL_BYTES,
 *    MINARROW_HIP_DEV_POOL_BYTES (block-cache limits).
 *
 * Status codes mirror KernelError (src/enums/error.rs:157-187) as far as the numeric kernels can raise them.
 */
#ifndef MINARROW_HIP_H
#define MINARROW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 3): + ma_ctx_hip_device, ma_pointer_device, ma_group_issue_kind, ma_group_peer_access; ma_ctx_device now
 * returns the LIBRARY ordinal (round-trip safe through ma_ctx_create); group calls are issued by per-member threads.
 */
/* 3 (round 4): + ma_ctx_mark / ma_ctx_mark_elapsed_ms (timing marks), ma_sum_fused (several long columns in ONE launch),
 * ma_sum_fused_stamped + ma_stamp_alloc / _free + ma_comm_sum_exchange_overlapped_on_stamp (event-free hand-off to the
 * exchange stream), ma_group_enqueue_sum_table, ma_group_exchange_stats, ma_comm_exchange_stats; ma_hip_runtime_path, ma_rccl_path; ma_dev_alloc_output searches only when asked to
 * (MINARROW_HIP_OUTPUT_SEARCH=1 / ma_dev_output_search).
 * A binding compares ma_abi_version() with the MA_ABI_VERSION it was generated from. */
/* 4 (round 5): + bounded waits and first-contact safety for the multi-GPU forms: ma_group_synchronize_for,
 * ma_group_is_broken, ma_group_flags, ma_group_set_handoff / ma_group_handoff, ma_group_rebuild_exchange, ma_group_selftest
 * (+ ma_selftest_report), ma_comm_synchronize_for, ma_comm_abort, ma_comm_is_broken, ma_comm_selftest, ma_stamp_is_signal;
 * testing hooks ma_group_test_stall_next_exchange / _corrupt_next_exchange, ma_comm_test_stall_next_exchange /
 * _corrupt_next_exchange (since round 6 in minarrow_hip_testing.h). A broken group no longer has to be destroyed: ma_group_rebuild_exchange gives it a fresh exchange.
 * + ma_sum_fused_stamped_early, ma_ctx_wait_value, MA_GROUP_SCAN_LANES, ma_group_join_lanes (consecutive scans on two streams, the
 * next one gated on the early stamp of the one before); ma_scan_lanes_* (the same pipeline for a host that drives one GPU
 * without a group). */
/* 5 (round 6): + ma_scan_lanes_synchronize_for, ma_scan_lanes_is_broken (the single-GPU pipeline's waits are bounded too;
 * ma_scan_lanes_destroy is). The testing hooks (ma_*_test_*, ma_test_pow_series) moved to include/minarrow_hip_testing.h and are
 * inert (MA_ERR_UNSUPPORTED) unless MINARROW_HIP_TEST_HOOKS=1 was in the environment when the library was loaded.
 * MINARROW_HIP_RCCL_PATH names the collective library to open instead of the system's RCCL (ma_rccl_path). */
#define MA_ABI_VERSION 5

typedef struct ma_ctx ma_ctx;
typedef int32_t ma_status;

enum {
    MA_OK = 0,
    MA_ERR_LENGTH_MISMATCH = 1,  /* KernelError::LengthMismatch — src/utils.rs:163-171 */
    MA_ERR_DIVIDE_BY_ZERO = 2,   /* dense integer Div/Rem/FloorDiv saw a zero divisor: the reference panics
                                    (src/kernels/arithmetic/std.rs:53-77); output contents are unspecified */
    MA_ERR_UNSUPPORTED = 3,      /* KernelError::UnsupportedType */
    MA_ERR_INVALID_ARGUMENT = 4, /* KernelError::InvalidArguments: null pointer, misaligned pointer, bad op code */
    MA_ERR_DEVICE = 5,           /* a HIP runtime call failed; see ma_last_error_string() */
    MA_ERR_NO_DEVICE = 6         /* no usable GPU: the library never falls back to the CPU */
};

/* ArithmeticOperator — discriminant order of src/enums/operators.rs:18-48 */
enum {
    MA_OP_ADD = 0,
    MA_OP_SUBTRACT = 1,
    MA_OP_MULTIPLY = 2,
    MA_OP_DIVIDE = 3,
    MA_OP_REMAINDER = 4,
    MA_OP_POWER = 5,
    MA_OP_FLOORDIV = 6
};

/* LogicalOperator — src/enums/operators.rs:86-100 */
enum { MA_LOGICAL_AND = 0, MA_LOGICAL_OR = 1, MA_LOGICAL_XOR = 2 };

/* ------------------------------------------------------------------------------------------------
 * Library / context
 * ---------------------------------------------------------------------------------------------- */

int32_t ma_abi_version(void);
/* The libamdhip64 this library is running on (the path the dynamic loader resolved its HIP entry points to). A process may
 * hold more than one HIP runtime — PyTorch bundles its own — and device pointers, streams and RCCL communicators belong to
 * exactly one of them. */
const char* ma_hip_runtime_path(void);
/* Number of devices the library may use: the visible HIP devices, or the entries of MINARROW_HIP_DEVICES (0 when there
 * is none; never initialises a device context). Device ordinals of this ABI index that list. */
int32_t ma_device_count(void);
/* Column length below which a host wrapper should keep the reference's CPU kernels, per kind of call — derived from
 * measurements (INTEGRATION.md §5 has the table and the profiles/ files): a synchronous call on a resident column costs
 * ~12 us whatever the size, ONE host thread sums 18.8 cache-resident i64 rows per ns (the reference's 1000-row scalar sum:
 * 85 ns, src/lib.rs:58) and adds two f64 columns at 2-4 rows per ns:
 *   MA_KIND_REDUCTION     sums / means         262 144 rows  (MINARROW_HIP_MIN_ROWS)
 *   MA_KIND_ELEMENTWISE   a (+) b, a (+) x      32 768 rows  (MINARROW_HIP_MIN_ROWS_ELEMENTWISE)
 *   MA_KIND_BITMASK_SCAN  popcount / all_true  2 097 152 bits (MINARROW_HIP_MIN_BITS_SCAN)
 * ma_min_device_rows() is the reduction figure. For MANY small columns use one ma_sum_columns call (0.24 us per column)
 * or a replayed hipGraph rather than a call each. Advice for the host shim only — the library itself has no CPU path and
 * accepts any length. */
enum { MA_KIND_REDUCTION = 0, MA_KIND_ELEMENTWISE = 1, MA_KIND_BITMASK_SCAN = 2 };
int64_t ma_min_device_rows_for(int32_t kind);
int64_t ma_min_device_rows(void);
/* Thread-local description of the last non-OK status returned on this thread. Never NULL. */
const char* ma_last_error_string(void);
const char* ma_status_name(ma_status s);

/* One context = one device + one stream + that stream's reduction scratch. */
ma_status ma_ctx_create(int32_t device_ordinal, ma_ctx** out_ctx);
/* As above but enqueue on a caller-owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream). */
ma_status ma_ctx_create_on_stream(int32_t device_ordinal, void* hip_stream, ma_ctx** out_ctx);
void ma_ctx_destroy(ma_ctx* ctx);
ma_status ma_ctx_synchronize(ma_ctx* ctx);
ma_status ma_ctx_set_async(ma_ctx* ctx, int32_t enabled);
void* ma_ctx_stream(ma_ctx* ctx);
/* The context's device as the LIBRARY ordinal (an index into the MINARROW_HIP_DEVICES list — the number ma_ctx_create and
 * ma_group_create take, so it round-trips), and as the HIP runtime's own ordinal (what hipSetDevice takes). */
int32_t ma_ctx_device(ma_ctx* ctx);
int32_t ma_ctx_hip_device(ma_ctx* ctx);
int32_t ma_ctx_compute_units(ma_ctx* ctx);
/* Lanes the context has grown so far, itself included (1 until two synchronous calls overlap; at most MINARROW_HIP_LANES). */
int32_t ma_ctx_lane_count(ma_ctx* ctx);
/* Launch geometry for the streaming kernels: workgroups per CU (0 = built-in default). */
ma_status ma_ctx_set_blocks_per_cu(ma_ctx* ctx, int32_t blocks_per_cu);
/* Tuning harness: absolute workgroup count for the streaming kernels (0 = built-in default). */
ma_status ma_ctx_set_grid(ma_ctx* ctx, int32_t workgroups);
/* Forces one of two PRODUCT paths that the library otherwise picks by the size or shape of its input, so that a test (or a host
 * that knows better) reaches the path a 10^9-row input takes with an input a CPU check can follow. 0 = the library's own choice.
 * The six FORM bits, live in every build:
 *     16 / 32   SuperArray ops: the 4 x 16-byte / the 8 x 16-byte tile per lane, whatever the chunk lengths
 *    128 / 256  chunk lists (consolidate, SuperArray ops): the tile form (one searched table) / the chunk-per-workgroup form
 *  16384        ma_sum_columns: never the fused few-long-columns scan (the general two-launch path)
 *  65536        ma_sum_columns: the fused few-long-columns scan whatever the total size
 * Any other bit is a TUNING form — an older launch shape, another unroll depth or load pacing, the piece-interleaved mapping,
 * the fenced publish, the early stamp's trigger: measured against the defaults and not kept (profiles/HISTORY.md). Those exist
 * only in a library built with `make -C minarrow_amd/csrc TUNING=1` (build/tuning/libminarrow_hip.so, what tools/ sweeps load);
 * the shipped library does not contain their kernels and returns MA_ERR_UNSUPPORTED for them instead of silently running a slower form. */
ma_status ma_ctx_set_variant(ma_ctx* ctx, int32_t variant);
/* Host-resident (pageable) operands of the elementwise entry points and of the sum / mean reductions — a Rust &[T] /
 * Vec64<T> that was not allocated with ma_alloc64_pinned; the reference's kernels read such slices in place
 * (src/kernels/arithmetic/dispatch.rs:74-133) — cross PCIe in tiles of tile_bytes per operand through a ring of
 * device buffers owned by the context: the copy-in of tile k+1, the kernels of tile k and the copy-out of tile k-1
 * overlap, and the device footprint is 12 tiles whatever the column size (a reduction folds its per-tile
 * {sum | hi, lo, count} records in tile order, so float sums keep their 1-ULP bound). Pinned operands (ma_alloc64_pinned) of a synchronous call take the same route — the copy
 * engines fill both directions of the link, a kernel addressing host memory in place does not; an async context
 * leaves them in place so that the call can return early. Default 32 MiB; calls shorter than two tiles, and every call
 * when tile_bytes == 0, stage whole pageable operands in temporary device buffers (and use pinned ones in place)
 * instead. Results are identical either way. */
ma_status ma_ctx_set_staging_tile(ma_ctx* ctx, size_t tile_bytes);

/* HIP-event timing on the context's stream (bench.py's roofline leg uses these). */
ma_status ma_ctx_timer_start(ma_ctx* ctx);
ma_status ma_ctx_timer_stop(ma_ctx* ctx);
/* Waits for the stop event; milliseconds between start and stop. */
ma_status ma_ctx_timer_elapsed_ms(ma_ctx* ctx, float* out_ms);
/* Timing marks, for a host without HIP headers that wants per-kernel durations inside a longer timed region: ma_ctx_mark
 * records mark `index` (0 <= index < MA_CTX_MAX_MARKS; created on first use, re-recordable) on the context's stream behind
 * everything enqueued so far; ma_ctx_mark_elapsed_ms waits for mark `to_index` and returns the milliseconds between the two.
 * A mark costs a few microseconds of stream time (one event packet). Not recordable into a graph. */
#define MA_CTX_MAX_MARKS 4096
ma_status ma_ctx_mark(ma_ctx* ctx, int32_t index);
ma_status ma_ctx_mark_elapsed_ms(ma_ctx* ctx, int32_t from_index, int32_t to_index, float* out_ms);

/* ------------------------------------------------------------------------------------------------
 * Memory — the Vec64 stand-in (64-byte aligned, src/lib.rs:99; Cargo.toml:54 vec64 0.4.3) and
 * device-resident buffers
 * ---------------------------------------------------------------------------------------------- */

/* hipHostMalloc-backed, 64-byte aligned, device-mapped. The pointer is valid on the host and in kernels. Blocks of
 * 4 KiB and more (device blocks: 1 MiB and more) are recycled through a size-class cache (sizes rounded up by at most 12.5 %; pinning pages costs ~40 ms
 * per 256 MiB, so a Vec64 allocator
 * built on raw hipHostMalloc would be 50x slower than malloc for large columns): ma_free_pinned parks them,
 * ma_alloc64_pinned reuses them. ma_pinned_pool_trim(keep) releases cached blocks down to `keep` bytes and makes that the
 * new cache limit (default 2 GiB; 0 = no caching). Thread safe. */
/* Contract of ma_free_pinned (and ma_dev_free): every piece of work that touches the block — kernels an ASYNC context
 * still has in flight, copies on other streams — must have completed first; a parked block is handed out again at
 * once, where hipHostFree used to wait. Freeing a block twice is refused (MA_ERR_INVALID_ARGUMENT).
 * ma_pinned_pool_set_limit changes the cache limit without releasing anything (undoes a trim's lowering). */
ma_status ma_alloc64_pinned(size_t bytes, void** out_ptr);
ma_status ma_free_pinned(void* ptr);
ma_status ma_pinned_pool_trim(size_t keep_bytes);
ma_status ma_pinned_pool_set_limit(size_t limit_bytes);
/* Pins an EXISTING host allocation in place (hipHostRegister, portable + mapped) — for a host that cannot change the
 * allocator of buffers it already owns: a Vec64<T> from the stock vec64 crate (Cargo.toml:54), a foreign buffer behind
 * SharedBuffer::from_owner (src/structs/shared_buffer/mod.rs:187-206), an mmap. From then on the range classifies as
 * pinned: kernels and the copy engines address it directly, no pageable staging. Registration costs a page-table walk
 * (milliseconds per GiB): do it once per long-lived buffer, not per call. ma_host_unregister before the memory is
 * freed. `ptr` need not be page aligned; a range may be registered once. */
ma_status ma_host_register(void* ptr, size_t bytes);
ma_status ma_host_unregister(void* ptr);
/* Device (HBM) blocks. ma_dev_free waits for the context's stream, then — for blocks of 1 MiB and more — parks the block
 * in a per-device size-class cache instead of calling hipFree, which synchronises the whole device and stalls every other
 * stream (~160 us); ma_dev_alloc reuses parked blocks and, when HBM runs out, releases the cache and retries.
 * ma_dev_pool_trim(ctx, keep) releases parked blocks of the context's device down to `keep` bytes and makes that the new
 * limit (default 16 GiB per device). A parked block may be handed out again at once: work that OTHER contexts or streams
 * enqueued on it must have completed before it is freed (hipFree used to hide that by stalling the device). */
ma_status ma_dev_alloc(ma_ctx* ctx, size_t bytes, void** out_dev_ptr);
/* A block meant to be WRITTEN by the streaming kernels — the `out` of apply_* (the reference allocates it itself:
 * `Vec64::with_capacity(len)`, src/kernels/arithmetic/dispatch.rs:88-89), a consolidated column. BY DEFAULT this is
 * ma_dev_alloc (round 4): the placement search described below is opt-in — MINARROW_HIP_OUTPUT_SEARCH=1 in the environment,
 * or ma_dev_output_search(1) — because what it buys depends on the blocks a process happens to draw (+5-10 % on the write
 * stream on some boxes, nothing on others, for ~8 ms and up to 25 % of free HBM held while it runs); a caller that brings
 * its own `out` gets its block's rate either way, and every read+write figure this repository reports is also given as a
 * fraction of a plain copy into the same block. ma_dev_output_search(enabled): 1 / 0 switch the search on / off for the
 * process, a negative value only queries; returns the previous setting. With the search on: on MI355X the write
 * rate of a region of HBM is a property of where the driver placed it: ~three quarters of the regions write at 5.4-5.8
 * TB/s under the kernels' store pattern, the rest at 6.3-6.8, while all of them read at 7.1-7.3. For
 * blocks of 256 MiB and more this entry point tries 6 candidate blocks
 * (parked blocks of the size class first, then fresh ones), and up to 6 more
 * while the best is still below the good rate. Candidates are held while the search runs (a block given back would be the
 * next one handed out), so the search is BOUNDED: the blocks alive at one time never exceed 25 % of the HBM that is free
 * when the call starts, the search stops at that many, and a request two of which do not
 * fit the bound — a 64-GB consolidated column on a 288-GB device — is served by the plain allocator without measuring
 * anything. Each candidate's write rate is measured once with three launches of the store pattern (~1.3 ms each per 8 GB;
 * the rate is remembered for as long as the library owns the block); the search stops at the first block that reaches the
 * good rate = 0.97 x the device's own write ceiling (the placement-independent tight-front pattern, measured once per device;
 * the tuning build reads the four figures from MINARROW_HIP_OUTPUT_*), returns the fastest and parks the others in the block cache, where ma_dev_alloc
 * picks them up as inputs. The probe tells the two classes apart but is no promise: on boxes where every candidate was a slow
 * region the best block wrote at 5.8 TB/s. The block's contents are undefined (the probe writes zeros). out_write_gbps (may
 * be NULL) receives the chosen block's measured rate, 0 when nothing was measured. Free with ma_dev_free.
 * ma_dev_alloc_output_stats: what the calling thread's last search cost — wall time, bytes held at its peak, blocks measured /
 * considered, the good rate it compared against (all 0 when the plain path was taken; any pointer may be NULL). */
ma_status ma_dev_alloc_output(ma_ctx* ctx, size_t bytes, void** out_dev_ptr, float* out_write_gbps);
int32_t ma_dev_output_search(int32_t enabled);
ma_status ma_dev_alloc_output_stats(double* out_search_ms, size_t* out_held_peak_bytes, int32_t* out_blocks_measured,
                                    int32_t* out_blocks_considered, float* out_good_gbps);
ma_status ma_dev_free(ma_ctx* ctx, void* dev_ptr);
ma_status ma_dev_pool_trim(ma_ctx* ctx, size_t keep_bytes);
/* Cache limit of the context's device without releasing anything. Every device allocation the library makes for
 * itself (scratch, staging rings, context state) releases the parked blocks and retries when HBM is full. */
ma_status ma_dev_pool_set_limit(ma_ctx* ctx, size_t limit_bytes);
ma_status ma_dev_upload(ma_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
ma_status ma_dev_download(ma_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
ma_status ma_dev_memset(ma_ctx* ctx, void* dst_dev, int32_t byte_value, size_t bytes);
/* Device-to-device copy by the runtime's copy path (hipMemcpyAsync on the context's stream; follows the sync / async
 * mode). bench.py times it next to the read+write kernels as the same-process reference rate. */
ma_status ma_dev_copy(ma_ctx* ctx, void* dst_dev, const void* src_dev, size_t bytes);
/* 0 = pageable host, 1 = pinned/registered host, 2 = device, 3 = managed. */
int32_t ma_pointer_kind(const void* ptr);
/* The HIP ordinal of the device whose memory `ptr` points into; -1 for host memory (pageable or pinned), NULL and
 * pointers the runtime does not know. The residency rule of the ma_group_* calls is stated in terms of it. */
int32_t ma_pointer_device(const void* ptr);

/* ------------------------------------------------------------------------------------------------
 * Synthetic inputs generated in place on the device (patterns of
 * benches/benchmark_parallel_simd.rs:103,115). `dst` must be device-reachable.
 * ---------------------------------------------------------------------------------------------- */

/* dst[i] = start + i                                   (Vec64<i64> = (0..N).collect()) */
ma_status ma_synth_iota_i64(ma_ctx* ctx, int64_t* dst, size_t n, int64_t start);
/* dst[i] = (double)(start + i)                         ((0..N).map(|x| x as f64)) */
ma_status ma_synth_iota_f64(ma_ctx* ctx, double* dst, size_t n, int64_t start);
ma_status ma_synth_iota_i32(ma_ctx* ctx, int32_t* dst, size_t n, int32_t start);
ma_status ma_synth_iota_f32(ma_ctx* ctx, float* dst, size_t n, int32_t start);
/* dst[i] = splitmix64(seed + first_index + i) reinterpreted as i64 */
ma_status ma_synth_splitmix_i64(ma_ctx* ctx, int64_t* dst, size_t n, uint64_t seed, uint64_t first_index);
/* dst[i] = uniform [-1,1): (splitmix64(seed + first_index + i) >> 11) * 2^-52 - 1 */
ma_status ma_synth_splitmix_f64(ma_ctx* ctx, double* dst, size_t n, uint64_t seed, uint64_t first_index);
/* Validity bitmap: bit i = (splitmix64(seed + first_index + i) % null_every != 0); whole u64 words are
 * written (8*ceil(n_bits/64) bytes), bits >= n_bits zero. null_every = 10 gives ~10 % nulls. */
ma_status ma_synth_validity(ma_ctx* ctx, uint8_t* dst_bits, size_t n_bits, uint64_t seed, uint64_t first_index,
                            uint32_t null_every);

/* ------------------------------------------------------------------------------------------------
 * Reductions — the reference's only sum implementations are in its bench binaries:
 *   simd_sum_i64 / simd_sum_f64        benches/benchmark_parallel_simd.rs:44-59, 63-78
 *   rayon_simd_sum_{i64,f64}           benches/benchmark_parallel_simd.rs:81-98
 *   4x-unrolled variants               benches/hotloop_benchmark_simd.rs:56-174
 *   scalar loops                       benches/hotloop_benchmark_std.rs:49-57
 * Hand-off shape follows NumericArrayV::guarantee_f64 (src/structs/views/collections/
 * numeric_array_view.rs:302-317): `data` already points at the window, `mask_bits` is the un-windowed
 * validity buffer (or NULL) with the window's first bit at `mask_bit_offset`, `null_count` is the cached
 * null count of the window (-1 = unknown, 0 = take the dense kernel).
 *
 * Semantics: integers wrap (two's complement) exactly like the reference's release build; i32/u32 are
 * accumulated in 64 bits (the wrapping 32-bit sum is the low half). f32/f64 are accumulated in
 * double-double so the result is within 1 ULP of the exactly rounded sum whatever the order.
 * Bitmask-gated sum, valid-count and mean do not exist in the reference: sum = Σ data[i] over set bits,
 * count = popcount, mean = sum/count (NaN when count == 0).
 * Any output pointer may be NULL.
 * ---------------------------------------------------------------------------------------------- */

ma_status ma_i64_sum(ma_ctx* ctx, const int64_t* data, size_t n, const uint8_t* mask_bits, size_t mask_bit_offset,
                     int64_t null_count, int64_t* out_sum, uint64_t* out_valid_count);
ma_status ma_u64_sum(ma_ctx* ctx, const uint64_t* data, size_t n, const uint8_t* mask_bits, size_t mask_bit_offset,
                     int64_t null_count, uint64_t* out_sum, uint64_t* out_valid_count);
ma_status ma_i32_sum(ma_ctx* ctx, const int32_t* data, size_t n, const uint8_t* mask_bits, size_t mask_bit_offset,
                     int64_t null_count, int64_t* out_sum, uint64_t* out_valid_count);
ma_status ma_u32_sum(ma_ctx* ctx, const uint32_t* data, size_t n, const uint8_t* mask_bits, size_t mask_bit_offset,
                     int64_t null_count, uint64_t* out_sum, uint64_t* out_valid_count);
/* The reference's extended_numeric_types (i8 / u8 / i16 / u16 columns, src/enums/collections/numeric_array.rs:81-99):
 * 64-bit wrapping sums like the 32-bit types; 16 / 8 rows per 16-byte load are summed inside 32-bit registers. */
ma_status ma_i16_sum(ma_ctx* ctx, const int16_t* data, size_t n, const uint8_t* mask_bits, size_t mask_bit_offset,
                     int64_t null_count, int64_t* out_sum, uint64_t* out_valid_count);
ma_status ma_u16_sum(ma_ctx* ctx, const uint16_t* data, size_t n, const uint8_t* mask_bits, size_t mask_bit_offset,
                     int64_t null_count, uint64_t* out_sum, uint64_t* out_valid_count);
ma_status ma_i8_sum(ma_ctx* ctx, const int8_t* data, size_t n, const uint8_t* mask_bits, size_t mask_bit_offset,
                    int64_t null_count, int64_t* out_sum, uint64_t* out_valid_count);
ma_status ma_u8_sum(ma_ctx* ctx, const uint8_t* data, size_t n, const uint8_t* mask_bits, size_t mask_bit_offset,
                    int64_t null_count, uint64_t* out_sum, uint64_t* out_valid_count);
ma_status ma_f64_sum(ma_ctx* ctx, const double* data, size_t n, const uint8_t* mask_bits, size_t mask_bit_offset,
                     int64_t null_count, double* out_sum, uint64_t* out_valid_count);
ma_status ma_f32_sum(ma_ctx* ctx, const float* data, size_t n, const uint8_t* mask_bits, size_t mask_bit_offset,
                     int64_t null_count, double* out_sum, uint64_t* out_valid_count);
/* As ma_f64_sum / ma_f32_sum but returns the unevaluated double-double (hi + lo) so partial sums of row
 * chunks (one per GPU) can be combined without losing the 1-ULP guarantee. */
ma_status ma_f64_sum_dd(ma_ctx* ctx, const double* data, size_t n, const uint8_t* mask_bits, size_t mask_bit_offset,
                        int64_t null_count, double* out_hi, double* out_lo, uint64_t* out_valid_count);
ma_status ma_f32_sum_dd(ma_ctx* ctx, const float* data, size_t n, const uint8_t* mask_bits, size_t mask_bit_offset,
                        int64_t null_count, double* out_hi, double* out_lo, uint64_t* out_valid_count);

/* Arithmetic mean = sum / valid_count as f64 (NaN when valid_count == 0). */
ma_status ma_i64_mean(ma_ctx* ctx, const int64_t* data, size_t n, const uint8_t* mask_bits, size_t mask_bit_offset,
                      int64_t null_count, double* out_mean, uint64_t* out_valid_count);
ma_status ma_u64_mean(ma_ctx* ctx, const uint64_t* data, size_t n, const uint8_t* mask_bits, size_t mask_bit_offset,
                      int64_t null_count, double* out_mean, uint64_t* out_valid_count);
ma_status ma_i32_mean(ma_ctx* ctx, const int32_t* data, size_t n, const uint8_t* mask_bits, size_t mask_bit_offset,
                      int64_t null_count, double* out_mean, uint64_t* out_valid_count);
ma_status ma_u32_mean(ma_ctx* ctx, const uint32_t* data, size_t n, const uint8_t* mask_bits, size_t mask_bit_offset,
                      int64_t null_count, double* out_mean, uint64_t* out_valid_count);
ma_status ma_i16_mean(ma_ctx* ctx, const int16_t* data, size_t n, const uint8_t* mask_bits, size_t mask_bit_offset,
                      int64_t null_count, double* out_mean, uint64_t* out_valid_count);
ma_status ma_u16_mean(ma_ctx* ctx, const uint16_t* data, size_t n, const uint8_t* mask_bits, size_t mask_bit_offset,
                      int64_t null_count, double* out_mean, uint64_t* out_valid_count);
ma_status ma_i8_mean(ma_ctx* ctx, const int8_t* data, size_t n, const uint8_t* mask_bits, size_t mask_bit_offset,
                     int64_t null_count, double* out_mean, uint64_t* out_valid_count);
ma_status ma_u8_mean(ma_ctx* ctx, const uint8_t* data, size_t n, const uint8_t* mask_bits, size_t mask_bit_offset,
                     int64_t null_count, double* out_mean, uint64_t* out_valid_count);
ma_status ma_f64_mean(ma_ctx* ctx, const double* data, size_t n, const uint8_t* mask_bits, size_t mask_bit_offset,
                      int64_t null_count, double* out_mean, uint64_t* out_valid_count);
ma_status ma_f32_mean(ma_ctx* ctx, const float* data, size_t n, const uint8_t* mask_bits, size_t mask_bit_offset,
                      int64_t null_count, double* out_mean, uint64_t* out_valid_count);

/* Per-column {sum, valid count} of n_cols columns of ONE element type in two launches, whatever n_cols is — the
 * per-column reduce of a wide Table or of the chunks of a SuperTable (BASELINE config 5). A launch costs ~4 us on
 * MI355X, so the column-at-a-time loop of the reference (one pass per Array, benches/hotloop_benchmark_std.rs:109-127)
 * is launch-bound for many small columns. format_code as in Arrow ('c','C','s','S','i','I','l','L','f','g'); column i =
 * (col_data[i], col_lens[i]) with optional validity col_masks[i] whose row 0 is bit col_mask_offsets[i] (both
 * tables may be NULL). Outputs are arrays of n_cols entries, any may be NULL: integer formats write the wrapping
 * 64-bit sum to out_sums_i64 and its conversion to out_sums_f64; float formats write out_sums_f64 only (within 1 ULP
 * of the exactly rounded sum, as ma_f64_sum). An empty column yields {0, 0}. Like every other entry point it
 * waits for the results unless the context is in async mode and all buffers are device-reachable.
 * (A FEW LONG device-resident 8-byte columns — up to 16 of 2^21 rows or more, e.g. the batches of a SuperTable column on one
 * GPU — run through ma_sum_fused's kernel, four columns per launch, instead of the segment kernel: same results.)
 * The four tables are read before the call returns (the caller may reuse them at once). From 8192 columns of a segment or
 * less each — a chunked column handed over chunk by chunk — the description of the columns is copied to the device on a stream
 * of the context's own while the stream still runs the call before; in async mode the call then holds the host at most four such
 * calls ahead of the GPU. */
ma_status ma_sum_columns(ma_ctx* ctx, int32_t format_code, size_t n_cols, const void* const* col_data,
                         const size_t* col_lens, const uint8_t* const* col_masks, const size_t* col_mask_offsets,
                         double* out_sums_f64, int64_t* out_sums_i64, uint64_t* out_valid_counts);

/* The sum (and valid count) of ONE column held as a list of chunks — a SuperArray's chunks, one column of the batches of a
 * SuperTable (src/structs/chunked/super_array.rs, super_table.rs): what the reference's bench computes over one slice with
 * `par_chunks(..).map(simd_sum).sum()` (benches/benchmark_parallel_simd.rs:81-98), for a column that is already chunked.
 * Arguments as ma_sum_columns with the chunks in place of the columns; the outputs are single values. Pass 1 is
 * ma_sum_columns' (122 000 chunks of 8192 rows per 10^9 rows: a wave per chunk on in-place descriptors); pass 2 folds the
 * partials of ALL chunks — wrapping adds, or error-free double-double merges in a fixed order, so that the f64 total is within
 * 1 ULP of the exactly rounded sum and reproducible for a given chunk list (a host-side addition of per-chunk rounded sums
 * is neither). An empty list gives {0, 0}. */
ma_status ma_sum_chunks(ma_ctx* ctx, int32_t format_code, size_t n_chunks, const void* const* chunk_data,
                        const size_t* chunk_lens, const uint8_t* const* chunk_masks, const size_t* chunk_mask_offsets,
                        double* out_sum_f64, int64_t* out_sum_i64, uint64_t* out_valid_count);

/* The sums of 1..MA_FUSED_MAX_COLUMNS LONG 8-byte columns — i64 ('l'), u64 ('L'), f64 ('g'), each dense or Bitmask-gated — in ONE
 * launch: the per-column reduce of a table's numeric columns (BASELINE config 5), or the reference's two bench loops
 * (`rayon_simd_sum_i64`, then `rayon_simd_sum_f64`, benches/benchmark_parallel_simd.rs:99-125) over one GPU's row chunk.
 * A sum launch costs ~3.3 us beyond its bytes on MI355X (1.5 us until the first data arrives, 1.8 us of cross-workgroup
 * hand-off after the last row; profiles/r04_probe_epilogue.jsonl) — per column with ma_i64_sum / ma_f64_sum_dd, once per
 * call here: the columns' tiles run through the same workgroups back to back. Per-column semantics are those of
 * ma_i64_sum / ma_u64_sum / ma_f64_sum_dd. Columns and outputs must be device-reachable (device or ma_alloc64_pinned
 * memory: nothing is staged); the call only enqueues on an async context. Integer formats write out[0] = wrapping sum,
 * out[1] = valid count; 'g' writes out[0], out[1] = the (hi, lo) double-double pair (bit patterns), out[2] = valid count —
 * the slots of an exchange record (ma_fold_sum_records) when `out` is &record[0] resp. &record[2]. */
#define MA_FUSED_MAX_COLUMNS 4
typedef struct ma_fused_column {
    const void* data;          /* element pointer of the window (8-byte aligned) */
    size_t n;                  /* rows */
    const uint8_t* mask_bits;  /* Arrow validity bitmap or NULL */
    size_t mask_bit_offset;    /* bit of row 0 */
    int64_t null_count;        /* 0 = known all-valid (dense kernel, like the reference's all_true gate); -1 = unknown */
    int32_t format_code;       /* 'l', 'L' or 'g' */
    int32_t reserved;          /* 0 */
    uint64_t* out;             /* see above */
} ma_fused_column;
ma_status ma_sum_fused(ma_ctx* ctx, size_t n_cols, const ma_fused_column* cols);
/* The same launch, whose final thread additionally stores `stamp_value` to `*stamp` (device-reachable, 8-byte aligned) with a
 * system-scope release BEHIND the results: a hand-off that another stream — or the host — can wait on without an event
 * packet on this context's stream (hipStreamWaitValue64 on memory from ma_stamp_alloc; ma_comm_sum_exchange_overlapped_on_stamp
 * does exactly that). An event record between two back-to-back scans costs the stream several microseconds; the stamp costs
 * it nothing. */
ma_status ma_sum_fused_stamped(ma_ctx* ctx, size_t n_cols, const ma_fused_column* cols, uint64_t* stamp, uint64_t stamp_value);
/* The same, and additionally stamp_value is stored to `*early_stamp` while the launch DRAINS: when the workgroups of two of the
 * chip's eight XCDs have all scanned their rows (grids of up to 96 workgroups: by the first workgroup that has). A scan on ANOTHER
 * context of the device that waits for it (ma_ctx_wait_value(other, early_stamp, stamp_value)) starts its ramp under this launch's
 * stragglers and hand-off instead of behind them, without running beside its whole scan: consecutive independent scans alternated
 * over two contexts this way read 7.04 -> 7.31 TB/s at 125 M rows per column, 5.67 -> 7.04 at 2^24
 * (profiles/r05_probe_early_stamp.jsonl). The trigger is a measured choice: stored by the FIRST workgroup to finish, the overlap
 * grows from step to step in two processes of three until scans run side by side for most of their length and the gain is gone;
 * from the first whole XCD on it does not (profiles/r05_early_mode.txt: 36 of 36 processes at 0.274-0.279 ms per step against
 * 0.286-0.293 on one stream). early_stamp may be word 1 of the line ma_stamp_alloc returned for `stamp` (a device-word stamp is
 * 64 bytes). */
ma_status ma_sum_fused_stamped_early(ma_ctx* ctx, size_t n_cols, const ma_fused_column* cols, uint64_t* stamp, uint64_t stamp_value,
                                     uint64_t* early_stamp);
/* A zeroed word on the context's device that a stream can be made to wait on (hipStreamWaitValue64) and a kernel's
 * system-scope store reaches. A plain 64-byte device line by default: the wait is then a one-wave kernel that spins, which
 * leaves the other streams' dispatches alone; the tuning build's MINARROW_HIP_STAMP_SIGNAL=1 asks for the runtime's signal memory instead (8 bytes,
 * host memory; the wait becomes a packet the command processor polls — measured to hold up the scan stream of an overlapped
 * step by 19-25 %, profiles/r05_share_1gpu.txt — but a host store can release it). Free with ma_stamp_free. */
ma_status ma_stamp_alloc(ma_ctx* ctx, uint64_t** out_stamp);
ma_status ma_stamp_free(ma_ctx* ctx, uint64_t* stamp);
/* Makes the context's stream wait until `*word >= value` (hipStreamWaitValue64; `word` from ma_stamp_alloc): the consumer side of
 * ma_sum_fused_stamped for a host that orders two contexts without events. Enqueue-only. MA_ERR_UNSUPPORTED on a runtime without
 * stream memory operations. Whoever waits must be sure the value will be written: a wait nobody ends holds the stream for good
 * (ma_scan_lanes_* and ma_group_* wrap their own waits in bounded forms: prefer those to gating by hand). */
ma_status ma_ctx_wait_value(ma_ctx* ctx, const uint64_t* word, uint64_t value);
/* 1 when `stamp` (from ma_stamp_alloc) is the runtime's signal memory, 0 when it is a plain device word (the fall-back; a
 * wait on it works the same), -1 when the pointer is not a live stamp. */
int32_t ma_stamp_is_signal(const uint64_t* stamp);

/* ---- back-to-back sums on one GPU as a pipeline ----------------------------------------------------------------------------
 * The reference's hot loop — `for _ in 0..N { sum(&arr) }` over an IntegerArray / FloatArray, one pass per call
 * (benches/hotloop_benchmark_avg_std.rs:48-62 and hotloop_benchmark_avg_simd.rs:34,217: ITERATIONS passes, an i64 and an f64 sum
 * each; the pass itself: hotloop_benchmark_std.rs:109-127) — enqueued on ONE stream pays a launch's fixed
 * cost (ramp + hand-off, ~3.3 us) and the spread of the workgroups' finish times behind every scan: 2^24-row i64 + f64 steps run
 * at 0.70-0.73 of the HBM peak that way, 125 M-row steps at 0.885. A ma_scan_lanes puts consecutive ma_sum_fused scans on two
 * streams of the context's device in turn — the context's own and one it owns — and starts each when the scan in front of it has
 * begun to drain (ma_sum_fused_stamped_early's early stamp), not beside its whole length: what MA_GROUP_SCAN_LANES does per
 * member of a group, for a host without one (figures: profiles/r05_scan_lanes_api.jsonl).
 *   ma_scan_lanes_sum_fused  enqueue-only whatever mode the context is in; arguments and results as ma_sum_fused. The scans in
 *                            flight must not share `out` records (two lanes may finish in either order); the columns are only read.
 *                            Work the host enqueued on `ctx` BEFORE the call is seen (a per-context call counter) and the scan is
 *                            ordered behind all of it.
 *   ma_scan_lanes_join       ctx's stream behind everything the second lane has been given (enqueue-only): call it before
 *                            anything else on `ctx` that reads the scans' results or overwrites their columns.
 *   ma_scan_lanes_synchronize  waits for both lanes; a latched device condition of either comes back as its status.
 *   ma_scan_lanes_synchronize_for  the same under a deadline (below): the form for a host's loop.
 * A pipeline and its context are driven by ONE host thread at a time (entries another thread makes on the context while a scan
 * is being enqueued are ordered behind by the NEXT scan on the second lane, not by this one).
 * MA_ERR_UNSUPPORTED from _create on a runtime without stream memory operations (hipStreamWaitValue64). One pipeline per context
 * at a time; destroy it before the context. NOT under a profiler that collects hardware counters (rocprofv3 --pmc): it lets one
 * kernel run at a time whatever its stream, the wait on a device word is itself a kernel that polls, and a scan gated on the other
 * stream's early stamp can be let in ahead of the scan that stores it — the same holds for ma_ctx_wait_value, for the stamp
 * hand-off of the overlapped exchanges and for MA_GROUP_SCAN_LANES (their waits at least are bounded: ma_scan_lanes_synchronize_for, ma_group_synchronize_for).
 * Kernel tracing (--kernel-trace) does not serialise and is fine (profiles/r05_lanes_kernel_trace.txt). */
typedef struct ma_scan_lanes ma_scan_lanes;
ma_status ma_scan_lanes_create(ma_ctx* ctx, ma_scan_lanes** out_lanes);
ma_status ma_scan_lanes_sum_fused(ma_scan_lanes* lanes, size_t n_cols, const ma_fused_column* cols);
/* ONE column of any numeric type per scan (format_code as in Arrow: c C s S i I l L f g), through the single-column kernels of
 * ma_<t>_sum / ma_<t>_sum_dd: arguments as theirs. out_sum receives the wrapping 64-bit sum (integer formats) or the f64 sum
 * within 1 ULP (float formats; with out_lo != NULL the double-double pair, as ma_<t>_sum_dd); out_lo and out_valid_count may be
 * NULL. Every output must be device-reachable (device or ma_alloc64_pinned memory) and the column resident: the call only
 * enqueues. */
ma_status ma_scan_lanes_sum(ma_scan_lanes* lanes, int32_t format_code, const void* data, size_t n, const uint8_t* mask_bits,
                            size_t mask_bit_offset, int64_t null_count, void* out_sum, double* out_lo, uint64_t* out_valid_count);
ma_status ma_scan_lanes_join(ma_scan_lanes* lanes);
ma_status ma_scan_lanes_synchronize(ma_scan_lanes* lanes);
/* ma_scan_lanes_synchronize with a deadline — what a host that loops like benches/hotloop_benchmark_avg_std.rs:48-62 should call:
 * the pipeline's gates are waits across streams, and a wait nobody ends holds a stream for good. Polls both lanes' streams for
 * at most timeout_ms (<= 0: no deadline, plain ma_scan_lanes_synchronize). Past the deadline it stores all-ones into every
 * stamp word of the pipeline through a stream of its own in the low priority class (made at creation: its hardware queue is
 * never behind a held ordinary stream), so that every gate opens and both streams — lane 0 is the CALLER's context — run empty,
 * waits up to 2 s for that, marks the pipeline broken and returns MA_ERR_DEVICE; the error string names each lane that was
 * still pending, how many of its scans had finished, and the early stamp its gate was waiting for (have / want). The results of
 * the scans that were in flight are undefined. A broken pipeline takes no more scans (MA_ERR_DEVICE): destroy it — the context
 * takes a new one. ma_scan_lanes_is_broken: 0 healthy, 1 broken with both streams empty, 2 broken with a stream still busy
 * (the device may need a reset; ma_scan_lanes_destroy then leaves what that stream may still use to the process). */
ma_status ma_scan_lanes_synchronize_for(ma_scan_lanes* lanes, double timeout_ms);
int32_t ma_scan_lanes_is_broken(ma_scan_lanes* lanes);
uint64_t ma_scan_lanes_scans(ma_scan_lanes* lanes);
/* Bounded like ma_group_destroy: MINARROW_HIP_DESTROY_WAIT_MS (10 s) for the scans in flight, then the release above. */
void ma_scan_lanes_destroy(ma_scan_lanes* lanes);

/* Fold of per-rank (or per-chunk) reduction records after their exchange — the `.sum()` over per-chunk partials of
 * rayon_simd_sum_* (benches/benchmark_parallel_simd.rs:87) for a row-chunk partition over GPUs. record r =
 * stride_words x u64 (>= 5 used): [0] integer sum, [1] integer valid count, [2] f64 hi bits, [3] f64 lo bits (the
 * ma_f64_sum_dd pair), [4] float valid count. Folded strictly in record order (wrapping adds; error-free two-sum for
 * the pairs), so every rank obtains bit-identical finals and the f64 total stays within 1 ULP of the exactly rounded
 * sum. out4 = [integer sum, integer count, f64 sum bits, float count]. Buffers: device-reachable for an async
 * (enqueue-only) call, or host memory in sync mode. */
ma_status ma_fold_sum_records(ma_ctx* ctx, const uint64_t* records, size_t n_records, size_t stride_words,
                              uint64_t* out4);

/* ------------------------------------------------------------------------------------------------
 * Elementwise arithmetic — same names, argument order and meaning as the reference's L3 functions:
 *   apply_int_{i32,u32,i64,u64}(lhs, rhs, op, mask) -> Result<IntegerArray<T>, KernelError>
 *                                            src/kernels/arithmetic/dispatch.rs:65-133, :376-379
 *   apply_float_{f32,f64}(lhs, rhs, op, mask)  src/kernels/arithmetic/dispatch.rs:138-206, :389-402
 *   apply_fma_{f32,f64}(lhs, rhs, acc, mask)   src/kernels/arithmetic/dispatch.rs:211-290, :404-418
 * whose bodies are int_{dense,masked}_body_{std,simd}, float_{dense,masked}_body_*, fma_*_body_*
 * (src/kernels/arithmetic/std.rs:41-230, src/kernels/arithmetic/simd.rs:52-751).
 *
 *  - `op` is an ArithmeticOperator code (MA_OP_*).
 *  - lhs_len != rhs_len (or acc_len)  -> MA_ERR_LENGTH_MISMATCH            (confirm_equal_len, dispatch.rs:81)
 *  - mask_bits == NULL  <=> `mask: None`: dense body, out_mask_bits is ignored (may be NULL).
 *    mask_bits != NULL  <=> `Some(&mask)`: bit (mask_bit_offset + i) gates row i; null rows store 0
 *    (simd.rs:315, std.rs:132); out_mask_bits receives the result validity, bits >= len zero.
 *    The reference passes a whole Bitmask (offset 0); mask_bit_offset lets an Arrow slice be used as is.
 *  - Integers: Add/Sub/Mul wrap; Power = repeated wrapping multiply with exponent `rhs.to_u32().unwrap_or(0)`;
 *    FloorDiv rounds toward -inf; MIN / -1 wraps (the reference's SIMD lanes do, its scalar tail panics).
 *    Dense Div/Rem/FloorDiv with a zero divisor -> MA_ERR_DIVIDE_BY_ZERO (the reference panics, std.rs:53-77).
 *    Masked Div/Rem/FloorDiv with a zero divisor -> value 0 and validity bit cleared (simd.rs:170-181, 319-326).
 *  - Floats: IEEE 754 (x/0 = +-Inf or NaN, never an error); Remainder = fmod; Power = exp(rhs * ln(lhs))
 *    (std.rs:153 — two libm calls, so last-ulp differences to a given CPU libm are expected);
 *    FloorDiv = floor(lhs / rhs). FMA is fused (mul_add).
 *  - `*_scalar_rhs` / `*_scalar_lhs` are the fused form of a length-1 operand: the reference materialises
 *    vec64![x; n] and runs the two-array kernel (src/kernels/routing/broadcast.rs:25-112,
 *    src/kernels/broadcast/array.rs:139-184); results are identical, 8 B/row less traffic.
 *  - Inputs may alias each other but not `out`.
 * ---------------------------------------------------------------------------------------------- */

ma_status ma_apply_int_i32(ma_ctx* ctx, const int32_t* lhs, size_t lhs_len, const int32_t* rhs, size_t rhs_len, int32_t op,
                          const uint8_t* mask_bits, size_t mask_bit_offset, int32_t* out, uint8_t* out_mask_bits);
ma_status ma_apply_int_i32_scalar_rhs(ma_ctx* ctx, const int32_t* lhs, size_t lhs_len, int32_t scalar, int32_t op,
                                     const uint8_t* mask_bits, size_t mask_bit_offset, int32_t* out,
                                     uint8_t* out_mask_bits);
ma_status ma_apply_int_i32_scalar_lhs(ma_ctx* ctx, int32_t scalar, const int32_t* rhs, size_t rhs_len, int32_t op,
                                     const uint8_t* mask_bits, size_t mask_bit_offset, int32_t* out,
                                     uint8_t* out_mask_bits);

ma_status ma_apply_int_u32(ma_ctx* ctx, const uint32_t* lhs, size_t lhs_len, const uint32_t* rhs, size_t rhs_len, int32_t op,
                          const uint8_t* mask_bits, size_t mask_bit_offset, uint32_t* out, uint8_t* out_mask_bits);
ma_status ma_apply_int_u32_scalar_rhs(ma_ctx* ctx, const uint32_t* lhs, size_t lhs_len, uint32_t scalar, int32_t op,
                                     const uint8_t* mask_bits, size_t mask_bit_offset, uint32_t* out,
                                     uint8_t* out_mask_bits);
ma_status ma_apply_int_u32_scalar_lhs(ma_ctx* ctx, uint32_t scalar, const uint32_t* rhs, size_t rhs_len, int32_t op,
                                     const uint8_t* mask_bits, size_t mask_bit_offset, uint32_t* out,
                                     uint8_t* out_mask_bits);

ma_status ma_apply_int_i64(ma_ctx* ctx, const int64_t* lhs, size_t lhs_len, const int64_t* rhs, size_t rhs_len, int32_t op,
                          const uint8_t* mask_bits, size_t mask_bit_offset, int64_t* out, uint8_t* out_mask_bits);
ma_status ma_apply_int_i64_scalar_rhs(ma_ctx* ctx, const int64_t* lhs, size_t lhs_len, int64_t scalar, int32_t op,
                                     const uint8_t* mask_bits, size_t mask_bit_offset, int64_t* out,
                                     uint8_t* out_mask_bits);
ma_status ma_apply_int_i64_scalar_lhs(ma_ctx* ctx, int64_t scalar, const int64_t* rhs, size_t rhs_len, int32_t op,
                                     const uint8_t* mask_bits, size_t mask_bit_offset, int64_t* out,
                                     uint8_t* out_mask_bits);

ma_status ma_apply_int_u64(ma_ctx* ctx, const uint64_t* lhs, size_t lhs_len, const uint64_t* rhs, size_t rhs_len, int32_t op,
                          const uint8_t* mask_bits, size_t mask_bit_offset, uint64_t* out, uint8_t* out_mask_bits);
ma_status ma_apply_int_u64_scalar_rhs(ma_ctx* ctx, const uint64_t* lhs, size_t lhs_len, uint64_t scalar, int32_t op,
                                     const uint8_t* mask_bits, size_t mask_bit_offset, uint64_t* out,
                                     uint8_t* out_mask_bits);
ma_status ma_apply_int_u64_scalar_lhs(ma_ctx* ctx, uint64_t scalar, const uint64_t* rhs, size_t rhs_len, int32_t op,
                                     const uint8_t* mask_bits, size_t mask_bit_offset, uint64_t* out,
                                     uint8_t* out_mask_bits);

ma_status ma_apply_float_f32(ma_ctx* ctx, const float* lhs, size_t lhs_len, const float* rhs, size_t rhs_len, int32_t op,
                          const uint8_t* mask_bits, size_t mask_bit_offset, float* out, uint8_t* out_mask_bits);
ma_status ma_apply_float_f32_scalar_rhs(ma_ctx* ctx, const float* lhs, size_t lhs_len, float scalar, int32_t op,
                                     const uint8_t* mask_bits, size_t mask_bit_offset, float* out,
                                     uint8_t* out_mask_bits);
ma_status ma_apply_float_f32_scalar_lhs(ma_ctx* ctx, float scalar, const float* rhs, size_t rhs_len, int32_t op,
                                     const uint8_t* mask_bits, size_t mask_bit_offset, float* out,
                                     uint8_t* out_mask_bits);

ma_status ma_apply_float_f64(ma_ctx* ctx, const double* lhs, size_t lhs_len, const double* rhs, size_t rhs_len, int32_t op,
                          const uint8_t* mask_bits, size_t mask_bit_offset, double* out, uint8_t* out_mask_bits);
ma_status ma_apply_float_f64_scalar_rhs(ma_ctx* ctx, const double* lhs, size_t lhs_len, double scalar, int32_t op,
                                     const uint8_t* mask_bits, size_t mask_bit_offset, double* out,
                                     uint8_t* out_mask_bits);
ma_status ma_apply_float_f64_scalar_lhs(ma_ctx* ctx, double scalar, const double* rhs, size_t rhs_len, int32_t op,
                                     const uint8_t* mask_bits, size_t mask_bit_offset, double* out,
                                     uint8_t* out_mask_bits);

ma_status ma_apply_fma_f32(ma_ctx* ctx, const float* lhs, size_t lhs_len, const float* rhs, size_t rhs_len,
                        const float* acc, size_t acc_len, const uint8_t* mask_bits, size_t mask_bit_offset,
                        float* out, uint8_t* out_mask_bits);

ma_status ma_apply_fma_f64(ma_ctx* ctx, const double* lhs, size_t lhs_len, const double* rhs, size_t rhs_len,
                        const double* acc, size_t acc_len, const uint8_t* mask_bits, size_t mask_bit_offset,
                        double* out, uint8_t* out_mask_bits);

/* ------------------------------------------------------------------------------------------------
 * Bitmask kernels — same names as the reference's dispatch layer (src/kernels/bitmask/dispatch.rs:47-295),
 * bodies in src/kernels/bitmask/{simd,std}.rs. A window is (bits, offset_bits, len_bits) = BitmaskVT
 * (src/aliases.rs:172). Outputs are new bitmaps of `len` bits starting at bit 0, written as whole u64 words
 * (8*ceil(len/64) bytes, 8-byte aligned), bits >= len zero (clear_trailing_bits, bitmask/mod.rs:141-150).
 *
 * The reference addresses windows at different granularities; each entry point reproduces its function's rule
 * so results are identical for every offset:
 *   and/or/xor/not        window starts at BYTE offset/8 (bitmask_window_bytes, bitmask/mod.rs:124-128;
 *                         simd.rs:95-203) — a sub-byte offset is not shifted out
 *   popcount_mask         counting starts at WORD offset/64 (simd.rs:603-611)
 *   in_mask / not_in_mask rhs is scanned from WORD rhs_offset/64 (simd.rs:345); the result is all-true, a
 *                         bit-exact copy of the lhs window (slice_clone), its byte-granular NOT, or all-false
 *   eq_mask / ne_mask     offsets must be multiples of 64, else MA_ERR_INVALID_ARGUMENT (the reference panics,
 *                         simd.rs:411-416)
 *   all_eq / all_ne       len < 64: the words at offset/64 are compared under a len-bit mask (simd.rs:523-528);
 *                         otherwise offsets must be multiples of 64 (panic in the reference, :530-535).
 *                         all_ne = !all_eq ("not all equal", simd.rs:490-494)
 *   all_true/all_false    every logical bit set / clear (std.rs:300-366). The SIMD twin's false negative
 *                         for len % 64 != 0 && n_words % LANES == 0 (simd.rs:668-674) depends on a CPU
 *                         build constant and is not reproduced; it only ever selected a slower path.
 *   merge_bitmasks_to_new per-row AND of two optional bitmaps at bit 0 (bitmask/mod.rs:171-196);
 *                         *out_is_some = 0 and nothing is written when both are NULL (=> None)
 *   simd_eq_mask_u*       bit j = ((data[j] & field_mask) == target) (simd.rs:741-788)
 *   bitmask_slice         Bitmask::slice_clone: bit-exact copy of a window to bit 0
 * ---------------------------------------------------------------------------------------------- */

ma_status ma_bitmask_binop(ma_ctx* ctx, int32_t logical_op, const uint8_t* lhs_bits, size_t lhs_offset,
                           const uint8_t* rhs_bits, size_t rhs_offset, size_t len, uint8_t* out_bits);
ma_status ma_and_masks(ma_ctx* ctx, const uint8_t* lhs_bits, size_t lhs_offset, const uint8_t* rhs_bits,
                       size_t rhs_offset, size_t len, uint8_t* out_bits);
ma_status ma_or_masks(ma_ctx* ctx, const uint8_t* lhs_bits, size_t lhs_offset, const uint8_t* rhs_bits,
                      size_t rhs_offset, size_t len, uint8_t* out_bits);
ma_status ma_xor_masks(ma_ctx* ctx, const uint8_t* lhs_bits, size_t lhs_offset, const uint8_t* rhs_bits,
                       size_t rhs_offset, size_t len, uint8_t* out_bits);
ma_status ma_not_mask(ma_ctx* ctx, const uint8_t* src_bits, size_t offset, size_t len, uint8_t* out_bits);
ma_status ma_bitmask_slice(ma_ctx* ctx, const uint8_t* src_bits, size_t offset, size_t len, uint8_t* out_bits);
ma_status ma_in_mask(ma_ctx* ctx, const uint8_t* lhs_bits, size_t lhs_offset, const uint8_t* rhs_bits,
                     size_t rhs_offset, size_t len, uint8_t* out_bits);
ma_status ma_not_in_mask(ma_ctx* ctx, const uint8_t* lhs_bits, size_t lhs_offset, const uint8_t* rhs_bits,
                         size_t rhs_offset, size_t len, uint8_t* out_bits);
ma_status ma_eq_mask(ma_ctx* ctx, const uint8_t* a_bits, size_t a_offset, const uint8_t* b_bits, size_t b_offset,
                     size_t len, uint8_t* out_bits);
ma_status ma_ne_mask(ma_ctx* ctx, const uint8_t* a_bits, size_t a_offset, const uint8_t* b_bits, size_t b_offset,
                     size_t len, uint8_t* out_bits);
ma_status ma_all_eq(ma_ctx* ctx, const uint8_t* a_bits, size_t a_offset, const uint8_t* b_bits, size_t b_offset,
                    size_t len, int32_t* out_bool);
ma_status ma_all_ne(ma_ctx* ctx, const uint8_t* a_bits, size_t a_offset, const uint8_t* b_bits, size_t b_offset,
                    size_t len, int32_t* out_bool);
ma_status ma_popcount_mask(ma_ctx* ctx, const uint8_t* bits, size_t offset, size_t len, uint64_t* out_count);
ma_status ma_all_true_mask(ma_ctx* ctx, const uint8_t* bits, size_t len, int32_t* out_bool);
ma_status ma_all_false_mask(ma_ctx* ctx, const uint8_t* bits, size_t len, int32_t* out_bool);
ma_status ma_merge_bitmasks_to_new(ma_ctx* ctx, const uint8_t* lhs_bits, const uint8_t* rhs_bits, size_t len,
                                   uint8_t* out_bits, int32_t* out_is_some);
ma_status ma_simd_eq_mask_u8(ma_ctx* ctx, const uint8_t* data, size_t n, uint8_t field_mask, uint8_t target,
                             uint8_t* out_bits);
ma_status ma_simd_eq_mask_u16(ma_ctx* ctx, const uint16_t* data, size_t n, uint16_t field_mask, uint16_t target,
                              uint8_t* out_bits);
ma_status ma_simd_eq_mask_u32(ma_ctx* ctx, const uint32_t* data, size_t n, uint32_t field_mask, uint32_t target,
                              uint8_t* out_bits);
ma_status ma_simd_eq_mask_u64(ma_ctx* ctx, const uint64_t* data, size_t n, uint64_t field_mask, uint64_t target,
                              uint8_t* out_bits);

/* ------------------------------------------------------------------------------------------------
 * hipGraph capture — launch-bound loops (many small columns, the shape of the reference's 1000-row hot-loop
 * benches, benches/hotloop_benchmark_avg_simd.rs:205-208) pay one graph launch instead of one kernel launch per call.
 * Between ma_ctx_capture_begin and ma_ctx_capture_end every call on this context is RECORDED instead of executed
 * (the context is in async mode for the duration): reductions / means into device-reachable outputs, ma_apply_*,
 * ma_apply_fma_*, ma_apply_promote_*, ma_apply_datetime_*, the bitmask word ops (and/or/xor/not/slice/eq/ne/in with
 * a non-constant rhs), ma_simd_eq_mask_*, ma_dev_memset and the synth generators. Every buffer must be
 * device-reachable (device or pinned); calls that have to synchronise with the host — scans returning a value
 * (popcount, all_*), staging of pageable memory, consolidation, the Arrow export / stream entry points, allocation,
 * timers — fail with MA_ERR_INVALID_ARGUMENT and leave the capture intact. ma_graph_launch replays the recorded work on the context's
 * stream against the SAME addresses (refill the buffers, replay); in sync mode it waits and reports a dense
 * integer division by zero like the eager call, in async mode that is reported by the next ma_ctx_synchronize.
 * ---------------------------------------------------------------------------------------------- */
typedef struct ma_graph ma_graph;
ma_status ma_ctx_capture_begin(ma_ctx* ctx);
ma_status ma_ctx_capture_end(ma_ctx* ctx, ma_graph** out_graph);
ma_status ma_graph_launch(ma_ctx* ctx, ma_graph* graph);
ma_status ma_graph_node_count(const ma_graph* graph, size_t* out_nodes);
void ma_graph_destroy(ma_graph* graph);

/* ------------------------------------------------------------------------------------------------
 * Arrow C Data Interface — the reference's only pre-existing C surface (src/ffi/arrow_c_ffi.rs:
 * #[repr(C)] ArrowArray / ArrowSchema; read from C by tests/c_inspect_arrow.c:17-41, 56-171).
 * Numeric primitive arrays only: buffers[0] = validity bitmap or NULL, buffers[1] = values; format
 * "i" i32, "I" u32, "l" i64, "L" u64, "f" f32, "g" f64 (tests/arrow_c_integration.rs:62-80).
 * `offset` is honoured for both buffers (values: element offset, validity: bit offset); the reference always
 * exports 0 (arrow_c_ffi.rs:1773) and ignores it on import (arrow_c_ffi.rs:1098-1111). `null_count` = 0 selects
 * the dense kernel, -1 means unknown. `release` is never called: the producer keeps ownership.
 * ---------------------------------------------------------------------------------------------- */
#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE
struct ArrowSchema {
    const char* format;
    const char* name;
    const char* metadata;
    int64_t flags;
    int64_t n_children;
    struct ArrowSchema** children;
    struct ArrowSchema* dictionary;
    void (*release)(struct ArrowSchema*);
    void* private_data;
};
struct ArrowArray {
    int64_t length;
    int64_t null_count;
    int64_t offset;
    int64_t n_buffers;
    int64_t n_children;
    const void** buffers;
    struct ArrowArray** children;
    struct ArrowArray* dictionary;
    void (*release)(struct ArrowArray*);
    void* private_data;
};
#endif

/* Sum of a primitive array. Integer formats: *out_sum_i64 = wrapping 64-bit sum (bit pattern) and
 * *out_sum_f64 = that value converted; float formats: *out_sum_f64 only. Any output may be NULL. */
ma_status ma_sum_arrow(ma_ctx* ctx, const struct ArrowArray* array, const struct ArrowSchema* schema,
                       double* out_sum_f64, int64_t* out_sum_i64, uint64_t* out_valid_count);
ma_status ma_mean_arrow(ma_ctx* ctx, const struct ArrowArray* array, const struct ArrowSchema* schema, double* out_mean,
                        uint64_t* out_valid_count);
/* lhs (op) rhs for two primitive arrays, routed like resolve_binary_arithmetic
 * (src/kernels/routing/arithmetic.rs:214-222): equal lengths, or one side of length 1, which is broadcast
 * (src/kernels/routing/broadcast.rs:87-112 — fused, not materialised). Type matrix of arithmetic_dispatch
 * (routing/arithmetic.rs:278-406): same-format pairs; "i" with "g" or "f" in either order is promoted to the
 * float type (result format = the float side's); anything else -> MA_ERR_UNSUPPORTED (:403-405).
 * `out_values` receives max(len) elements of the result type.
 * Validity: when neither operand carries nulls the dense kernel runs and *out_has_validity = 0; otherwise rows
 * are gated by the AND of the attached bitmaps (merge_bitmasks_to_new, src/kernels/bitmask/mod.rs:171-196),
 * `out_validity` (8*ceil(len/64) bytes) receives the result validity and *out_has_validity = 1. */
ma_status ma_apply_arrow(ma_ctx* ctx, int32_t op, const struct ArrowArray* lhs, const struct ArrowSchema* lhs_schema,
                         const struct ArrowArray* rhs, const struct ArrowSchema* rhs_schema, void* out_values,
                         uint8_t* out_validity, int32_t* out_has_validity);

/* Results as Arrow C Data — the producer side (create_arrow_export, src/ffi/arrow_c_ffi.rs:1742-1821).
 * Same routing as ma_apply_arrow, but the library allocates the result: values and validity live in pinned host
 * memory the kernels wrote directly (64-byte aligned as check_alignment asserts, arrow_c_ffi.rs:1722-1738) and are
 * returned as an owned ArrowArray / ArrowSchema pair with the field values create_arrow_export writes: offset 0
 * (:1773), n_buffers 2, buffers[0] = NULL and null_count 0 when no operand carried nulls, null_count -1 (unknown)
 * otherwise (:1750), format = the routed result type, flags = ARROW_FLAG_NULLABLE (2, what the reference's import
 * tests, :2631) when a validity buffer is attached. The consumer owns both structs and must call their `release`
 * (sets release = NULL; the pinned memory — one allocation per call, shared by all columns of a batch — goes when the
 * last array that points into it is released). `name` NULL -> the left operand's field name. On failure nothing is
 * allocated and both `release` are NULL. */
ma_status ma_apply_arrow_export(ma_ctx* ctx, int32_t op, const struct ArrowArray* lhs, const struct ArrowSchema* lhs_schema,
                                const struct ArrowArray* rhs, const struct ArrowSchema* rhs_schema, const char* name,
                                struct ArrowArray* out_array, struct ArrowSchema* out_schema);
/* Table (op) Table over record batches (struct arrays, format "+s", one child per column — what the reference's
 * record-batch stream yields, arrow_c_ffi.rs:1823-1834) = broadcast_table_with_operator
 * (src/kernels/broadcast/table.rs:31-63): column counts must match ("Table column count mismatch", ->
 * MA_ERR_LENGTH_MISMATCH), result column i = lhs.cols[i] (op) rhs.cols[i] routed as ma_apply_arrow_export under
 * the LEFT table's field name (:55-57); the result struct carries the left table's name (:62). A struct-level
 * `offset` is applied to the children; struct-level validity is MA_ERR_UNSUPPORTED. The first failing column's
 * status is returned and everything already produced is freed. */
ma_status ma_apply_arrow_batch_export(ma_ctx* ctx, int32_t op, const struct ArrowArray* lhs_batch,
                                      const struct ArrowSchema* lhs_schema, const struct ArrowArray* rhs_batch,
                                      const struct ArrowSchema* rhs_schema, struct ArrowArray* out_batch,
                                      struct ArrowSchema* out_schema);

/* ------------------------------------------------------------------------------------------------
 * Consolidation of a chunked numeric column (BASELINE config 5) —
 *   Consolidate::consolidate / consolidate_concat / consolidate_arena  src/structs/chunked/super_table.rs:657-743
 *   consolidate_{int,float}_variant!, extend_null_mask                 src/traits/consolidate.rs:80-207
 *   Arena::write_slices                                                src/structs/arena.rs:264-308
 * chunk i = (chunk_data[i], chunk_lens[i] rows of elem_size bytes, optional validity chunk_masks[i] whose row 0
 * is bit chunk_mask_offsets[i]; chunk_masks / chunk_mask_offsets themselves may be NULL). Values are concatenated
 * in chunk order into out_data. The result has validity iff at least one chunk has (*out_has_mask); chunks
 * without a bitmap contribute all-valid rows (consolidate.rs:80-105). n_chunks == 0 -> MA_ERR_INVALID_ARGUMENT
 * (the reference panics: "consolidate() called on empty SuperTable", super_table.rs:693-696).
 * ---------------------------------------------------------------------------------------------- */
ma_status ma_consolidate_column(ma_ctx* ctx, size_t elem_size, size_t n_chunks, const void* const* chunk_data,
                                const size_t* chunk_lens, const uint8_t* const* chunk_masks,
                                const size_t* chunk_mask_offsets, void* out_data, uint8_t* out_mask,
                                int32_t* out_has_mask);

/* Whole-table consolidation into ONE arena — consolidate_tables_arena for the numeric columns of a SuperTable
 *   consolidate_arena -> consolidate_tables_arena   src/structs/chunked/super_table.rs:727-743, src/structs/arena.rs:1187-1340
 *   Arena::reserve_slice / align_cursor / write_slices   src/structs/arena.rs:152-232, 264-308
 *   Arena::capacity_for_regions                          src/structs/arena.rs:442-447
 * Layout (ma_arena_layout; pure host arithmetic, no device needed): per column, in column order, a data region of
 * n_rows * elem_sizes[c] bytes at the next 64-byte boundary, then — iff has_nulls[c] — a validity region of
 * ceil(n_rows / 8) bytes at the next 64-byte boundary. out_mask_offsets[c] = SIZE_MAX for a column without nulls.
 * *out_capacity_bytes = sum of the 64-byte-rounded regions (what the reference allocates), *out_used_bytes = end of the
 * last region (what Arena::freeze keeps). Any output pointer may be NULL.
 *
 * ma_consolidate_table_arena joins batch b = 0..n_batches-1 of every column c into that layout inside `arena`
 * (device, pinned or pageable memory of at least the capacity, 64-byte aligned): cell (c, b) is
 * cell_data[c * n_batches + b] with batch_rows[b] rows and optional validity cell_masks[c * n_batches + b] whose row 0
 * is bit cell_mask_offsets[c * n_batches + b] (both tables may be NULL). A column has nulls iff any of its cells has a
 * mask (arena.rs:1207-1209); cells without one contribute all-valid bits (arena.rs:291-296). All columns of one element
 * width are copied by ONE launch, each nullable column's validity is assembled by one more, whatever n_batches is — the
 * 100-batch x 20-column shape of benches/consolidate.rs costs a handful of launches and one descriptor upload instead of
 * 2 000 memcpys. Region padding bytes are left untouched in a device-reachable arena and zero in a pageable one. */
ma_status ma_arena_layout(size_t n_cols, const size_t* elem_sizes, const int32_t* has_nulls, size_t n_rows,
                          size_t* out_data_offsets, size_t* out_mask_offsets, size_t* out_capacity_bytes,
                          size_t* out_used_bytes);
ma_status ma_consolidate_table_arena(ma_ctx* ctx, size_t n_cols, size_t n_batches, const size_t* elem_sizes,
                                     const size_t* batch_rows, const void* const* cell_data,
                                     const uint8_t* const* cell_masks, const size_t* cell_mask_offsets, void* arena,
                                     size_t arena_bytes, size_t* out_data_offsets, size_t* out_mask_offsets,
                                     size_t* out_used_bytes);

/* Bit-packed columns — BooleanArray data bits and stand-alone bitmaps:
 *   Bitmask::extend_from_bitmask_range / extend_from_slice   src/structs/bitmask.rs:520-592
 *   BooleanArray::append_range                                src/structs/variants/boolean.rs:627-653
 *   Arena::write_boolean_slices                               src/structs/arena.rs:391-430
 * chunk i contributes bits [chunk_bit_offsets[i], +chunk_lens[i]) of chunk_bits[i] (LSB-first, Arrow order); the
 * windows are joined at arbitrary bit positions into out_bits (8*ceil(total/64) bytes, 8-byte aligned, bits >=
 * total zero). Validity as ma_consolidate_column: present iff any chunk has one, absent ones = all valid.
 * The offset tables may be NULL (= 0). */
ma_status ma_consolidate_boolean_column(ma_ctx* ctx, size_t n_chunks, const uint8_t* const* chunk_bits,
                                        const size_t* chunk_bit_offsets, const size_t* chunk_lens,
                                        const uint8_t* const* chunk_masks, const size_t* chunk_mask_offsets,
                                        uint8_t* out_bits, uint8_t* out_mask, int32_t* out_has_mask);

/* Narrow integers — the reference's `extended_numeric_types` feature (src/kernels/arithmetic/dispatch.rs:380-387).
 * Same contract as ma_apply_int_i32 above. */
ma_status ma_apply_int_i8(ma_ctx* ctx, const int8_t* lhs, size_t lhs_len, const int8_t* rhs, size_t rhs_len, int32_t op,
                          const uint8_t* mask_bits, size_t mask_bit_offset, int8_t* out, uint8_t* out_mask_bits);
ma_status ma_apply_int_i8_scalar_rhs(ma_ctx* ctx, const int8_t* lhs, size_t lhs_len, int8_t scalar, int32_t op,
                                     const uint8_t* mask_bits, size_t mask_bit_offset, int8_t* out,
                                     uint8_t* out_mask_bits);
ma_status ma_apply_int_i8_scalar_lhs(ma_ctx* ctx, int8_t scalar, const int8_t* rhs, size_t rhs_len, int32_t op,
                                     const uint8_t* mask_bits, size_t mask_bit_offset, int8_t* out,
                                     uint8_t* out_mask_bits);
ma_status ma_apply_int_u8(ma_ctx* ctx, const uint8_t* lhs, size_t lhs_len, const uint8_t* rhs, size_t rhs_len, int32_t op,
                          const uint8_t* mask_bits, size_t mask_bit_offset, uint8_t* out, uint8_t* out_mask_bits);
ma_status ma_apply_int_u8_scalar_rhs(ma_ctx* ctx, const uint8_t* lhs, size_t lhs_len, uint8_t scalar, int32_t op,
                                     const uint8_t* mask_bits, size_t mask_bit_offset, uint8_t* out,
                                     uint8_t* out_mask_bits);
ma_status ma_apply_int_u8_scalar_lhs(ma_ctx* ctx, uint8_t scalar, const uint8_t* rhs, size_t rhs_len, int32_t op,
                                     const uint8_t* mask_bits, size_t mask_bit_offset, uint8_t* out,
                                     uint8_t* out_mask_bits);
ma_status ma_apply_int_i16(ma_ctx* ctx, const int16_t* lhs, size_t lhs_len, const int16_t* rhs, size_t rhs_len, int32_t op,
                          const uint8_t* mask_bits, size_t mask_bit_offset, int16_t* out, uint8_t* out_mask_bits);
ma_status ma_apply_int_i16_scalar_rhs(ma_ctx* ctx, const int16_t* lhs, size_t lhs_len, int16_t scalar, int32_t op,
                                     const uint8_t* mask_bits, size_t mask_bit_offset, int16_t* out,
                                     uint8_t* out_mask_bits);
ma_status ma_apply_int_i16_scalar_lhs(ma_ctx* ctx, int16_t scalar, const int16_t* rhs, size_t rhs_len, int32_t op,
                                     const uint8_t* mask_bits, size_t mask_bit_offset, int16_t* out,
                                     uint8_t* out_mask_bits);
ma_status ma_apply_int_u16(ma_ctx* ctx, const uint16_t* lhs, size_t lhs_len, const uint16_t* rhs, size_t rhs_len, int32_t op,
                          const uint8_t* mask_bits, size_t mask_bit_offset, uint16_t* out, uint8_t* out_mask_bits);
ma_status ma_apply_int_u16_scalar_rhs(ma_ctx* ctx, const uint16_t* lhs, size_t lhs_len, uint16_t scalar, int32_t op,
                                     const uint8_t* mask_bits, size_t mask_bit_offset, uint16_t* out,
                                     uint8_t* out_mask_bits);
ma_status ma_apply_int_u16_scalar_lhs(ma_ctx* ctx, uint16_t scalar, const uint16_t* rhs, size_t rhs_len, int32_t op,
                                     const uint8_t* mask_bits, size_t mask_bit_offset, uint16_t* out,
                                     uint8_t* out_mask_bits);

/* apply_datetime_{i32,u32,i64,u64}(lhs: DatetimeAVT, rhs: DatetimeAVT, op) — src/kernels/arithmetic/dispatch.rs:309-372,
 * :420-427: a DatetimeAVT is (array, offset, len). The data windows are data[offset .. offset+len]; the result
 * validity is merge_bitmasks_to_new(lhs mask, rhs mask, len) = per-row AND counted from bit 0 of each bitmap
 * (dispatch.rs:321-322 — the reference does not window the masks by the view offset), then the integer kernels run
 * unchanged. lhs_len != rhs_len -> MA_ERR_LENGTH_MISMATCH ("apply_datetime: length mismatch").
 * Both masks NULL -> dense kernel, *out_has_mask = 0, out_mask_bits untouched. */
ma_status ma_apply_datetime_i32(ma_ctx* ctx, const int32_t* lhs_data, size_t lhs_offset, size_t lhs_len,
                               const uint8_t* lhs_mask_bits, const int32_t* rhs_data, size_t rhs_offset, size_t rhs_len,
                               const uint8_t* rhs_mask_bits, int32_t op, int32_t* out, uint8_t* out_mask_bits,
                               int32_t* out_has_mask);
ma_status ma_apply_datetime_u32(ma_ctx* ctx, const uint32_t* lhs_data, size_t lhs_offset, size_t lhs_len,
                               const uint8_t* lhs_mask_bits, const uint32_t* rhs_data, size_t rhs_offset, size_t rhs_len,
                               const uint8_t* rhs_mask_bits, int32_t op, uint32_t* out, uint8_t* out_mask_bits,
                               int32_t* out_has_mask);
ma_status ma_apply_datetime_i64(ma_ctx* ctx, const int64_t* lhs_data, size_t lhs_offset, size_t lhs_len,
                               const uint8_t* lhs_mask_bits, const int64_t* rhs_data, size_t rhs_offset, size_t rhs_len,
                               const uint8_t* rhs_mask_bits, int32_t op, int64_t* out, uint8_t* out_mask_bits,
                               int32_t* out_has_mask);
ma_status ma_apply_datetime_u64(ma_ctx* ctx, const uint64_t* lhs_data, size_t lhs_offset, size_t lhs_len,
                               const uint8_t* lhs_mask_bits, const uint64_t* rhs_data, size_t rhs_offset, size_t rhs_len,
                               const uint8_t* rhs_mask_bits, int32_t op, uint64_t* out, uint8_t* out_mask_bits,
                               int32_t* out_has_mask);

/* ------------------------------------------------------------------------------------------------
 * Mixed-type arithmetic with the promotion fused into the kernel — arithmetic_dispatch's promote_to_float64! /
 * promote_to_float32! arms (src/kernels/routing/arithmetic.rs:244-269, 342-373): (Int32, Float64) and
 * (Float64, Int32) -> f64, (Int32, Float32) and (Float32, Int32) -> f32. The reference materialises two casted
 * Vec64s and calls apply_float_*; results here are bit-identical to that, without the two extra passes.
 * Same contract as ma_apply_float_* (length check, optional mask, IEEE semantics).
 * ---------------------------------------------------------------------------------------------- */
ma_status ma_apply_promote_i32_f64(ma_ctx* ctx, const int32_t* lhs, size_t lhs_len, const double* rhs, size_t rhs_len,
                                   int32_t op, const uint8_t* mask_bits, size_t mask_bit_offset, double* out,
                                   uint8_t* out_mask_bits);
ma_status ma_apply_promote_i32_f64_scalar_rhs(ma_ctx* ctx, const int32_t* lhs, size_t lhs_len, double scalar, int32_t op,
                                              const uint8_t* mask_bits, size_t mask_bit_offset, double* out,
                                              uint8_t* out_mask_bits);
ma_status ma_apply_promote_i32_f64_scalar_lhs(ma_ctx* ctx, int32_t scalar, const double* rhs, size_t rhs_len, int32_t op,
                                              const uint8_t* mask_bits, size_t mask_bit_offset, double* out,
                                              uint8_t* out_mask_bits);
ma_status ma_apply_promote_f64_i32(ma_ctx* ctx, const double* lhs, size_t lhs_len, const int32_t* rhs, size_t rhs_len,
                                   int32_t op, const uint8_t* mask_bits, size_t mask_bit_offset, double* out,
                                   uint8_t* out_mask_bits);
ma_status ma_apply_promote_f64_i32_scalar_rhs(ma_ctx* ctx, const double* lhs, size_t lhs_len, int32_t scalar, int32_t op,
                                              const uint8_t* mask_bits, size_t mask_bit_offset, double* out,
                                              uint8_t* out_mask_bits);
ma_status ma_apply_promote_f64_i32_scalar_lhs(ma_ctx* ctx, double scalar, const int32_t* rhs, size_t rhs_len, int32_t op,
                                              const uint8_t* mask_bits, size_t mask_bit_offset, double* out,
                                              uint8_t* out_mask_bits);
ma_status ma_apply_promote_i32_f32(ma_ctx* ctx, const int32_t* lhs, size_t lhs_len, const float* rhs, size_t rhs_len,
                                   int32_t op, const uint8_t* mask_bits, size_t mask_bit_offset, float* out,
                                   uint8_t* out_mask_bits);
ma_status ma_apply_promote_i32_f32_scalar_rhs(ma_ctx* ctx, const int32_t* lhs, size_t lhs_len, float scalar, int32_t op,
                                              const uint8_t* mask_bits, size_t mask_bit_offset, float* out,
                                              uint8_t* out_mask_bits);
ma_status ma_apply_promote_i32_f32_scalar_lhs(ma_ctx* ctx, int32_t scalar, const float* rhs, size_t rhs_len, int32_t op,
                                              const uint8_t* mask_bits, size_t mask_bit_offset, float* out,
                                              uint8_t* out_mask_bits);
ma_status ma_apply_promote_f32_i32(ma_ctx* ctx, const float* lhs, size_t lhs_len, const int32_t* rhs, size_t rhs_len,
                                   int32_t op, const uint8_t* mask_bits, size_t mask_bit_offset, float* out,
                                   uint8_t* out_mask_bits);
ma_status ma_apply_promote_f32_i32_scalar_rhs(ma_ctx* ctx, const float* lhs, size_t lhs_len, int32_t scalar, int32_t op,
                                              const uint8_t* mask_bits, size_t mask_bit_offset, float* out,
                                              uint8_t* out_mask_bits);
ma_status ma_apply_promote_f32_i32_scalar_lhs(ma_ctx* ctx, float scalar, const int32_t* rhs, size_t rhs_len, int32_t op,
                                              const uint8_t* mask_bits, size_t mask_bit_offset, float* out,
                                              uint8_t* out_mask_bits);

/* route_super_array_broadcast — src/kernels/broadcast/super_array.rs:180-251: SuperArray (op) SuperArray chunk by
 * chunk (the reference loops sequentially, "// TODO: Parallelise", :193; here the chunks are enqueued back to
 * back on the context's stream, and a host with one context per GPU gives each a subset of the chunks).
 * format_code: Arrow format character of the element type ('i','I','l','L','f','g').
 * chunk i: lhs_lens[i] != rhs_lens[i] -> MA_ERR_LENGTH_MISMATCH (:202-212). Common mask per chunk (:215-229):
 * none -> dense; one side -> that bitmap; both -> lhs.union(rhs) = bitwise OR (src/structs/bitmask.rs:661 — not the
 * AND of merge_bitmasks_to_new); `null_mask_override` replaces it for every chunk when non-NULL (:231).
 * lhs_masks / rhs_masks / out_masks (tables or entries) may be NULL; out_has_mask[i] reports whether
 * out_masks[i] was written. */
ma_status ma_route_super_array_broadcast(ma_ctx* ctx, int32_t format_code, int32_t op, size_t n_chunks,
                                         const void* const* lhs_data, const size_t* lhs_lens,
                                         const uint8_t* const* lhs_masks, const void* const* rhs_data,
                                         const size_t* rhs_lens, const uint8_t* const* rhs_masks,
                                         const uint8_t* null_mask_override, void* const* out_data,
                                         uint8_t* const* out_masks, int32_t* out_has_mask);

/* broadcast_superarray_to_scalar / broadcast_scalar_to_superarray — src/kernels/broadcast/super_array.rs:87-116 and
 * src/kernels/broadcast/scalar.rs:214-243 (their SuperArrayView twins: super_array.rs:120-148, scalar.rs:247-276; routed
 * from broadcast/mod.rs:232-251): every chunk (op) the scalar, or the scalar (op) every chunk when scalar_is_lhs != 0.
 * The reference maps broadcast_value over the chunks, one array kernel call with a length-1 operand each; here ALL chunks
 * go in one launch (segments for very long lists) with the scalar as a kernel argument. `scalar` points at ONE host
 * element of the type format_code names ('i','I','l','L','f','g'; cast mixed-type scalars first, as the C++ mirror does).
 * chunk_masks: the reference passes None for every chunk (array.rs:183: the chunks' own validity is not consulted, the
 * result chunks are dense) = NULL here. A non-NULL entry gates chunk i like the mask argument of ma_apply_* (bit 0 = row
 * 0): out_masks[i] receives it and out_has_mask[i] says so. Dense integer Div / Rem / FloorDiv by zero ->
 * MA_ERR_DIVIDE_BY_ZERO; masked ones clear the row's bit instead (simd.rs:319-326) — inside the same launch, unless a masked chunk's output starts off a
 * 16-byte boundary (then chunk by chunk). */
ma_status ma_broadcast_super_array_scalar(ma_ctx* ctx, int32_t format_code, int32_t op, int32_t scalar_is_lhs,
                                          const void* scalar, size_t n_chunks, const void* const* chunk_data,
                                          const size_t* chunk_lens, const uint8_t* const* chunk_masks,
                                          void* const* out_data, uint8_t* const* out_masks, int32_t* out_has_mask);

/* ------------------------------------------------------------------------------------------------
 * Arrow C Stream ingestion — the reference moves chunked tables (SuperTable) through ArrowArrayStream
 * (src/ffi/arrow_c_ffi.rs:160-184 struct, :2104-2260 export / import). Sum and valid count of one column over all
 * record batches of a stream; `column` indexes the children of "+s" (record batch) arrays, or is -1 / 0 for a
 * stream of primitive arrays. Each batch is uploaded at PCIe line rate, released, and its sum kernel runs while
 * the host pulls the next batch from the producer; batches under 4 MiB (a SuperTable rechunked at 8192 rows) are gathered
 * into pinned 8-MiB tiles first — one copy and one sum per tile (43-48 GB/s at 8192-row batches; 1.5-2.2 batch by batch). The stream is consumed to its end but NOT
 * released (the caller owns it). Outputs as ma_sum_arrow; *out_rows / *out_batches count what was read.
 * ---------------------------------------------------------------------------------------------- */
#ifndef ARROW_C_STREAM_INTERFACE
#define ARROW_C_STREAM_INTERFACE
struct ArrowArrayStream {
    int (*get_schema)(struct ArrowArrayStream*, struct ArrowSchema* out);
    int (*get_next)(struct ArrowArrayStream*, struct ArrowArray* out);
    const char* (*get_last_error)(struct ArrowArrayStream*);
    void (*release)(struct ArrowArrayStream*);
    void* private_data;
};
#endif
ma_status ma_sum_arrow_stream(ma_ctx* ctx, struct ArrowArrayStream* stream, int64_t column, double* out_sum_f64,
                              int64_t* out_sum_i64, uint64_t* out_valid_count, uint64_t* out_rows,
                              uint64_t* out_batches);

/* SuperTable (op) SuperTable as a streaming operator — broadcast_super_table_with_operator
 * (src/kernels/broadcast/super_table.rs:37-72) over the record-batch streams the reference moves SuperTables in
 * (arrow_c_ffi.rs:2104-2260). Both input streams are MOVED into the operator (their `release` is set to NULL) and
 * *out_stream becomes an ArrowArrayStream whose get_next pulls one batch from either side, computes every column pair
 * on the GPU (ma_apply_arrow_batch_export) and returns an owned struct array in pinned memory; get_schema gives the
 * routed result schema (left field names). Errors surface through the stream protocol (non-zero return +
 * get_last_error): "Table column count mismatch", "SuperTable chunk count mismatch" when one side ends first, a row
 * count or type-matrix failure in some batch. Releasing the operator releases both inputs. `ctx` must outlive it.
 * Batch pairs under 1 MiB per column (8192-row batches) are gathered into pinned tiles of up to 2^20 rows: a tile is ONE call
 * of the batch operator, its result batches are handed out one per get_next as slices (Arrow `offset`) of the tile's result
 * under one shared owner — same batch boundaries and values; an error inside a tile is replayed batch by batch, so it is
 * reported at its own batch after every batch in front of it (28 GB/s of input + output at 8192 rows; 1.1 batch by batch). */
ma_status ma_apply_arrow_stream_export(ma_ctx* ctx, int32_t op, struct ArrowArrayStream* lhs_stream,
                                       struct ArrowArrayStream* rhs_stream, struct ArrowArrayStream* out_stream);

/* ------------------------------------------------------------------------------------------------
 * Row-chunk reductions over several GPUs driven from ONE process — the reference's Rayon path
 * (`slice.par_chunks(1 << 20).map(simd_sum).sum()`, benches/benchmark_parallel_simd.rs:81-98) for a host such as the
 * Rust library itself. A group owns one context per listed device ordinal. Member i scans chunk i (resident on, or
 * reachable from, its device) concurrently with the others and writes {sum | hi, lo, count} into its 64-byte record of
 * the reduction's `column` (0 .. MA_GROUP_MAX_COLUMNS-1: several reductions — the columns of a SuperTable, or an i64
 * and an f64 column as in the reference's bench — share ONE exchange). Split rows with 64-row-aligned boundaries so
 * that a chunk's validity window starts on a word (any bit offset works, aligned ones are free).
 *
 * The exchange that replaces Rayon's `.sum()` of the partials:
 *   MA_GROUP_EXCHANGE_RCCL   ncclCommInitAll over the members' devices (which must be distinct); ma_group_exchange
 *                            enqueues ONE grouped ncclAllGather of the members' record blocks over xGMI on the members'
 *                            streams, then on every device the member-ordered fold (wrapping adds; error-free two-sum
 *                            for the (hi, lo) pairs): all GPUs end up with bit-identical finals, the f64 total within
 *                            1 ULP of the exactly rounded sum. This is an all-reduce; ncclAllReduce itself is not used
 *                            for the float pairs because its sum rounds at every hop.
 *   (default)                the kernels write their records into pinned host memory and the host folds the G x 64
 *                            bytes in member order after the streams drain: no collective is needed inside one process.
 *   MA_GROUP_EXCHANGE_OVERLAP  with MA_GROUP_EXCHANGE_RCCL: ma_group_exchange issues the all-gather + fold of the record set
 *                            just filled on a second stream per member, behind that member's scans, and the members'
 *                            streams go straight on with the next step's scans into a SECOND record set (two steps in
 *                            flight; a set is re-filled only behind its last exchange). For back-to-back steps whose scans
 *                            are short against the exchange — a 10^9-row column over 8 GPUs: 0.14 ms per scan. ma_group_result
 *                            reads the set of the most recent exchange: a record slot read after an exchange must have been
 *                            enqueued in THAT step (MA_ERR_INVALID_ARGUMENT otherwise — it would come back from the other
 *                            set, two steps old). (The multi-process twin: ma_comm_sum_exchange_overlapped.)
 *   MA_GROUP_SCAN_LANES      with MA_GROUP_EXCHANGE_RCCL | MA_GROUP_EXCHANGE_OVERLAP: every member gets a SECOND scan stream.
 *                            Record set 0 is filled by scans on the member's own stream, set 1 by scans on the second one, so
 *                            consecutive ma_group_enqueue_sum_table steps run on two streams — and each is gated on the EARLY
 *                            stamp of the step before it (stored when the workgroups of two of the eight XCDs have scanned
 *                            their rows — ma_sum_fused_stamped_early): its ramp runs under the previous step's stragglers and
 *                            hand-off instead of behind them (125 M rows per column and member WITH the exchange: 0.288-0.292
 *                            -> 0.275-0.278 ms per step, profiles/r05_share_lanes_ab.txt; 2^24 rows: -19 %; 10^9: no change —
 *                            profiles/r05_probe_early_stamp.jsonl), without the two scans running side by side for their
 *                            whole length. Ordering: work the host enqueues ITSELF on a member's context
 *                            is seen by the group (the next step on the second lane is ordered behind all of it), but such work
 *                            is ordered behind a step that runs on the second lane only after ma_group_join_lanes (enqueue-only)
 *                            or ma_group_synchronize. Ignored (ma_group_exchange_note says so) without the overlapped RCCL
 *                            exchange or where stamps are not plain device words.
 *   MA_GROUP_EXCHANGE_FALLBACK_HOST  with MA_GROUP_EXCHANGE_RCCL: use the host fold when RCCL cannot be initialised
 *                            (library missing, members sharing a device); ma_group_exchange_note() then says why.
 * ma_group_create() = ma_group_create_ex() with flags 0, or RCCL|FALLBACK_HOST when the environment variable
 * MINARROW_HIP_GROUP_EXCHANGE is "rccl". librccl.so.1 is opened on first use, never at library load.
 *
 * ma_group_enqueue_sum_* and ma_group_exchange only ENQUEUE (steps may be issued back to back without waiting; a
 * later step overwrites the records of an earlier one in stream order); ma_group_synchronize waits for every member
 * and makes ma_group_result valid: out_int_* from the *_i64 reduction of that column, out_f64_* from its *_f64 reduction
 * (any may be NULL). ma_group_member_result reads the finals GPU `member` holds (identical on all members).
 * ma_group_sum_i64 / _f64 are the synchronous one-call forms on column 0. The member contexts (ma_group_ctx: use them
 * to allocate and fill each device's chunk) stay in async mode for the life of the group.
 *
 * Who issues: every member has a persistent issue thread (the reference's Rayon pool issues from all cores,
 * benches/benchmark_parallel_simd.rs:83-87). A group call validates on the calling thread, the members' threads enqueue
 * their own launches concurrently (own device, own stream, own RCCL rank — no ncclGroup needed), and the call returns
 * when every member has enqueued: host time per call is ONE member's launches plus a hand-off, not the sum over the
 * members. MA_GROUP_ISSUE_CALLER (or MINARROW_HIP_GROUP_ISSUE=caller) keeps everything on the calling thread.
 *
 * Residency: chunk i is dereferenced by member i's kernels, so its data and validity must be resident on that member's
 * device (ma_pointer_device(ptr) == ma_ctx_hip_device(ma_group_ctx(group, i))) or be host memory; a pointer into
 * another GPU's memory is MA_ERR_INVALID_ARGUMENT before anything is enqueued. Peer capability between the members'
 * devices is probed (hipDeviceCanAccessPeer) and enabled once at creation — ma_group_peer_access, summarised in
 * ma_group_exchange_note — and ma_group_consolidate_column returns MA_ERR_UNSUPPORTED for an owner -> destination pair
 * without it.
 * ---------------------------------------------------------------------------------------------- */
typedef struct ma_group ma_group;
#define MA_GROUP_MAX_COLUMNS 16
enum {
    MA_GROUP_EXCHANGE_RCCL = 1, MA_GROUP_EXCHANGE_FALLBACK_HOST = 2, MA_GROUP_ISSUE_CALLER = 4, MA_GROUP_EXCHANGE_OVERLAP = 8,
    MA_GROUP_SCAN_LANES = 16
};
ma_status ma_group_create(const int32_t* device_ordinals, int32_t n_members, ma_group** out_group);
ma_status ma_group_create_ex(const int32_t* device_ordinals, int32_t n_members, uint32_t flags, ma_group** out_group);
/* Waits at most 10 s for what the members' streams still hold, then aborts the communicators (as ma_group_synchronize_for does) —
 * never an unbounded wait; what a stream that still has not run empty holds is left to the process. */
void ma_group_destroy(ma_group* group);
int32_t ma_group_size(ma_group* group);
ma_ctx* ma_group_ctx(ma_group* group, int32_t index);
/* 1 = RCCL all-gather + device fold, 0 = host fold. */
int32_t ma_group_exchange_kind(ma_group* group);
/* Why the exchange is not the one asked for (if so), the peer-access summary, who issues. Never NULL. */
const char* ma_group_exchange_note(ma_group* group);
/* 1 = one issue thread per member, 0 = the calling thread issues for every member. */
int32_t ma_group_issue_kind(ma_group* group);
/* 1 when member from_member's device can address member to_member's device memory (same device, or peer access probed
 * and enabled at creation), 0 when not, -1 for a bad argument. */
int32_t ma_group_peer_access(ma_group* group, int32_t from_member, int32_t to_member);
ma_status ma_group_enqueue_sum_i64(ma_group* group, int32_t column, const int64_t* const* chunk_data,
                                   const size_t* chunk_lens, const uint8_t* const* chunk_masks,
                                   const size_t* chunk_mask_offsets);
ma_status ma_group_enqueue_sum_f64(ma_group* group, int32_t column, const double* const* chunk_data,
                                   const size_t* chunk_lens, const uint8_t* const* chunk_masks,
                                   const size_t* chunk_mask_offsets);
/* The partitioned step in ONE launch per member: chunk i (on member i's device) of each of n_cols (1..MA_FUSED_MAX_COLUMNS)
 * long 8-byte columns is scanned by a single ma_sum_fused launch — the reference's bench runs its i64 and its f64 loop
 * back to back over the same partition (benches/benchmark_parallel_simd.rs:99-125). columns[k] = the record slot
 * (0 <= slot < MA_GROUP_MAX_COLUMNS) that column k's result goes to: formats 'l' / 'L' fill its integer half, 'g' its float
 * half (an integer and a float column may share a slot, as ma_group_enqueue_sum_i64 / _f64 on one slot do).
 * chunk_data[k][i], chunk_lens[k][i]: member i's chunk of column k; chunk_masks (or chunk_masks[k]) and chunk_mask_offsets
 * (or [k]) may be NULL. Then ma_group_exchange / ma_group_synchronize / ma_group_result as usual. */
ma_status ma_group_enqueue_sum_table(ma_group* group, int32_t n_cols, const int32_t* columns, const int32_t* format_codes,
                                     const void* const* const* chunk_data, const size_t* const* chunk_lens,
                                     const uint8_t* const* const* chunk_masks, const size_t* const* chunk_mask_offsets);
/* Where an exchange's time goes, for the first run on a multi-GPU node to explain itself: every 4th ma_group_exchange
 * carries three HIP events on member 0's exchange stream (in front of the all-gather, behind it, behind the fold kernel).
 * Averages in microseconds over the exchanges sampled since the last call (which waits for sampled exchanges still in
 * flight: call it after ma_group_synchronize), their number, and the rank count RCCL itself reports for member 0's
 * communicator (ncclCommCount; 0 with the host exchange, whose "fold" is the host loop inside ma_group_synchronize and
 * whose all-gather time is 0: the kernels write their records straight into pinned host memory). Any output may be NULL. */
ma_status ma_group_exchange_stats(ma_group* group, double* out_all_gather_us, double* out_fold_us, int32_t* out_samples,
                                  int32_t* out_rccl_ranks);
ma_status ma_group_exchange(ma_group* group);
ma_status ma_group_synchronize(ma_group* group);
/* Timing marks around the scans of a group step, for a host that wants every member's scan time inside a longer timed region:
 * ma_group_mark_next_scan makes the NEXT ma_group_enqueue_sum_table record marks from_index / to_index (ma_ctx_mark's index space)
 * on every member's scanning context right around its scan launch — behind whatever the launch waits for (with
 * MA_GROUP_SCAN_LANES: the early stamp of the step before), in front of the exchange; ma_group_mark_elapsed_ms waits for member's
 * mark to_index and returns the milliseconds between the two (with two scan lanes consecutive steps overlap by design: the sum of
 * the scans' times then exceeds the steps' wall time). */
ma_status ma_group_mark_next_scan(ma_group* group, int32_t from_index, int32_t to_index);
ma_status ma_group_mark_elapsed_ms(ma_group* group, int32_t member, int32_t from_index, int32_t to_index, float* out_ms);

/* ---- first contact with a multi-GPU node: bounded waits, a way down, a self-test -----------------------------------------
 * The reference's parallel reduction is a plain `main` over a Rayon pool (benches/benchmark_parallel_simd.rs:81-125): a
 * worker that fails ends the process with a message. A collective whose peer never arrives does not fail — the host blocks
 * in ma_group_synchronize for good. A host that must not (a service, a benchmark under somebody else's clock) uses:
 *
 * ma_group_synchronize_for  ma_group_synchronize with a deadline in milliseconds (<= 0: none). It polls the members'
 *     streams; when they have run empty it behaves exactly like ma_group_synchronize. Past the deadline it releases every
 *     value a stream of the group may be held behind, aborts every communicator (ncclCommAbort: the collective kernels in
 *     flight end), waits a bounded time for the streams to run empty, marks the group BROKEN and returns MA_ERR_DEVICE;
 *     ma_last_error_string() names the members and phases that were still pending ("member 3 (device 3): exchange stream
 *     (all-gather or fold in flight)"). A broken group refuses ma_group_exchange / _synchronize / _selftest with
 *     MA_ERR_DEVICE instead of blocking; its member contexts, and every column allocated from them, stay usable.
 * ma_group_is_broken        0 = healthy, 1 = broken and its streams have run empty (ma_group_rebuild_exchange will work),
 *     2 = broken and some stream is STILL busy (a kernel that does not end: the device may need a reset; the group can only
 *     be destroyed, which then leaks what it holds rather than wait).
 * ma_group_rebuild_exchange a fresh exchange for the same members, as ma_group_create_ex(flags) would set one up: other
 *     flags = one notch down (overlapped -> in-stream, issue threads -> MA_GROUP_ISSUE_CALLER's grouped calls, RCCL -> the
 *     host fold). Works on a healthy group too (drains it first). Results of earlier exchanges are gone.
 * ma_group_flags            the MA_GROUP_* flags that describe the exchange and issue form now in effect (after fall-backs).
 * ma_group_set_handoff      overlapped exchanges only: how a member's exchange stream learns that the step's records are
 *     complete. MA_GROUP_HANDOFF_STAMP (default): the fused scan's final thread stores a sequence number the exchange stream
 *     waits on (hipStreamWaitValue64), nothing but scans on the scan stream; steps not enqueued by ma_group_enqueue_sum_table
 *     use the event by themselves. MA_GROUP_HANDOFF_EVENT: always an event recorded on the scan stream (one more notch down
 *     that needs no rebuild). ma_group_handoff: which one a stamped step would get now (-1: the group does not overlap).
 * ma_group_selftest         proves the group's machinery with tagged data before a host trusts it with a job, every step under
 *     `timeout_ms` (> 0): (MA_SELFTEST_EXCHANGE) rank-tagged 64-byte records through ma_group_exchange in the group's own
 *     hand-off and issue form — with MA_SELFTEST_EXCHANGE_ALL_FORMS in every form the group can run (stamp / event hand-off x
 *     issue threads / calling thread) — checking on EVERY member that the gathered blocks are the members' records in rank
 *     order and that the finals are the member-ordered fold; (MA_SELFTEST_STAMPS) on every member a stamp stored by a kernel
 *     on its scan stream and waited on by its exchange stream; (MA_SELFTEST_PEER_COPIES) a 1-MiB hipMemcpyPeerAsync round trip
 *     i -> j -> i between every ordered pair of distinct, peer-capable devices. what = 0: those three. MA_OK only when
 *     everything tried passed; a step that ran into the deadline leaves the group broken (as ma_group_synchronize_for).
 *     out_report (may be NULL) says what was tried and what passed; `text` is the one-line summary a log wants. The group's
 *     records are zeroed afterwards; anything the host had in flight is waited for first (same deadline). */
typedef struct ma_selftest_report {
    uint32_t struct_bytes;        /* sizeof(ma_selftest_report) of the library that filled it */
    int32_t n_members, n_devices; /* members; distinct HIP devices among them */
    int32_t exchange_kind;        /* 1 = RCCL all-gather + device fold, 0 = host fold */
    int32_t rccl_ranks;           /* ncclCommCount of member 0's (or this rank's) communicator, 0 without RCCL */
    uint32_t forms_tried, forms_ok; /* bit MA_SELFTEST_FORM_* per exchange form */
    int32_t peer_pairs, peer_pairs_ok;
    int32_t stamp_waits, stamp_waits_ok;
    int32_t failed_form;          /* MA_SELFTEST_FORM_* of the form that failed, -1 */
    int32_t failed_member;        /* the member (rank) a failure was seen on, -1 */
    int32_t timed_out;            /* 1: the failure was a deadline (the group / communicator is broken now) */
    double form_us[8];            /* wall microseconds of each form's exchange + wait */
    double peer_us_max, stamp_us_max;
    char text[1024];              /* "PASS: ..." / "FAIL: ..." */
} ma_selftest_report;
#define MA_SELFTEST_FORMS 8
enum {
    MA_SELFTEST_FORM_IN_STREAM_THREADS = 0, MA_SELFTEST_FORM_IN_STREAM_CALLER = 1,
    MA_SELFTEST_FORM_OVERLAP_EVENT_THREADS = 2, MA_SELFTEST_FORM_OVERLAP_EVENT_CALLER = 3,
    MA_SELFTEST_FORM_OVERLAP_STAMP_THREADS = 4, MA_SELFTEST_FORM_OVERLAP_STAMP_CALLER = 5,
    MA_SELFTEST_FORM_HOST_FOLD = 6
};
enum {
    MA_SELFTEST_EXCHANGE = 1, MA_SELFTEST_EXCHANGE_ALL_FORMS = 2, MA_SELFTEST_PEER_COPIES = 4, MA_SELFTEST_STAMPS = 8,
    MA_SELFTEST_OVERLAP_EVENT = 16, MA_SELFTEST_OVERLAP_STAMP = 32 /* ma_comm_selftest only: one overlapped form each */
};
enum { MA_GROUP_HANDOFF_STAMP = 0, MA_GROUP_HANDOFF_EVENT = 1 };
ma_status ma_group_synchronize_for(ma_group* group, double timeout_ms);
/* MA_GROUP_SCAN_LANES: every member's own stream waits (an event) for everything its second scan lane has been given so far —
 * what a host calls before it enqueues work of its own on a member's context that must come AFTER the group's steps. A no-op
 * for a group without lanes. Enqueue-only. */
ma_status ma_group_join_lanes(ma_group* group);
/* MA_GROUP_SCAN_LANES: use the second lanes (1) or run every step on the members' own streams (0) without a rebuild. The lanes gain
 * -4 ... -5 % per step of the 8-way share while consecutive scans overlap by their tails only (36 of 36 processes on this pool's
 * one-GPU boxes: profiles/r05_early_mode.txt); a host that wants to be sure on its own node measures a few dozen un-timed steps
 * each way and keeps the faster, as bench.py does. on = 2: use them on FRESH second contexts (new streams; the steps start from
 * rest: worth a try when the lanes measured no faster). Drains the group first. MA_ERR_UNSUPPORTED for on != 0 on a group without
 * lanes. */
ma_status ma_group_set_scan_lanes(ma_group* group, int32_t on);
int32_t ma_group_is_broken(ma_group* group);
ma_status ma_group_rebuild_exchange(ma_group* group, uint32_t flags);
uint32_t ma_group_flags(ma_group* group);
ma_status ma_group_set_handoff(ma_group* group, int32_t kind);
int32_t ma_group_handoff(ma_group* group);
ma_status ma_group_selftest(ma_group* group, uint32_t what, double timeout_ms, ma_selftest_report* out_report);
/* The sum of ONE column held as many chunks spread over the group's GPUs — a SuperArray, or one column of the batches of a
 * SuperTable at the reference's own batch sizes (BASELINE config 5 with 8192-row batches: 122 000 chunks per 10^9 rows).
 * Chunk i belongs to member i % size and must be resident there; every member sums ITS chunks with one ma_sum_chunks
 * pass into its record of `column` (integer formats: the integer slots; 'f' / 'g': a (hi, lo) pair in the float slots),
 * all members concurrently. Enqueue-only like ma_group_enqueue_sum_*: ma_group_exchange, ma_group_synchronize, then
 * ma_group_result(column) holds the total — wrapping, or within 1 ULP of the exactly rounded sum. format_code as
 * ma_sum_columns. */
ma_status ma_group_enqueue_sum_chunks(ma_group* group, int32_t column, int32_t format_code, size_t n_chunks,
                                      const void* const* chunk_data, const size_t* chunk_lens,
                                      const uint8_t* const* chunk_masks, const size_t* chunk_mask_offsets);

/* route_super_array_broadcast (src/kernels/broadcast/super_array.rs:180-251; its chunk loop carries "// TODO:
 * Parallelise", :193) over the GPUs of a group: chunk pair i is computed by member i % ma_group_size(group), on whose
 * device its buffers must be resident (or in host memory); the pairs of one member run as one launch, the members
 * concurrently, nothing crosses between GPUs and the result stays chunked where its inputs are. Arguments and rules as
 * ma_route_super_array_broadcast, except member_mask_overrides: NULL, or one pointer per MEMBER (the override bitmap as
 * resident on that member's device; an entry may be NULL). ENQUEUES only: ma_group_synchronize waits and reports a dense
 * integer division by zero (MA_ERR_DIVIDE_BY_ZERO); out_has_mask is filled before the call returns. */
ma_status ma_group_route_super_array_broadcast(ma_group* group, int32_t format_code, int32_t op, size_t n_chunks,
                                               const void* const* lhs_data, const size_t* lhs_lens,
                                               const uint8_t* const* lhs_masks, const void* const* rhs_data,
                                               const size_t* rhs_lens, const uint8_t* const* rhs_masks,
                                               const uint8_t* const* member_mask_overrides, void* const* out_data,
                                               uint8_t* const* out_masks, int32_t* out_has_mask);
/* SuperTable::consolidate (src/structs/chunked/super_table.rs:657-743, src/traits/consolidate.rs:80-207) for a column
 * whose batches live on different GPUs: chunk i is device memory of member i % ma_group_size(group); the consolidated
 * column (and its validity, present iff some chunk has one; chunks without contribute all-valid rows) is written to
 * out_data / out_mask, device memory of member dest_member. Owners push their chunks into place with peer copies over
 * xGMI on their own streams, concurrently; validity is gathered on byte boundaries and joined at bit granularity on the
 * destination. Arguments otherwise as ma_consolidate_column. ENQUEUES only: the destination member's stream is ordered
 * behind the copies, ma_group_synchronize waits for everything. A batch-sharded table needs this only when one
 * contiguous column is explicitly asked for — its per-column reduce moves no bytes (ma_group_enqueue_sum_*). */
ma_status ma_group_consolidate_column(ma_group* group, int32_t dest_member, size_t elem_size, size_t n_chunks,
                                      const void* const* chunk_data, const size_t* chunk_lens,
                                      const uint8_t* const* chunk_masks, const size_t* chunk_mask_offsets, void* out_data,
                                      uint8_t* out_mask, int32_t* out_has_mask);
ma_status ma_group_result(ma_group* group, int32_t column, int64_t* out_int_sum, uint64_t* out_int_count,
                          double* out_f64_sum, uint64_t* out_f64_count);
ma_status ma_group_member_result(ma_group* group, int32_t member, int32_t column, int64_t* out_int_sum,
                                 uint64_t* out_int_count, double* out_f64_sum, uint64_t* out_f64_count);
ma_status ma_group_sum_i64(ma_group* group, const int64_t* const* chunk_data, const size_t* chunk_lens,
                           const uint8_t* const* chunk_masks, const size_t* chunk_mask_offsets, int64_t* out_sum,
                           uint64_t* out_valid_count);
ma_status ma_group_sum_f64(ma_group* group, const double* const* chunk_data, const size_t* chunk_lens,
                           const uint8_t* const* chunk_masks, const size_t* chunk_mask_offsets, double* out_sum,
                           uint64_t* out_valid_count);

/* ------------------------------------------------------------------------------------------------
 * The same exchange between PROCESSES (one process per GPU — e.g. one Rust worker per device, or ranks started by a
 * launcher): an RCCL communicator bound to a context. Rank 0 calls ma_comm_unique_id and hands the 128 bytes to the
 * other ranks by the host's own means (a file, a socket, MPI, torch.distributed's store); every rank then calls
 * ma_comm_create (collective: it returns once all n_ranks ranks have joined). Collectives are enqueued on the
 * context's stream and follow its sync / async mode; buffers must be device-reachable.
 *   ma_comm_sum_exchange  local_records = slots_per_rank x n_columns records of 8 x u64 ([0] integer sum, [1] integer
 *       valid count, [2] f64 hi bits, [3] f64 lo bits, [4] float valid count — what ma_<t>_sum / ma_<t>_sum_dd write when
 *       pointed at them); ONE ncclAllGather into `gathered` (n_ranks x that), then column c is folded over (rank, slot)
 *       in that order into out_finals[4c .. 4c+3] = [integer sum, integer count, f64 sum bits, float count] — the
 *       job's finals, bit-identical on every rank (ma_fold_sum_records' rule). slots_per_rank > 1 serves a rank that
 *       holds several batches of a SuperTable (src/structs/chunked/super_table.rs:78-83): the fold is in batch order.
 *   ma_comm_sum_exchange_overlapped  the same exchange on a stream of the communicator's own, behind everything the
 *       context's stream has been given so far; the context's stream goes on at once — the scans of step k + 1 run while
 *       step k's records cross the fabric (a partitioned 10^9-row column leaves each of 8 GPUs 0.14 ms of scan per column,
 *       an all-gather's latency is a fifth of that). Two record sets alternate: `slot` (0 / 1) names the one being
 *       exchanged; ma_comm_slot_wait(slot) puts the context's stream behind that slot's last exchange — call it before the
 *       kernels that overwrite the slot's records and before reading its out_finals in stream order;
 *       ma_comm_synchronize waits for both streams. Always enqueue-only, whatever the context's mode.
 *   ma_comm_all_gather / ma_comm_all_reduce_sum_i64  the bare collectives (wrapping integer sum).
 * ma_rccl_version: ncclGetVersion's code (e.g. 22707), 0 when RCCL cannot be loaded.
 * ---------------------------------------------------------------------------------------------- */
typedef struct ma_comm ma_comm;
#define MA_COMM_ID_BYTES 128
int32_t ma_rccl_version(void);
/* Which librccl the library opened: the one MINARROW_HIP_RCCL_PATH names (and no other: a path that does not open is an
 * error), else the one beside its own HIP runtime (see ma_hip_runtime_path); "" when none. When the named library is the
 * loopback collective double of tests/loopback_rccl (a test stand-in that lets several ranks share ONE device so that the
 * multi-rank paths can be rehearsed on a one-GPU box) the string starts with "REHEARSAL", ma_rccl_version is 9900, and
 * ma_group_exchange_note starts with "REHEARSAL" too: nothing measured through it is a multi-GPU figure. */
const char* ma_rccl_path(void);
ma_status ma_comm_unique_id(uint8_t* out_id);
ma_status ma_comm_create(ma_ctx* ctx, const uint8_t* id, int32_t rank, int32_t n_ranks, ma_comm** out_comm);
/* Bounded like ma_group_destroy: 10 s for what is in flight, then the abort path. */
void ma_comm_destroy(ma_comm* comm);
int32_t ma_comm_rank(ma_comm* comm);
int32_t ma_comm_size(ma_comm* comm);
ma_status ma_comm_all_gather(ma_comm* comm, const void* send, void* recv, size_t bytes_per_rank);
ma_status ma_comm_all_reduce_sum_i64(ma_comm* comm, const int64_t* send, int64_t* recv, size_t count);
ma_status ma_comm_sum_exchange(ma_comm* comm, const uint64_t* local_records, size_t slots_per_rank, size_t n_columns,
                               uint64_t* gathered, uint64_t* out_finals);
ma_status ma_comm_sum_exchange_overlapped(ma_comm* comm, int32_t slot, const uint64_t* local_records, size_t slots_per_rank,
                                          size_t n_columns, uint64_t* gathered, uint64_t* out_finals);
/* ma_comm_sum_exchange_overlapped whose exchange stream waits for `*stamp >= stamp_value` (hipStreamWaitValue64) instead of
 * an event recorded on the context's stream: the scans that fill this slot's records must end with a launch that stamps it
 * (ma_sum_fused_stamped). The context's stream then carries nothing but the scans. The stamp stays the CALLER's: should the
 * communicator be aborted while its exchange stream waits on it, the abort releases that wait by storing all-ones into the
 * word — only while the word is still a live stamp (never into one ma_stamp_free has taken back) — and stores `stamp_value`
 * back once the streams have run empty; if they never do (ma_comm_is_broken == 2) the word keeps all-ones and must not be
 * handed to another communicator: take a fresh stamp. */
ma_status ma_comm_sum_exchange_overlapped_on_stamp(ma_comm* comm, int32_t slot, uint64_t* stamp, uint64_t stamp_value,
                                                   const uint64_t* local_records, size_t slots_per_rank, size_t n_columns,
                                                   uint64_t* gathered, uint64_t* out_finals);
ma_status ma_comm_slot_wait(ma_comm* comm, int32_t slot);
/* ma_comm_slot_wait for a host that fills the slot's records from ANOTHER context of the communicator's device (two scan
 * contexts in turn, the second gated on the first's early stamp: ma_sum_fused_stamped_early): `ctx`'s stream is the one put
 * behind the slot's last exchange. */
ma_status ma_comm_slot_wait_on(ma_comm* comm, int32_t slot, ma_ctx* ctx);
/* As ma_group_exchange_stats, for this rank's ma_comm_sum_exchange / _overlapped calls. */
ma_status ma_comm_exchange_stats(ma_comm* comm, double* out_all_gather_us, double* out_fold_us, int32_t* out_samples,
                                 int32_t* out_rccl_ranks);
ma_status ma_comm_synchronize(ma_comm* comm);
/* The multi-process twins of ma_group_synchronize_for / _is_broken / _selftest (see there). ma_comm_synchronize_for waits for
 * the context's stream and the communicator's exchange stream for at most timeout_ms; past the deadline the communicator is
 * aborted (ma_comm_abort) and MA_ERR_DEVICE names the stream that was pending. An aborted communicator refuses every
 * collective with MA_ERR_DEVICE and can only be destroyed; the ranks agree by their own means (the channel that carried the
 * unique id) whether to make a new one (a fresh ma_comm_unique_id) or to go on without. ma_comm_abort is what a rank calls
 * when ANOTHER rank reports the deadline: its own wait may have succeeded, the communicator is dead all the same.
 * ma_comm_selftest is collective: every rank calls it; rank-tagged records go through ma_comm_sum_exchange (in-stream),
 * ma_comm_sum_exchange_overlapped (event) and _on_stamp (stamp) — forms IN_STREAM_CALLER, OVERLAP_EVENT_CALLER,
 * OVERLAP_STAMP_CALLER of the report — each under the deadline, the gathered blocks and the finals checked on this rank;
 * `what`: MA_SELFTEST_EXCHANGE = the in-stream form, MA_SELFTEST_OVERLAP_EVENT / MA_SELFTEST_OVERLAP_STAMP = that overlapped
 * form (both record sets), 0 or MA_SELFTEST_EXCHANGE_ALL_FORMS = all three; peer copies are not a communicator's. */
ma_status ma_comm_synchronize_for(ma_comm* comm, double timeout_ms);
ma_status ma_comm_abort(ma_comm* comm);
int32_t ma_comm_is_broken(ma_comm* comm);
ma_status ma_comm_selftest(ma_comm* comm, uint32_t what, double timeout_ms, ma_selftest_report* out_report);


#ifdef __cplusplus
} /* extern "C" */
#endif

#endif /* MINARROW_HIP_H */
