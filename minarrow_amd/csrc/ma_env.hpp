// The library's ENVIRONMENT SURFACE in one place. Every MINARROW_HIP_* variable a source file of the product reads has a line
// here — `// ENV <name> | <default> | <meaning>` — and tools/gen_env_table.py turns these lines into INTEGRATION.md section 5's
// table (tests/test_abi_symbols.py: the lines, the variables the sources actually read, and the table must agree). The reference's
// build-time knobs are cargo features and build.rs lane counts (Cargo.toml:66-226, build.rs:53-65); a device library needs run-time
// ones. [tuning] = read by the tuning build only (make -C minarrow_amd/csrc TUNING=1): the shipped library keeps the default.
#pragma once
// ENV MINARROW_HIP_DEVICES | all visible devices | `"2,3,5"`: the library's device ordinal *i* is HIP device *list[i]* (`ma_device_count()` = list length) — a per-library `HIP_VISIBLE_DEVICES`
// ENV MINARROW_HIP_MIN_ROWS | 262144 | `ma_min_device_rows_for(MA_KIND_REDUCTION)`: below this length a host shim keeps the reference's CPU kernels (the library never computes on the CPU)
// ENV MINARROW_HIP_MIN_ROWS_ELEMENTWISE | 32768 | the same for `MA_KIND_ELEMENTWISE`
// ENV MINARROW_HIP_MIN_BITS_SCAN | 2097152 | the same for `MA_KIND_BITMASK_SCAN`
// ENV MINARROW_HIP_STAGING_TILE | 32 MiB | bytes of one operand per tile of the pipelined host-operand path (`ma_ctx_set_staging_tile`); 0 = stage whole operands
// ENV MINARROW_HIP_LANES | 4 | lanes per context for overlapping synchronous calls from several threads (1 = serialise)
// ENV MINARROW_HIP_POLL_US | 60 | a synchronous call polls a pinned completion word for at most this many µs before the blocking stream wait (whose wake-up alone costs ~10 µs); 0 = always block
// ENV MINARROW_HIP_FENCED_REDUCE | off | `1` (at context creation): the sum kernels publish their per-workgroup partials with agent-scope release / acquire fences instead of write-through stores — a safety valve, results are identical, 20–50 % slower on mid-size columns
// ENV MINARROW_HIP_OUTPUT_SEARCH | off | `1`: `ma_dev_alloc_output` measures candidate blocks and returns the fastest-writing one (also `ma_dev_output_search(1)`); otherwise it is `ma_dev_alloc`
// ENV MINARROW_HIP_OUTPUT_CANDIDATES | 6 | [tuning] with the search on: candidate blocks measured per request
// ENV MINARROW_HIP_OUTPUT_MIN_BYTES | 256 MiB | [tuning] … from which block size
// ENV MINARROW_HIP_OUTPUT_GOOD_GBPS | calibrated | [tuning] … the write rate at which the search stops (0 = 0.97 × the device's tight-front write rate)
// ENV MINARROW_HIP_OUTPUT_HOLD_PERCENT | 25 | [tuning] … the share of the free HBM the candidates alive at one time may take
// ENV MINARROW_HIP_PINNED_POOL_BYTES | 2 GiB | limit of the pinned block cache (`ma_pinned_pool_trim` / `_set_limit` at run time)
// ENV MINARROW_HIP_DEV_POOL_BYTES | 16 GiB | limit of the per-device block cache
// ENV MINARROW_HIP_GROUP_EXCHANGE | host fold | `rccl`: `ma_group_create` uses the RCCL exchange (host fold as fallback)
// ENV MINARROW_HIP_GROUP_ISSUE | threads | `caller`: `ma_group_*` calls issue every member's launches from the calling thread instead of one issue thread per member
// ENV MINARROW_HIP_RCCL_PATH | the librccl beside the library's own libamdhip64 | the collective library to open instead, and no other (a path that does not open is an error): another RCCL build, or the loopback double of `tests/loopback_rccl` — then `ma_rccl_path` / `ma_group_exchange_note` start with REHEARSAL
// ENV MINARROW_HIP_GUARD_LOG | off | `1`: the bounded waits, aborts and rebuilds of `ma_group_*` / `ma_comm_*` / `ma_scan_lanes_*` write a timestamped trace to stderr
// ENV MINARROW_HIP_DESTROY_WAIT_MS | 10000 | how long `ma_group_destroy` / `ma_comm_destroy` / `ma_scan_lanes_destroy` wait for work still in flight before they release and abort instead of waiting for good
// ENV MINARROW_HIP_TEST_HOOKS | off | `1` WHEN THE LIBRARY IS LOADED: the fault hooks of `include/minarrow_hip_testing.h` act; otherwise every one of them returns `MA_ERR_UNSUPPORTED`
// ENV MINARROW_HIP_STAMP_SIGNAL | device word | [tuning] `1`: `ma_stamp_alloc` takes the runtime's 8-byte signal memory (the command processor's wait on it holds up the scan stream of an overlapped step by 19–25 %)
// ENV MINARROW_HIP_SCAN_LANE_CLASS | ordinary | [tuning] `high`: a group's second scan lanes in the high stream priority class
// ENV MINARROW_HIP_STREAM_PRIORITY | ordinary | [tuning] `high` / `low` (read at every context creation): the context's stream in that priority class
// ENV MINARROW_HIP_LIB | minarrow_amd/lib/libminarrow_hip.so | (read by `minarrow_amd/ffi.py`, not by the library) another build of the library to load: the tuning build, `build/tuning/libminarrow_hip.so`
