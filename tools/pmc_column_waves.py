#!/usr/bin/env python3
"""The reference's default chunk shape (RechunkStrategy::Auto: 8192-row chunks, src/structs/chunked/super_array.rs:51-59) as the
per-column sums see it, launched a few times for rocprofv3 passes (tools/pmc_column_waves.sh): 60 000 i32 (and i64) columns of 8192
rows through ma_sum_columns (column_waves_kernel<T, 8, false>: a {sum, count} per column), the SAME chunk list as one column
through ma_sum_chunks (column_waves_kernel<T, 8, true>: the same loop without the per-column epilogue), and the same bytes as ONE
contiguous column through ma_<t>_sum — each dense and with 10 % nulls. argv[2] = columns (default 60000; 122070 = 10^9 rows),
argv[3] = a substring of the job names ("dense", "gated", "columns_i32_gated" ...) to launch only those. With 20 or more
repetitions the process's first timed job is preceded by 0.4 s of the same calls (clock ramp; CW_NO_RAMP=1 leaves it out, for
per-launch traces).
Prints one JSON line of wall-clock figures (the judged ones are the profiler's)."""
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from minarrow_amd.host import Context  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
K = int(sys.argv[2]) if len(sys.argv) > 2 else 60000
ONLY = sys.argv[3] if len(sys.argv) > 3 else ""  # "dense" / "gated": only those jobs (the kernels' names do not tell them apart)
PER = 8192
ctx = Context(0)
res = {}
out = ctx.alloc(3 * 8 * 131072)
for tag, fmt, size in (("i32", "i", 4), ("i64", "l", 8)):
    n = K * PER
    a = ctx.alloc(n * size + 64)
    mask = ctx.alloc(n // 8 + 128)
    ctx.synth_iota(tag, a, n, 1)
    ctx.synth_validity(mask, n, seed=5, null_every=10)
    ptrs = C.cast((C.c_void_p * K)(*[a.ptr + i * PER * size for i in range(K)]), C.c_void_p)
    lens = C.cast((C.c_size_t * K)(*([PER] * K)), C.c_void_p)
    mks = C.cast((C.c_void_p * K)(*[mask.ptr + i * (PER // 8) for i in range(K)]), C.c_void_p)
    ctx.set_async(True)
    slot = out.ptr

    def cols(masked):
        assert ctx.lib.ma_sum_columns(ctx.handle, ord(fmt), K, ptrs, lens, mks if masked else None, None, out.ptr, out.ptr + 8 * 131072,
                                      out.ptr + 16 * 131072) == 0

    def chunks(masked):
        assert ctx.lib.ma_sum_chunks(ctx.handle, ord(fmt), K, ptrs, lens, mks if masked else None, None, out.ptr, out.ptr + 8 * 131072,
                                     out.ptr + 16 * 131072) == 0

    jobs = {f"columns_{tag}_dense": lambda: cols(False), f"columns_{tag}_gated": lambda: cols(True),
            f"chunks_as_one_{tag}_dense": lambda: chunks(False), f"chunks_as_one_{tag}_gated": lambda: chunks(True),
            f"one_column_{tag}_dense": lambda: ctx.sum_into(tag, a, n, out_sum=slot, out_count=slot + 8),
            f"one_column_{tag}_gated": lambda: ctx.sum_into(tag, a, n, out_sum=slot, out_count=slot + 8, mask=mask)}
    for name, fn in jobs.items():
        if ONLY not in name:
            continue
        fn()
        fn()
        ctx.synchronize()
        if not res and reps >= 20 and not os.environ.get("CW_NO_RAMP"):  # the process's first timed job: bring the clocks up first (bench.py's --ramp-ms), or it reads 5-8 % low
            t_ramp = time.perf_counter()
            while time.perf_counter() - t_ramp < 0.4:
                for _ in range(8):
                    fn()
                ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        nbytes = n * size + (n / 8 if name.endswith("gated") else 0)
        res[name] = {"ms": round(ms, 4), "tbps": round(nbytes / ms / 1e9, 3), "frac_of_8TBps": round(nbytes / ms / 1e9 / 8, 3)}
    ctx.set_async(False)
    ctx.synchronize()
    a.free()
    mask.free()
print(json.dumps({"columns": K, "rows_per_column": PER, "wall": res}))
