"""Back-to-back fused i64 + f64 sums of mid-size columns on one GPU: one stream (ma_sum_fused on an async context) against the
pipeline (ma_scan_lanes_*: consecutive scans on two streams, each started by the early stamp of the one before). Four distinct
column pairs in turn (nothing is served from a cache that a stepping host would not have), 5 x `reps` scans each way, the
fastest batch counts (the slowest is printed too); wall time per scan and TB/s of the 16 bytes per row pair. MA_BENCH_VARIANT=<int>: the context's variant
word (bits 19-21 = early_mode + 1: when the early stamp is stored, ma_reduce_fused.hip). MA_BENCH_SINGLE=1: ONE i64 column per
scan (8 bytes per row) instead of the i64 + f64 pair; MA_BENCH_SINGLE=<format character> (c C s S i I l L f g): one column of that
type through the single-column kernels (ctx.sum_into on one stream against ma_scan_lanes_sum).
-> profiles/r05_scan_lanes_api.jsonl"""
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from minarrow_amd.host import Context, ScanLanes  # noqa: E402


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [1 << 20, 1 << 22, 1 << 24, 1 << 26, 125_000_000, 250_000_000]
    with Context(0) as ctx:
        ctx.set_async(True)
        variant = int(os.environ.get("MA_BENCH_VARIANT", "0"))
        single = os.environ.get("MA_BENCH_SINGLE", "") not in ("", "0")
        typed = os.environ.get("MA_BENCH_SINGLE", "") if os.environ.get("MA_BENCH_SINGLE", "") in tuple("cCsSiIlLfg") else ""
        tags = {"c": "i8", "C": "u8", "s": "i16", "S": "u16", "i": "i32", "I": "u32", "l": "i64", "L": "u64", "f": "f32", "g": "f64"}
        elem = {"c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}.get(typed, 8)
        row_bytes = elem if typed else (8 if single else 16)
        ctx.set_variant(variant)
        for rows in sizes:
            pairs = []
            for k in range(4):
                ci, cf = ctx.alloc(rows * 8 + 64), ctx.alloc(rows * 8 + 64)
                ctx.synth_iota("i64", ci, rows, k * rows)
                ctx.synth_iota("f64", cf, rows, k * rows)
                pairs.append((ci, cf))
            rec = ctx.alloc(64 * 4)
            reps = max(40, min(400, int(2e9 / rows)))  # (per scan: rows x 8 or 16 bytes)
            def table(k, ci, cf):
                cols = [("l", ci, rows, rec.ptr + 64 * k), ("g", cf, rows, rec.ptr + 64 * k + 16)]
                return cols[:1] if single else cols

            if typed:  # the column's bytes re-read as `rows` elements of the type (sums are not checked: `parity` says None)
                def one_stream(k, ci):
                    return lambda: ctx.sum_into(tags[typed], ci, rows, out_sum=rec.ptr + 64 * k, out_count=rec.ptr + 64 * k + 8)

                plain = [one_stream(k, ci) for k, (ci, cf) in enumerate(pairs)]
            else:
                plain = [ctx.prepare_sum_fused(table(k, ci, cf)) for k, (ci, cf) in enumerate(pairs)]
            with ScanLanes(ctx) as lanes:
                if typed:
                    piped = [lanes.prepare_sum(typed, ci, rows, rec.ptr + 64 * k, out_count=rec.ptr + 64 * k + 8) for k, (ci, cf) in enumerate(pairs)]
                else:
                    piped = [lanes.prepare_sum_fused(table(k, ci, cf)) for k, (ci, cf) in enumerate(pairs)]
                t_settle = time.perf_counter()  # clocks up, and the driver's background clear of VRAM an earlier process released
                while time.perf_counter() - t_settle < 0.4:  # left behind (profiles/r04_read_rate_states_root_cause.txt)
                    for k in range(8):
                        plain[k & 3]()
                    ctx.synchronize()
                best = {"one_stream": float("inf"), "scan_lanes": float("inf")}
                worst = {"one_stream": 0.0, "scan_lanes": 0.0}
                for _ in range(5):
                    for name, calls, drain in (("one_stream", plain, ctx.synchronize), ("scan_lanes", piped, lanes.synchronize)):
                        for k in range(8):
                            calls[k & 3]()
                        drain()
                        t0 = time.perf_counter()
                        for k in range(reps):
                            calls[k & 3]()
                        drain()
                        best[name] = min(best[name], (time.perf_counter() - t0) / reps)
                        worst[name] = max(worst[name], (time.perf_counter() - t0) / reps)
                want = [(k * rows * rows + rows * (rows - 1) // 2) & ((1 << 64) - 1) for k in range(4)]
                got = [int(rec.download(np.uint64, 1, 64 * k)[0]) for k in range(4)]
            print(json.dumps({"rows_per_column": rows, "columns_per_scan": 1 if single else 2, "variant": variant, "scans": reps, "type": tags.get(typed, "i64" if single else "i64+f64"), "parity": None if typed and typed != "l" else got == want,
                              "one_stream_us": round(best["one_stream"] * 1e6, 2), "scan_lanes_us": round(best["scan_lanes"] * 1e6, 2),
                              "one_stream_tbps": round(rows * row_bytes / best["one_stream"] / 1e12, 3),
                              "scan_lanes_tbps": round(rows * row_bytes / best["scan_lanes"] / 1e12, 3),
                              "ratio": round(best["scan_lanes"] / best["one_stream"], 4),
                              "slowest_batch_us": [round(worst["one_stream"] * 1e6, 2), round(worst["scan_lanes"] * 1e6, 2)]}), flush=True)
            for ci, cf in pairs:
                ci.free()
                cf.free()
            rec.free()


if __name__ == "__main__":
    main()
