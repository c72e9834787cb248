#!/usr/bin/env python3
"""Where does a segmented chunk-list consolidate lose against the one-segment form? (round 4)
    rocprofv3 --kernel-trace -d <dir> -o seg --output-format csv -- python3 tools/trace_segments.py run
    python3 tools/trace_segments.py show <dir>
`run`: 122 000 x 8192-row i64 chunks consolidated 4 times in each form (segments alternating over two streams = default,
segments on one stream = variant 2048, one segment = 1024), a 64-row fill between the forms as a marker.
`show`: per form and call, each concat_chunk_kernel dispatch {queue, start, duration, gap to the previous end} in us."""
import ctypes as C
import csv
import glob
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def run():
    from minarrow_amd.host import Context
    ctx = Context(0)
    per, k = 8192, 122_000
    n = per * k
    a, b = ctx.alloc(n * 8), ctx.alloc(n * 8)  # the allocation sequence of tools/ab_chunked.py (the output block's placement matters)
    o = ctx.alloc_output(n * 8)
    ctx.synth_iota("i64", a, n, 3)
    tab = lambda xs: C.cast((C.c_void_p * len(xs))(*xs), C.c_void_p)  # noqa: E731
    t_d = tab([a.ptr + i * per * 8 for i in range(k)])
    t_n = C.cast((C.c_size_t * k)(*([per] * k)), C.c_void_p)
    has = C.c_int32()
    ctx.set_async(True)
    for variant in (0, 2048, 1024):
        ctx.set_variant(variant)
        for _ in range(4):
            assert ctx.lib.ma_consolidate_column(ctx.handle, 8, k, t_d, t_n, None, None, o.ptr, None, C.addressof(has)) == 0
        ctx.synchronize()
        ctx.set_variant(0)
        ctx.synth_iota("i64", a, 64, 3)  # marker (rewrites the same values)
    ctx.set_async(False)
    ctx.close()


def show(d):
    for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
        form, forms = 0, ["two streams (default)", "one stream", "one segment"]
        t0, prev_end, seen = None, None, False
        for r in rows:
            name = r["Kernel_Name"]
            if "concat_chunk_kernel" not in name:
                if seen:
                    form, seen, prev_end = form + 1, False, None
                continue
            seen = True
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            t0 = t0 or s
            print(json.dumps({"form": forms[min(form, 2)], "queue": r.get("Queue_Id"), "grid": r.get("Grid_Size_X", r.get("Grid_Size")),
                              "start_us": round((s - t0) / 1e3, 1), "dur_us": round((e - s) / 1e3, 1),
                              "gap_to_prev_end_us": None if prev_end is None else round((s - prev_end) / 1e3, 1)}))
            prev_end = e if prev_end is None else max(prev_end, e)


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else show(sys.argv[2])
