#!/usr/bin/env python3
"""ma_sum_chunks / ma_sum_columns at 60 000 x 8192-row chunks: event time per call back to back, per library variant
(0 = descriptors read in place from pinned memory; 4096 = the table copied to the device on the stream first).
Run under `rocprofv3 --kernel-trace --stats` for the kernels' own durations."""
import ctypes as C
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from minarrow_amd.host import Context  # noqa: E402

ctx = Context(0)
k, per = 60_000, 8192
if len(sys.argv) > 4:  # e.g. 1000x536832: the workgroup-per-segment form (columns longer than a segment)
    k, per = (int(x) for x in sys.argv[4].split("x"))
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "4096", "8192", "12288"])]
bpcs = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["0"])]
only = sys.argv[3].split(",") if len(sys.argv) > 3 else None
for tag, fmt, sz in (("i64", "l", 8), ("i32", "i", 4), ("f64", "g", 8), ("f32", "f", 4), ("i16", "s", 2), ("u8", "C", 1)):
    if only and tag not in only:
        continue
    n = k * per
    a = ctx.alloc(n * sz + 64)
    m = ctx.alloc(n // 8 + 64)
    slot = ctx.alloc(256)
    if tag in ("i16", "u8"):  # no generator for the narrow types: any bytes will do (the variants must agree on the sum)
        ctx.synth_splitmix("i64", a, (n * sz) // 8, 7)
    else:
        ctx.synth_iota(tag, a, n, 1)
    ctx.synth_validity(m, n, seed=1, null_every=10)
    tab = lambda xs: C.cast((C.c_void_p * k)(*xs), C.c_void_p)  # noqa: E731
    t_a = tab([a.ptr + i * per * sz for i in range(k)])
    t_m = tab([m.ptr + i * per // 8 for i in range(k)])
    t_n = C.cast((C.c_size_t * k)(*([per] * k)), C.c_void_p)
    ctx.set_async(True)
    seen = {}
    for variant, bpc in [(v, b) for v in variants for b in bpcs]:
        ctx.lib.ma_ctx_set_variant(ctx.handle, variant)
        ctx.lib.ma_ctx_set_blocks_per_cu(ctx.handle, bpc)
        for masked in (False, True):
            def fn():
                st = ctx.lib.ma_sum_chunks(ctx.handle, ord(fmt), k, t_a, t_n, t_m if masked else None, None,
                                           slot.ptr if tag in ("f64", "f32") else None, slot.ptr + 8 if tag not in ("f64", "f32") else None, slot.ptr + 16)
                assert st == 0, st
            fn(); fn()
            best = None
            for _ in range(3):
                ctx.synchronize()
                fn()
                ctx.timer_start()
                for _ in range(20):
                    fn()
                ctx.timer_stop()
                ms = ctx.timer_elapsed_ms() / 20
                best = ms if best is None else min(best, ms)
            ctx.synchronize()
            got = tuple(int(x) for x in slot.download("uint64", 3))
            want = seen.setdefault((tag, masked), got)
            assert got == want, (tag, variant, masked, got, want)
            print(json.dumps({"type": tag, "variant": variant, "blocks_per_cu": bpc, "masked": masked, "ms": round(best, 4),
                              "tbps": round(n * sz / best / 1e9, 3)}), flush=True)
    ctx.lib.ma_ctx_set_variant(ctx.handle, 0)
    ctx.lib.ma_ctx_set_blocks_per_cu(ctx.handle, 0)
    red = getattr(ctx.lib, f"ma_{tag}_sum")
    def one():
        if tag in ("f64", "f32"):
            st = red(ctx.handle, a.ptr, n, None, 0, 0, slot.ptr, slot.ptr + 16)
        else:
            st = red(ctx.handle, a.ptr, n, None, 0, 0, slot.ptr + 8, slot.ptr + 16)
        assert st == 0, st
    one(); one()
    ctx.synchronize()
    ctx.timer_start()
    for _ in range(20):
        one()
    ctx.timer_stop()
    ms = ctx.timer_elapsed_ms() / 20
    print(json.dumps({"type": tag, "plain_sum_ms": round(ms, 4), "tbps": round(n * sz / ms / 1e9, 3)}), flush=True)
    ctx.set_async(False)
    ctx.synchronize()
    a.free(); m.free(); slot.free()
