#!/usr/bin/env python3
"""ma_consolidate_table_arena with many small batches (a SuperTable at RechunkStrategy::Auto batch sizes): 20 000 batches x
8192 rows x 4 columns (i64, f64 with validity, i32, f64), against the same-process copy."""
import ctypes as C
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from minarrow_amd.host import Context, arena_layout  # noqa: E402

ctx = Context(0)
nb, rows = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (20000, 8192)
elem = [8, 8, 4, 8]
total = nb * rows
src = [ctx.alloc(total * e + 64) for e in elem]
for s, e in zip(src, elem):
    ctx.synth_iota("i64" if e == 8 else "i32", s, total, 1)
mask = ctx.alloc(total // 8 + 64)
ctx.synth_validity(mask, total, seed=5, null_every=10)
d_off, m_off, cap, used = arena_layout(elem, [False, True, False, False], total)
arena = ctx.alloc_output(cap + 64)
k = len(elem) * nb
cells = (C.c_void_p * k)(*[src[c].ptr + b * rows * elem[c] for c in range(len(elem)) for b in range(nb)])
masks = (C.c_void_p * k)(*[(mask.ptr + b * rows // 8) if c == 1 else None for c in range(len(elem)) for b in range(nb)])
es = (C.c_size_t * len(elem))(*elem)
br = (C.c_size_t * nb)(*([rows] * nb))
do, mo, us = (C.c_size_t * len(elem))(), (C.c_size_t * len(elem))(), C.c_size_t()
cast = lambda a: C.cast(a, C.c_void_p)  # noqa: E731
ctx.set_async(True)


def call():
    st = ctx.lib.ma_consolidate_table_arena(ctx.handle, len(elem), nb, cast(es), cast(br), cast(cells), cast(masks), None, arena.ptr,
                                            cap + 64, cast(do), cast(mo), C.addressof(us))
    assert st == 0, st


def timed(fn, reps=5):
    fn(); fn(); ctx.synchronize(); fn(); ctx.timer_start()
    for _ in range(reps):
        fn()
    ctx.timer_stop()
    return ctx.timer_elapsed_ms() / reps


o = ctx.alloc_output(total * 8 + 64)
cms = timed(lambda: ctx.consolidate_column(8, [src[0]], [total], o))
copy = 2 * total * 8 / cms / 1e6
ms = timed(call)
b = 2 * total * sum(elem) + 2 * total / 8
print(json.dumps({"batches": nb, "rows": rows, "ms": round(ms, 3), "gbps": round(b / ms / 1e6, 1), "of_copy": round(b / ms / 1e6 / copy, 3),
                  "copy_gbps": round(copy, 1)}))
