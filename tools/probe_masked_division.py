import sys, json, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
from minarrow_amd.host import Context
ctx = Context(0)
k, per = 60000, 8192
n = k * per
a, b, o = ctx.alloc(n * 4 + 64), ctx.alloc(n * 4 + 64), ctx.alloc_output(n * 4 + 64)
m, om = ctx.alloc(n // 8 + 64), ctx.alloc(n // 8 + 64)
ctx.synth_iota("i32", a, n, 1); ctx.synth_iota("i32", b, n, 0)
ctx.synth_validity(m, n, seed=2, null_every=10)
tab = lambda xs: C.cast((C.c_void_p * k)(*xs), C.c_void_p)
t_a, t_b, t_o = tab([a.ptr + i * per * 4 for i in range(k)]), tab([b.ptr + i * per * 4 for i in range(k)]), tab([o.ptr + i * per * 4 for i in range(k)])
t_m, t_om = tab([m.ptr + i * per // 8 for i in range(k)]), tab([om.ptr + i * per // 8 for i in range(k)])
t_n = C.cast((C.c_size_t * k)(*([per] * k)), C.c_void_p)
ctx.set_async(True)
def timed(fn, reps=5):
    fn(); ctx.synchronize(); fn(); ctx.timer_start()
    for _ in range(reps): fn()
    ctx.timer_stop(); return ctx.timer_elapsed_ms() / reps
for op, name in ((0, "add"), (3, "div"), (6, "floordiv")):
    ms = timed(lambda: ctx.lib.ma_route_super_array_broadcast(ctx.handle, ord("i"), op, k, t_a, t_n, t_m, t_b, t_n, None, None, t_o, t_om, None))
    print(json.dumps({"op": name + " masked, 60000 x 8192 i32", "ms": round(ms, 3), "gbps": round((3 * n * 4 + 2 * n / 8) / ms / 1e6, 1)}), flush=True)
