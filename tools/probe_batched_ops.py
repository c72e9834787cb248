import sys, json, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np
from minarrow_amd.host import Context
ctx = Context(0)
B = 1 << 32
a, b, o = ctx.alloc(B + 64), ctx.alloc(B + 64), ctx.alloc_output(B + 64)
ctx.synth_iota("i32", a, B // 4, 1); ctx.synth_iota("i32", b, B // 4, 3)
ctx.set_async(True)
k = 8; n = B // 4; per = (n // k) // 64 * 64
tab = lambda xs: C.cast((C.c_void_p * k)(*xs), C.c_void_p)
t_l, t_r, t_o = tab([a.ptr + i * per * 4 for i in range(k)]), tab([b.ptr + i * per * 4 for i in range(k)]), tab([o.ptr + i * per * 4 for i in range(k)])
t_n = C.cast((C.c_size_t * k)(*([per] * k)), C.c_void_p)
sc = np.array([3], dtype=np.int32)
def timed(fn, reps=10):
    fn(); fn(); ctx.synchronize(); ctx.timer_start()
    for _ in range(reps): fn()
    ctx.timer_stop(); return ctx.timer_elapsed_ms() / reps
for op, name in ((0, "add"), (1, "sub"), (2, "mul"), (6, "floordiv")):
    ms_s = timed(lambda: ctx.lib.ma_broadcast_super_array_scalar(ctx.handle, ord("i"), op, 0, sc.ctypes.data, k, t_l, t_n, None, t_o, None, None))
    ms_r = timed(lambda: ctx.lib.ma_route_super_array_broadcast(ctx.handle, ord("i"), op, k, t_l, t_n, None, t_r, t_n, None, None, t_o, None, None))
    ms_1 = timed(lambda: ctx.apply_scalar("i32", "rhs", a, k * per, 3, op, o))
    print(json.dumps({"op": name, "scalar_8chunks_ms": round(ms_s, 4), "route_8chunks_ms": round(ms_r, 4), "single_array_scalar_ms": round(ms_1, 4)}), flush=True)
