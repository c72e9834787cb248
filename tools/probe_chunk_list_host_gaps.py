#!/usr/bin/env python3
"""What 30 consecutive asynchronous ma_sum_columns calls over 60 000 chunks of 8192 i32 rows cost the HOST thread, call by call, and
the wall time per call (profiles/r06_step4_host_gaps.txt): shows a host waiting for a staging slot or a table buffer, and stalls
of the runtime when nothing throttles it. MINARROW_HIP_LIB selects another build of the library to compare with."""
import ctypes as C, json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from minarrow_amd.host import Context
K=60000; PER=8192
ctx=Context(0)
out=ctx.alloc(3*8*131072)
n=K*PER
a=ctx.alloc(n*4+64); mask=ctx.alloc(n//8+128)
ctx.synth_iota("i32", a, n, 1); ctx.synth_validity(mask, n, seed=5, null_every=10)
ptrs=C.cast((C.c_void_p*K)(*[a.ptr+i*PER*4 for i in range(K)]), C.c_void_p)
lens=C.cast((C.c_size_t*K)(*([PER]*K)), C.c_void_p)
mks=C.cast((C.c_void_p*K)(*[mask.ptr+i*(PER//8) for i in range(K)]), C.c_void_p)
ctx.set_async(True)
def cols(masked):
    assert ctx.lib.ma_sum_columns(ctx.handle, ord("i"), K, ptrs, lens, mks if masked else None, None, out.ptr, out.ptr+8*131072, out.ptr+16*131072)==0
for masked in (False, True, False):
    cols(masked); cols(masked); ctx.synchronize()
    ts=[time.perf_counter()]
    for _ in range(30):
        cols(masked); ts.append(time.perf_counter())
    ctx.synchronize(); te=time.perf_counter()
    gaps=[(b-a)*1e6 for a,b in zip(ts,ts[1:])]
    print("masked" if masked else "dense ", f"wall/call {(te-ts[0])/30*1e6:.0f} us; host enqueue us:", ' '.join(f"{g:.0f}" for g in gaps), f"| final sync {(te-ts[-1])*1e6:.0f}")
