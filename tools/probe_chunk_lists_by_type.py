#!/usr/bin/env python3
"""ma_sum_columns over 60 000 chunks of 8192 rows by element type (u8, i16, i32, f32; dense and with 10 % nulls), 30 asynchronous calls
each: call-to-call time against what the calls cost the host thread (profiles/r06_chunk_lists_by_type.txt) — where the two are
equal the host's time per chunk descriptor is the bound."""
import ctypes as C, json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from minarrow_amd.host import Context
K=60000; PER=8192
ctx=Context(0)
out=ctx.alloc(3*8*131072)
for tag,fmt,size in (("u8","C",1),("i16","s",2),("i32","i",4),("f32","f",4)):
    n=K*PER
    a=ctx.alloc(n*size+64); mask=ctx.alloc(n//8+128)
    ctx.synth_iota("i64", a, n*size//8, 0x0102030405060708); ctx.synth_validity(mask, n, seed=5, null_every=10)
    ptrs=C.cast((C.c_void_p*K)(*[a.ptr+i*PER*size for i in range(K)]), C.c_void_p)
    lens=C.cast((C.c_size_t*K)(*([PER]*K)), C.c_void_p)
    mks=C.cast((C.c_void_p*K)(*[mask.ptr+i*(PER//8) for i in range(K)]), C.c_void_p)
    ctx.set_async(True)
    for masked in (False, True):
        def call():
            assert ctx.lib.ma_sum_columns(ctx.handle, ord(fmt), K, ptrs, lens, mks if masked else None, None, out.ptr, out.ptr+8*131072, out.ptr+16*131072)==0
        call(); call(); ctx.synchronize()
        t0=time.perf_counter()
        for _ in range(30): call()
        th=time.perf_counter()
        ctx.synchronize(); t1=time.perf_counter()
        nbytes=n*size+(n/8 if masked else 0)
        print(tag, "nulls" if masked else "dense", f"call-to-call {(t1-t0)/30*1e6:.0f} us = {nbytes/((t1-t0)/30)/8e12:.3f} of peak; host enqueue {(th-t0)/30*1e6:.0f} us/call")
    ctx.set_async(False); ctx.synchronize(); a.free(); mask.free()
