"""Single-process multi-GPU group API (ma_group_*): row chunks on several contexts, folded on the host. The GPU
boxes of the test pool have one device, so the members share device 0 (independent contexts and streams); the
partition + fold logic is the same for 8 devices."""
import ctypes as C
import math

import numpy as np
import pytest

from minarrow_amd import ffi
from minarrow_amd.parallel import row_chunks

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("members", [1, 3, 8])
def test_group_sums_match_single_array(ctx, oracle, members):
    lib = ffi.load_library()
    n = 2_000_003
    rng = np.random.default_rng(members)
    ints = rng.integers(-(1 << 50), 1 << 50, size=n, dtype=np.int64)
    flts = rng.standard_normal(n) * 1e9
    bits = rng.integers(0, 256, size=n // 8 + 64, dtype=np.uint8)
    di, df, dm = ctx.to_device(ints, 64), ctx.to_device(flts, 64), ctx.to_device(bits, 16)
    g = C.c_void_p()
    devs = (C.c_int32 * members)(*([0] * members))
    ffi.check(lib.ma_group_create(C.cast(devs, C.c_void_p), members, C.byref(g)))
    try:
        assert lib.ma_group_size(g) == members and lib.ma_group_ctx(g, 0)
        chunks = row_chunks(n, members)
        lens = (C.c_size_t * members)(*[b - a for a, b in chunks])
        pi = (C.c_void_p * members)(*[di.ptr + a * 8 for a, _ in chunks])
        pf = (C.c_void_p * members)(*[df.ptr + a * 8 for a, _ in chunks])
        pm = (C.c_void_p * members)(*([dm.ptr] * members))        # the un-windowed validity buffer ...
        offs = (C.c_size_t * members)(*[a for a, _ in chunks])    # ... plus each chunk's first row as bit offset
        s, c = C.c_int64(), C.c_uint64()
        cast = lambda arr: C.cast(arr, C.c_void_p)  # noqa: E731
        ffi.check(lib.ma_group_sum_i64(g, cast(pi), cast(lens), None, None, C.byref(s), C.byref(c)))
        assert (s.value, c.value) == (int(ints.sum()), n)
        ffi.check(lib.ma_group_sum_i64(g, cast(pi), cast(lens), cast(pm), cast(offs), C.byref(s), C.byref(c)))
        assert (s.value, c.value) == oracle.masked_sum(ints, bits, 0)
        f = C.c_double()
        ffi.check(lib.ma_group_sum_f64(g, cast(pf), cast(lens), None, None, C.byref(f), C.byref(c)))
        exact = math.fsum(flts.tolist())
        assert c.value == n and abs(f.value - exact) <= math.ulp(exact)
        valid = np.unpackbits(bits, bitorder="little")[:n].astype(bool)
        ffi.check(lib.ma_group_sum_f64(g, cast(pf), cast(lens), cast(pm), cast(offs), C.byref(f), C.byref(c)))
        exact = math.fsum(flts[valid].tolist())
        assert c.value == int(valid.sum()) and abs(f.value - exact) <= math.ulp(exact)
    finally:
        lib.ma_group_destroy(g)


def test_device_fold_of_records_equals_the_host_fold(ctx):
    """ma_fold_sum_records (the step after the all-gather) against minarrow_amd.parallel's host fold, bit for bit,
    incl. cancellation between ranks, and against the exact sum."""
    import math

    from minarrow_amd.parallel import fold_dd, fold_int

    rng = np.random.default_rng(4)
    for world in (1, 2, 8, 64):
        rec = np.zeros((world, 8), dtype=np.uint64)
        ints = rng.integers(-2**62, 2**62, size=world, dtype=np.int64)
        his = (rng.standard_normal(world) * 1e17)
        his[::2] *= -1
        los = his * 2.0**-55 * rng.standard_normal(world)
        rec[:, 0] = ints.view(np.uint64)
        rec[:, 1] = rng.integers(0, 2**40, size=world, dtype=np.uint64)
        rec[:, 2] = his.view(np.uint64)
        rec[:, 3] = los.view(np.uint64)
        rec[:, 4] = rng.integers(0, 2**40, size=world, dtype=np.uint64)
        out = np.zeros(4, dtype=np.uint64)
        ctx.fold_sum_records(rec, world, 8, out)
        assert int(out[0]) == fold_int([int(v) for v in rec[:, 0]])
        assert int(out[1]) == int(rec[:, 1].astype(object).sum()) and int(out[3]) == int(rec[:, 4].astype(object).sum())
        want = fold_dd(list(zip(his.tolist(), los.tolist())))
        got = float(out[2:3].view(np.float64)[0])
        assert got == want
        exact = math.fsum(his.tolist() + los.tolist())
        assert abs(got - exact) <= 2 * math.ulp(exact) + abs(exact) * 2**-100
        # device-resident records and output (the async bench path)
        drec, dout = ctx.to_device(rec), ctx.alloc(32)
        ctx.fold_sum_records(drec, world, 8, dout)
        np.testing.assert_array_equal(dout.download(np.uint64, 4), out)
