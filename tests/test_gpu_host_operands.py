"""Host-resident (pageable) operands through the tiled staging pipeline (minarrow_amd/csrc/ma_pipeline.hip): the
elementwise entry points must give the oracle's results — values, validity, divide-by-zero behaviour — exactly as they
do for device-resident columns, whatever mix of host and device operands a call has and wherever the tile seams fall.

The tile is shrunk with ma_ctx_set_staging_tile so that a few hundred thousand rows already span several tiles plus a
ragged last one; the default 32-MiB tile is exercised once at 2^24 rows."""
import numpy as np
import pytest

from minarrow_amd import ffi
from test_gpu_arith import NP, OPS, assert_float_bits_equal, mask_bytes, rand_floats, rand_ints, unpack

pytestmark = pytest.mark.gpu


@pytest.fixture()
def tiled(ctx):
    ctx.set_staging_tile(256 << 10)  # 32768 8-byte rows per tile
    yield ctx
    ctx.set_staging_tile(32 << 20)


def rows_for(tag, tiles=4.4):
    return int((256 << 10) / np.dtype(NP[tag]).itemsize * tiles) + 13


@pytest.mark.parametrize("tag", ["i8", "u16", "i32", "u64", "i64"])
@pytest.mark.parametrize("op", ["add", "multiply", "divide", "floordiv", "power"])
def test_int_host_operands_masked_and_dense(tiled, oracle, tag, op):
    ctx = tiled
    rng = np.random.default_rng(hash((tag, op)) & 0xFFFF)
    n = rows_for(tag)
    lhs = rand_ints(rng, tag, n)
    rhs = rand_ints(rng, tag, n, small=(op == "power"))
    dense_rhs = rhs.copy()
    dense_rhs[dense_rhs == 0] = 3
    st, want, _, _ = oracle.apply_int(oracle.aligned_copy(lhs), oracle.aligned_copy(dense_rhs), op)
    out = np.zeros(n, dtype=NP[tag])
    ctx.apply(tag, lhs, dense_rhs, OPS[op], out, n, n)
    np.testing.assert_array_equal(out, want)
    # masked, validity window at bit offset 5 of a host bitmap; zero divisors become nulls (data-dependent validity)
    rhs[rng.integers(0, n, size=n // 9)] = 0
    bits = rng.integers(0, 256, size=(5 + n + 7) // 8 + 8, dtype=np.uint8)
    window = oracle.pad_bits(np.packbits(unpack(bits, n, 5), bitorder="little"), n)
    st, want, want_mask = oracle.int_body("masked_std", lhs, rhs, op, mask=window)
    out = np.zeros(n, dtype=NP[tag])
    om = np.full(mask_bytes(n) + 8, 0xAA, dtype=np.uint8)
    ctx.apply(tag, lhs, rhs, OPS[op], out, n, n, mask=bits, mask_bit_offset=5, out_mask=om)
    np.testing.assert_array_equal(out, want)
    np.testing.assert_array_equal(om[:mask_bytes(n)], want_mask[:mask_bytes(n)])


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_float_host_operands_all_forms(tiled, oracle, tag):
    ctx = tiled
    rng = np.random.default_rng(77)
    n = rows_for(tag, 3.2)
    a, b, c = (rand_floats(rng, tag, n) for _ in range(3))
    dt = NP[tag]
    for op in ("add", "divide", "remainder"):
        st, want, _, _ = oracle.apply_float(oracle.aligned_copy(a), oracle.aligned_copy(b), op)
        out = np.zeros(n, dtype=dt)
        ctx.apply(tag, a, b, OPS[op], out, n, n)
        assert_float_bits_equal(out, want)
    # fused scalar broadcast == the reference's materialised vec64![x; n] (routing/broadcast.rs:30-45)
    scalar = dt(2.5)
    full = np.full(n, scalar, dtype=dt)
    for side in ("rhs", "lhs"):
        l, r = (a, full) if side == "rhs" else (full, a)
        st, want, _, _ = oracle.apply_float(oracle.aligned_copy(l), oracle.aligned_copy(r), "subtract")
        out = np.zeros(n, dtype=dt)
        ctx.apply_scalar(tag, side, a, n, float(scalar), OPS["subtract"], out)
        assert_float_bits_equal(out, want)
    # FMA, masked
    bits = rng.integers(0, 256, size=n // 8 + 16, dtype=np.uint8)
    window = oracle.pad_bits(np.packbits(unpack(bits, n, 2), bitorder="little"), n)
    st, want, want_mask = oracle.apply_fma(oracle.aligned_copy(a), oracle.aligned_copy(b), oracle.aligned_copy(c), mask=window)
    out = np.zeros(n, dtype=dt)
    om = np.zeros(mask_bytes(n) + 8, dtype=np.uint8)
    ctx.apply_fma(tag, a, b, c, out, n, n, n, mask=bits, mask_bit_offset=2, out_mask=om)
    assert_float_bits_equal(out, want)
    np.testing.assert_array_equal(om[:mask_bytes(n)], want_mask[:mask_bytes(n)])


def test_mixed_host_and_device_operands(tiled, oracle):
    ctx = tiled
    rng = np.random.default_rng(5)
    n = rows_for("i64", 5.5)
    lhs, rhs = rand_ints(rng, "i64", n), rand_ints(rng, "i64", n)
    st, want, _, _ = oracle.apply_int(oracle.aligned_copy(lhs), oracle.aligned_copy(rhs), "subtract")
    d_l, d_r = ctx.to_device(lhs, 64), ctx.to_device(rhs, 64)
    d_o = ctx.alloc(n * 8 + 64)
    # host lhs, device rhs, host out
    out = np.zeros(n, dtype=np.int64)
    ctx.apply("i64", lhs, d_r, OPS["subtract"], out, n, n)
    np.testing.assert_array_equal(out, want)
    # device inputs, host out (only the drain runs)
    out[:] = 0
    ctx.apply("i64", d_l, d_r, OPS["subtract"], out, n, n)
    np.testing.assert_array_equal(out, want)
    # host inputs, device out at an odd 8-byte phase (only the feeder runs; stores are not 16-byte aligned)
    ctx.apply("i64", lhs, rhs, OPS["subtract"], d_o.offset(8), n, n)
    np.testing.assert_array_equal(d_o.download(np.int64, n, 8), want)


def test_dense_divide_by_zero_is_reported_from_any_tile(tiled):
    ctx = tiled
    n = rows_for("i32", 4.0)
    lhs = np.arange(n, dtype=np.int32)
    rhs = np.ones(n, dtype=np.int32)
    out = np.zeros(n, dtype=np.int32)
    ctx.apply("i32", lhs, rhs, OPS["divide"], out, n, n)
    np.testing.assert_array_equal(out, lhs)
    rhs[n - 7] = 0  # in the ragged last tile
    with pytest.raises(ffi.MinarrowHipError) as e:
        ctx.apply("i32", lhs, rhs, OPS["divide"], out, n, n)
    assert e.value.status == ffi.MA_ERR_DIVIDE_BY_ZERO
    rhs[n - 7] = 1
    ctx.apply("i32", lhs, rhs, OPS["remainder"], out, n, n)  # the latch was cleared
    assert not out.any()


def test_tiled_and_whole_operand_staging_agree_at_default_tile(ctx, oracle):
    rng = np.random.default_rng(9)
    n = (1 << 24) + 4099  # 4 tiles of 32 MiB
    a, b = rng.standard_normal(n), rng.standard_normal(n)
    tiled_out, whole_out = np.zeros(n), np.zeros(n)
    ctx.apply("f64", a, b, OPS["multiply"], tiled_out, n, n)
    ctx.set_staging_tile(0)
    try:
        ctx.apply("f64", a, b, OPS["multiply"], whole_out, n, n)
    finally:
        ctx.set_staging_tile(32 << 20)
    np.testing.assert_array_equal(tiled_out, whole_out)
    np.testing.assert_array_equal(tiled_out, a * b)


@pytest.mark.parametrize("fl", [np.float64, np.float32])
@pytest.mark.parametrize("int_side", ["lhs", "rhs"])
def test_promote_host_operands(tiled, oracle, fl, int_side):
    """Int32 <-> Float promotion (src/kernels/routing/arithmetic.rs:244-269): operands of different widths share the
    ring; results equal cast-then-apply_float, as the reference computes them."""
    ctx = tiled
    rng = np.random.default_rng(31)
    ftag = "f64" if fl == np.float64 else "f32"
    n = int((256 << 10) / np.dtype(fl).itemsize * 3.3) + 5
    ints = rng.integers(-(1 << 31), (1 << 31) - 1, size=n, dtype=np.int32)
    flts = (rng.standard_normal(n) * 1e3).astype(fl)
    lhs, rhs = (ints, flts) if int_side == "lhs" else (flts, ints)
    ltag, rtag = ("i32", ftag) if int_side == "lhs" else (ftag, "i32")
    st, want, _, _ = oracle.apply_float(oracle.aligned_copy(lhs.astype(fl)), oracle.aligned_copy(rhs.astype(fl)), "multiply")
    out = np.zeros(n, dtype=fl)
    ctx.apply_promote(ltag, rtag, lhs, rhs, OPS["multiply"], out, n, n)
    assert_float_bits_equal(out, want)
    bits = rng.integers(0, 256, size=n // 8 + 16, dtype=np.uint8)
    window = oracle.pad_bits(np.packbits(unpack(bits, n, 5), bitorder="little"), n)
    st, want, want_mask = oracle.float_body("masked_std", lhs.astype(fl), rhs.astype(fl), "add", mask=window)
    out = np.zeros(n, dtype=fl)
    om = np.zeros(mask_bytes(n) + 8, dtype=np.uint8)
    ctx.apply_promote(ltag, rtag, lhs, rhs, OPS["add"], out, n, n, mask=bits, mask_bit_offset=5, out_mask=om)
    assert_float_bits_equal(out, want)
    np.testing.assert_array_equal(om[:mask_bytes(n)], want_mask[:mask_bytes(n)])


@pytest.mark.parametrize("async_mode", [False, True])
def test_pinned_vec64_operands(tiled, oracle, async_mode):
    """ma_alloc64_pinned memory (the Vec64 stand-in): through the ring in synchronous mode, addressed in place by the
    kernels in async mode — same bits either way."""
    from minarrow_amd.host import PinnedBuffer

    ctx = tiled
    rng = np.random.default_rng(41)
    n = rows_for("f64", 3.7)
    a, b = rand_floats(rng, "f64", n), rand_floats(rng, "f64", n)
    st, want, _, _ = oracle.apply_float(oracle.aligned_copy(a), oracle.aligned_copy(b), "divide")
    pa_, pb_, po_ = (PinnedBuffer(n * 8) for _ in range(3))
    pa_.view(np.float64, n)[:] = a
    pb_.view(np.float64, n)[:] = b
    ctx.set_async(async_mode)
    try:
        ctx.apply("f64", pa_, pb_, OPS["divide"], po_, n, n)
        ctx.synchronize()
    finally:
        ctx.set_async(False)
    assert_float_bits_equal(po_.view(np.float64, n).copy(), want)
    for buf in (pa_, pb_, po_):
        buf.free()


@pytest.mark.parametrize("tag", ["i64", "u32", "f64", "f32"])
def test_host_resident_column_sums_in_tiles(tiled, oracle, tag):
    """Sum / valid count / mean of a pageable column: tile k's scan writes record k, records fold in tile order.
    Integers bit-exact against the oracle; floats within 1 ULP of the exactly rounded sum, like resident columns."""
    import math

    ctx = tiled
    rng = np.random.default_rng(17)
    n = rows_for(tag, 5.3)
    dt = NP[tag]
    if tag in ("f64", "f32"):
        data = (rng.standard_normal(n) * 10.0 ** rng.integers(0, 9, size=n)).astype(dt)
    else:
        data = rand_ints(rng, tag, n)
    bits = rng.integers(0, 256, size=(n + 11 + 7) // 8 + 16, dtype=np.uint8)
    valid = unpack(bits, n, 11)
    for mask, off, sel in ((None, 0, data), (bits, 11, data[valid])):
        s, c = ctx.sum(tag, data, n, mask=mask, mask_bit_offset=off)
        m, c2 = ctx.mean(tag, data, n, mask=mask, mask_bit_offset=off)
        assert c == c2 == sel.size
        if tag in ("f64", "f32"):
            exact = math.fsum(sel.astype(np.float64).tolist())
            assert abs(s - exact) <= math.ulp(exact)
            hi, lo, c3 = ctx.sum_dd(tag, data, n, mask=mask, mask_bit_offset=off)
            assert hi == s and c3 == c and abs(lo) <= math.ulp(hi)
            assert m == s / c
        else:
            want = int(sel.astype(object).sum())
            wrap = ((want + (1 << 63)) % (1 << 64)) - (1 << 63) if tag == "i64" else want % (1 << 64)
            assert s == wrap
            assert m == float(s) / c
    # same column resident in HBM: identical integer bits, float within the same bound
    d = ctx.to_device(data, 64)
    s_dev, _ = ctx.sum(tag, d, n)
    s_host, _ = ctx.sum(tag, data, n)
    if tag in ("f64", "f32"):
        assert abs(s_dev - s_host) <= math.ulp(s_dev)
    else:
        assert s_dev == s_host


def test_registering_an_existing_host_buffer(ctx, oracle):
    """ma_host_register pins a buffer the host already owns: it then classifies as pinned (kernels address it in place)
    and gives the same bits; after ma_host_unregister it is ordinary pageable memory again."""
    from minarrow_amd.host import Registered

    lib = ctx.lib
    rng = np.random.default_rng(3)
    n = 300_001
    a, b = rand_ints(rng, "i64", n), rand_ints(rng, "i64", n)
    out = np.zeros(n, dtype=np.int64)
    st, want, _, _ = oracle.apply_int(oracle.aligned_copy(a), oracle.aligned_copy(b), "add")
    assert lib.ma_pointer_kind(a.ctypes.data) == 0
    with Registered(a) as ra, Registered(b) as rb, Registered(out) as ro:
        assert [lib.ma_pointer_kind(x.ptr + 8 * 1000) for x in (ra, rb, ro)] == [1, 1, 1]
        ctx.set_async(True)  # async: pinned operands are addressed in place by the kernels
        try:
            ctx.apply("i64", ra, rb, OPS["add"], ro, n, n)
            ctx.synchronize()
        finally:
            ctx.set_async(False)
        np.testing.assert_array_equal(out, want)
        assert ctx.sum("i64", ra, n)[0] == int(a.astype(object).sum()) % (1 << 64) - (1 << 64) * (int(a.astype(object).sum()) % (1 << 64) >= (1 << 63))
    assert lib.ma_pointer_kind(a.ctypes.data) == 0
    with pytest.raises(ffi.MinarrowHipError):
        ffi.check(lib.ma_host_register(None, 0))


def test_pinned_allocator_recycles_large_blocks(ctx):
    """ma_alloc64_pinned / ma_free_pinned: blocks come back from a size-class cache (pinning pages is the expensive part
    of a pinned Vec64; from 4 KiB on since round 3: the result slab of an 8192-row record batch is ~130 KiB);
    ma_pinned_pool_trim empties it."""
    import ctypes as C

    lib = ctx.lib
    p1, p2, small = C.c_void_p(), C.c_void_p(), C.c_void_p()
    ffi.check(lib.ma_pinned_pool_trim(2 << 30))
    ffi.check(lib.ma_alloc64_pinned((4 << 20) + 5, C.byref(p1)))  # rounded up to 4.5 MiB (eight steps per power of two)
    assert p1.value % 64 == 0 and lib.ma_pointer_kind(p1.value) == 1
    C.memset(p1.value, 0x5A, (4 << 20) + 5)
    ffi.check(lib.ma_free_pinned(p1.value))
    ffi.check(lib.ma_alloc64_pinned((4 << 20) + 4096, C.byref(p2)))  # same 4.5-MiB class: the parked block
    assert p2.value == p1.value
    arr = np.ctypeslib.as_array(C.cast(p2.value, C.POINTER(C.c_int64)), shape=(1 << 19,))
    arr[:] = np.arange(1 << 19)
    assert ctx.sum("i64", p2.value, 1 << 19) == ((1 << 19) * ((1 << 19) - 1) // 2, 1 << 19)
    ffi.check(lib.ma_free_pinned(p2.value))
    ffi.check(lib.ma_pinned_pool_trim(0))       # nothing stays cached ...
    ffi.check(lib.ma_alloc64_pinned(4 << 20, C.byref(p2)))
    ffi.check(lib.ma_free_pinned(p2.value))     # ... and with a zero limit nothing is parked again
    ffi.check(lib.ma_pinned_pool_trim(2 << 30))
    ffi.check(lib.ma_alloc64_pinned(1000, C.byref(small)))  # the smallest class: 4 KiB
    first_small = small.value
    ffi.check(lib.ma_free_pinned(small.value))
    ffi.check(lib.ma_alloc64_pinned(3000, C.byref(small)))
    assert small.value == first_small
    ffi.check(lib.ma_free_pinned(small.value))
    ffi.check(lib.ma_alloc64_pinned(140_000, C.byref(small)))  # the 144-KiB class (128 KiB + 16 KiB)
    first_small = small.value
    ffi.check(lib.ma_free_pinned(small.value))
    ffi.check(lib.ma_alloc64_pinned(145_000, C.byref(small)))
    assert small.value == first_small
    ffi.check(lib.ma_free_pinned(small.value))


def test_device_blocks_are_recycled(ctx):
    """ma_dev_free parks blocks of 1 MiB and more per device (hipFree would stall every stream of the device);
    ma_dev_alloc hands them out again; ma_dev_pool_trim releases them."""
    import ctypes as C

    lib = ctx.lib
    ffi.check(lib.ma_dev_pool_trim(ctx.handle, 16 << 30))
    a = ctx.alloc((8 << 20) + 3)
    first = a.ptr
    a.upload(np.arange(1 << 20, dtype=np.int64))
    a.free()
    b = ctx.alloc((8 << 20) + 1000)  # same 9-MiB size step
    assert b.ptr == first
    b.upload(np.arange(1 << 20, dtype=np.int64)[::-1].copy())
    assert ctx.sum("i64", b, 1 << 20) == ((1 << 20) * ((1 << 20) - 1) // 2, 1 << 20)
    b.free()
    ffi.check(lib.ma_dev_pool_trim(ctx.handle, 0))
    c = ctx.alloc((8 << 20) + 1000)
    c.free()  # limit 0: released, not parked
    ffi.check(lib.ma_dev_pool_trim(ctx.handle, 16 << 30))
    small = ctx.alloc(4096)
    small.free()


def test_double_free_of_a_parked_block_is_refused(ctx):
    """A block that sits in the pinned / device cache must not be freed again: round 1 would have handed the same
    memory out twice (pinned) or hipFree'd a block still listed as parked. Now MA_ERR_INVALID_ARGUMENT, and the block
    stays usable for the next allocation."""
    import ctypes as C

    lib = ctx.lib
    p = C.c_void_p()
    ffi.check(lib.ma_alloc64_pinned(3 << 20, C.byref(p)))
    ffi.check(lib.ma_free_pinned(p.value))
    assert lib.ma_free_pinned(p.value) == ffi.MA_ERR_INVALID_ARGUMENT
    assert b"double free" in lib.ma_last_error_string()
    q = C.c_void_p()
    ffi.check(lib.ma_alloc64_pinned(3 << 20, C.byref(q)))
    assert q.value == p.value  # the parked block, once
    ffi.check(lib.ma_free_pinned(q.value))
    d = C.c_void_p()
    ffi.check(lib.ma_dev_alloc(ctx.handle, 5 << 20, C.byref(d)))
    ffi.check(lib.ma_dev_free(ctx.handle, d.value))
    assert lib.ma_dev_free(ctx.handle, d.value) == ffi.MA_ERR_INVALID_ARGUMENT


def test_pool_limits_can_be_restored_after_a_trim(ctx):
    """trim(keep) lowers the cache limit for good; *_pool_set_limit raises it again without releasing anything."""
    import ctypes as C

    lib = ctx.lib
    ffi.check(lib.ma_dev_pool_trim(ctx.handle, 0))
    a = ctx.alloc(6 << 20)
    first = a.ptr
    a.free()  # limit 0: released to the runtime
    ffi.check(lib.ma_dev_pool_set_limit(ctx.handle, 16 << 30))
    b = ctx.alloc(6 << 20)
    second = b.ptr
    b.free()  # parked again
    c = ctx.alloc(6 << 20)
    assert c.ptr == second
    c.free()
    ffi.check(lib.ma_pinned_pool_trim(0))
    ffi.check(lib.ma_pinned_pool_set_limit(2 << 30))
    p, q = C.c_void_p(), C.c_void_p()
    ffi.check(lib.ma_alloc64_pinned(2 << 20, C.byref(p)))
    ffi.check(lib.ma_free_pinned(p.value))
    ffi.check(lib.ma_alloc64_pinned(2 << 20, C.byref(q)))
    assert q.value == p.value
    ffi.check(lib.ma_free_pinned(q.value))
    del first


def test_output_allocator_picks_by_measured_write_rate(ctx):
    """ma_dev_alloc_output: blocks of 256 MiB and more are chosen among a few candidates by their measured write rate
    (DESIGN.md §3.4); the rejected candidates are parked and handed out by the next allocations; small blocks take the
    plain path. The block must be usable like any other."""
    n = 40_000_000  # 320 MB
    plain = ctx.alloc_output(n * 8)  # the search is opt-in (round 4): by default this is ma_dev_alloc, nothing is measured
    assert plain.write_gbps == 0.0 and plain.alloc_stats["blocks_measured"] == 0
    plain.free()
    before = ctx.lib.ma_dev_output_search(1)
    assert before == 0 and ctx.lib.ma_dev_output_search(-1) == 1
    out = ctx.alloc_output(n * 8)
    assert 1000.0 < out.write_gbps < 8000.0  # measured, and a plausible HBM figure
    a = ctx.alloc(n * 8)                     # a parked candidate when any was rejected, else a fresh block
    ctx.synth_iota("i64", a, n, 5)
    ctx.apply_scalar("i64", "rhs", a, n, 3, 0, out)  # out = a + 3
    assert ctx.sum("i64", out, n) == (n * (n - 1) // 2 + 8 * n, n)
    first = out.ptr
    out.free()
    again = ctx.alloc_output(n * 8)  # the block cache now holds measured blocks: no probe needed to choose among them
    assert again.write_gbps >= out.write_gbps * 0.97 or again.ptr == first
    small = ctx.alloc_output(4096)
    assert small.write_gbps == 0.0
    for b in (a, again, small):
        b.free()
    assert ctx.lib.ma_dev_output_search(0) == 1
