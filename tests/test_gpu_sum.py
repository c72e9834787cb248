"""GPU parity tests for the sum / valid-count / mean reductions, called through the C ABI.

Oracle: oracle/minarrow_oracle.c (restatement of benches/benchmark_parallel_simd.rs:44-98,
benches/hotloop_benchmark_simd.rs:56-174, benches/hotloop_benchmark_std.rs:49-57).
Bar: integers and counts bit-exact; f32/f64 within 1 ULP of the exactly rounded sum (math.fsum).
"""
import math

import numpy as np
import pytest

from minarrow_amd import ffi
from minarrow_amd.host import Context, tuning_build

pytestmark = pytest.mark.gpu

SIZES = [0, 1, 2, 3, 63, 64, 65, 127, 1000, 4095, 4096, 4097, 8191, 8192, 12345, (1 << 20) + 37]
ULP_TOL = 1  # f32/f64 sums: distance from the exactly rounded sum, in ULPs of that sum


def wrap_i64(x: int) -> int:
    x &= (1 << 64) - 1
    return x - (1 << 64) if x >= (1 << 63) else x


def ulps(a: float, b: float) -> float:
    if a == b:
        return 0.0
    return abs(a - b) / math.ulp(b)


def unpack(bits: np.ndarray, off: int, n: int) -> np.ndarray:
    return np.unpackbits(bits, bitorder="little")[off:off + n].astype(bool)


# ---- closed forms of the reference's bench inputs (the only "golden" sums it implies) -----------------

@pytest.mark.parametrize("n", [1000, 1_000_000])
def test_iota_closed_form(ctx, oracle, n):
    """benches/hotloop_benchmark_std.rs:49 (n = 1000) and BASELINE config 1 (n = 10^6): sum(0..n)."""
    expect = n * (n - 1) // 2
    a = np.arange(n, dtype=np.int64)
    assert oracle.sum_scalar(a) == expect
    assert oracle.chunked_sum(a) == expect
    d = ctx.to_device(a)
    assert ctx.sum("i64", d, n) == (expect, n)
    f = a.astype(np.float64)
    df = ctx.to_device(f)
    got, cnt = ctx.sum("f64", df, n)
    assert cnt == n and got == float(expect)  # every partial sum < 2^53: exact in any order
    assert oracle.simd_sum(f, 4) == float(expect)
    mean, cnt = ctx.mean("f64", df, n)
    assert mean == float(expect) / n


def test_iota_generated_on_device(ctx):
    n = 3_000_001
    buf = ctx.alloc(n * 8)
    ctx.synth_iota("i64", buf, n, 5)
    assert ctx.sum("i64", buf, n) == (n * (n - 1) // 2 + 5 * n, n)
    np.testing.assert_array_equal(buf.download(np.int64, 10), np.arange(5, 15))
    ctx.synth_iota("f64", buf, n, 5)
    got, _ = ctx.sum("f64", buf, n)
    assert got == float(n * (n - 1) // 2 + 5 * n)


# ---- dense, all types, ragged sizes ---------------------------------------------------------------------

@pytest.mark.parametrize("n", SIZES)
def test_dense_int_sums(ctx, oracle, n):
    rng = np.random.default_rng(n + 1)
    a = rng.integers(-(1 << 63), (1 << 63) - 1, size=n, dtype=np.int64)
    d = ctx.to_device(a, pad_bytes=64)
    expect = wrap_i64(int(a.astype(object).sum())) if n else 0
    assert oracle.sum_scalar(a) == expect
    assert oracle.simd_sum(a, 4) == expect and oracle.chunked_sum(a, 1 << 10, 4) == expect
    assert ctx.sum("i64", d, n) == (expect, n)
    u = a.view(np.uint64)
    assert ctx.sum("u64", d, n) == (expect & ((1 << 64) - 1), n)
    a32 = rng.integers(-(1 << 31), (1 << 31) - 1, size=n, dtype=np.int32)
    d32 = ctx.to_device(a32, pad_bytes=64)
    assert ctx.sum("i32", d32, n) == (int(a32.astype(np.int64).sum()), n)
    assert ctx.sum("u32", d32, n) == (int(a32.view(np.uint32).astype(np.uint64).sum()), n)
    assert u.size == n


@pytest.mark.parametrize("n", SIZES)
def test_dense_float_sums(ctx, oracle, n):
    rng = np.random.default_rng(n + 2)
    a = (rng.standard_normal(n) * 10.0 ** rng.integers(-3, 12, size=n)).astype(np.float64)
    d = ctx.to_device(a, pad_bytes=64)
    exact = math.fsum(a.tolist())
    got, cnt = ctx.sum("f64", d, n)
    assert cnt == n
    assert ulps(got, exact) <= ULP_TOL if exact != 0 else abs(got) <= 1e-300
    hi, lo, cnt = ctx.sum_dd("f64", d, n)
    assert hi == got and abs(lo) <= math.ulp(hi)
    a32 = a.astype(np.float32)
    a32[~np.isfinite(a32)] = 0
    d32 = ctx.to_device(a32, pad_bytes=64)
    exact32 = math.fsum(a32.astype(np.float64).tolist())
    got32, cnt = ctx.sum("f32", d32, n)
    assert cnt == n and (ulps(got32, exact32) <= ULP_TOL if exact32 != 0 else got32 == 0)
    if n:
        mean, _ = ctx.mean("f64", d, n)
        assert ulps(mean, exact / n) <= 2
    else:
        mean, cnt = ctx.mean("f64", d, 0)
        assert math.isnan(mean) and cnt == 0


def test_f64_cancellation_beats_naive(ctx, oracle):
    """Ill-conditioned input: the reference-order sum drifts, the GPU result stays within 1 ULP of exact."""
    n = 1 << 20
    rng = np.random.default_rng(99)
    a = rng.standard_normal(n) * 1e15
    a = np.concatenate([a, -a, rng.standard_normal(1000)])
    rng.shuffle(a)
    exact = math.fsum(a.tolist())
    got, _ = ctx.sum("f64", ctx.to_device(a), a.size)
    # Condition number ~1e20: beyond what any fixed-precision accumulator can hold to 1 ULP. The
    # double-double bound is |err| <= ulp(S) + n * eps^2 * sum|x| ("f64 sum accuracy").
    eps = 2.0 ** -53
    bound = math.ulp(exact) + a.size * eps * eps * float(np.abs(a).sum())
    assert abs(got - exact) <= bound
    # ... and it is many orders of magnitude closer than the reference's summation order gets
    ref_order = oracle.chunked_sum(a, 1 << 20, 4)
    assert abs(got - exact) * 1e6 < abs(ref_order - exact) or abs(ref_order - exact) == 0


def test_non_finite_values(ctx):
    a = np.array([1.0, np.inf, 2.0, 3.0] * 3000)
    got, _ = ctx.sum("f64", ctx.to_device(a), a.size)
    assert got == np.inf
    a[5] = -np.inf
    got, _ = ctx.sum("f64", ctx.to_device(a), a.size)
    assert math.isnan(got)
    a = np.array([1.0, np.nan] * 5000)
    got, _ = ctx.sum("f64", ctx.to_device(a), a.size)
    assert math.isnan(got)
    a = np.full(20000, 1.7e308)
    got, _ = ctx.sum("f64", ctx.to_device(a), a.size)
    assert got == np.inf  # overflow behaves like a plain IEEE sum


# ---- where the buffer lives -----------------------------------------------------------------------------

def test_pageable_pinned_and_device_agree(ctx):
    from minarrow_amd.host import PinnedBuffer

    n = 200_003
    rng = np.random.default_rng(5)
    a = rng.integers(-(1 << 40), 1 << 40, size=n, dtype=np.int64)
    expect = (int(a.sum()), n)
    assert ctx.sum("i64", a, n) == expect  # pageable numpy memory: staged by the library
    pin = PinnedBuffer(n * 8)              # the Vec64 stand-in: kernels read it in place
    assert pin.ptr % 64 == 0
    pin.view(np.int64, n)[:] = a
    assert ctx.sum("i64", pin, n) == expect
    assert ctx.sum("i64", ctx.to_device(a), n) == expect
    assert ctx.lib.ma_pointer_kind(a.ctypes.data) == 0
    assert ctx.lib.ma_pointer_kind(pin.ptr) == 1
    pin.free()


@pytest.mark.parametrize("shift", [1, 2, 3])
def test_unaligned_window(ctx, shift):
    """A view with a non-zero offset hands over data + offset: only element-aligned (src/kernels/routing/
    arithmetic.rs:284-285 slices by view offset)."""
    n = 100_000
    rng = np.random.default_rng(shift)
    a = rng.integers(-(1 << 50), 1 << 50, size=n + 8, dtype=np.int64)
    d = ctx.to_device(a)
    assert ctx.sum("i64", d.offset(shift * 8), n) == (int(a[shift:shift + n].sum()), n)
    f = rng.standard_normal(n + 8).astype(np.float32)
    d32 = ctx.to_device(f)
    got, _ = ctx.sum("f32", d32.offset(shift * 4), n)
    exact = math.fsum(f[shift:shift + n].astype(np.float64).tolist())
    assert ulps(got, exact) <= ULP_TOL


def test_misaligned_pointer_rejected(ctx):
    from minarrow_amd import ffi

    a = np.zeros(64, dtype=np.int64)
    d = ctx.to_device(a)
    with pytest.raises(ffi.MinarrowHipError) as e:
        ctx.sum("i64", d.offset(4), 8)
    assert e.value.status == ffi.MA_ERR_INVALID_ARGUMENT


# ---- Bitmask-gated sums (build-defined semantics; oracle = scalar loop over set bits) ----------------------

@pytest.mark.parametrize("n", [1, 5, 63, 64, 65, 200, 4096, 4097, 9000, 100_003, (1 << 20) + 37])
@pytest.mark.parametrize("bit_off", [0, 1, 3, 7, 8, 63, 64, 65, 130])
def test_masked_i64(ctx, oracle, n, bit_off):
    rng = np.random.default_rng(n * 131 + bit_off)
    a = rng.integers(-(1 << 63), (1 << 63) - 1, size=n, dtype=np.int64)
    bits = rng.integers(0, 256, size=(bit_off + n + 7) // 8 + 8, dtype=np.uint8)
    expect = oracle.masked_sum(a, bits, bit_off)
    valid = unpack(bits, bit_off, n)
    assert expect[1] == int(valid.sum())
    d, m = ctx.to_device(a, pad_bytes=64), ctx.to_device(bits, pad_bytes=16)
    assert ctx.sum("i64", d, n, mask=m, mask_bit_offset=bit_off) == expect
    # pageable mask + pageable data take the staging path
    assert ctx.sum("i64", a, n, mask=bits, mask_bit_offset=bit_off) == expect


@pytest.mark.parametrize("n", [7, 64, 4097, 70_001])
@pytest.mark.parametrize("bit_off", [0, 5, 64, 77])
def test_masked_other_types(ctx, oracle, n, bit_off):
    rng = np.random.default_rng(n * 7 + bit_off)
    bits = rng.integers(0, 256, size=(bit_off + n + 7) // 8 + 8, dtype=np.uint8)
    valid = unpack(bits, bit_off, n)
    m = ctx.to_device(bits, pad_bytes=16)
    a32 = rng.integers(-(1 << 31), (1 << 31) - 1, size=n, dtype=np.int32)
    assert ctx.sum("i32", ctx.to_device(a32, 64), n, mask=m, mask_bit_offset=bit_off) == oracle.masked_sum(a32, bits, bit_off)
    u32 = a32.view(np.uint32)
    assert ctx.sum("u32", ctx.to_device(u32, 64), n, mask=m, mask_bit_offset=bit_off) == (
        int(u32[valid].astype(np.uint64).sum()), int(valid.sum()))
    f = rng.standard_normal(n) * 1e8
    # null slots may hold anything, including NaN: they must be ignored, not multiplied by zero
    f[~valid] = np.nan
    got, cnt = ctx.sum("f64", ctx.to_device(f, 64), n, mask=m, mask_bit_offset=bit_off)
    exact = math.fsum(f[valid].tolist())
    assert cnt == int(valid.sum()) and (ulps(got, exact) <= ULP_TOL if exact else got == 0)
    ref_sum, ref_cnt = oracle.masked_sum(np.nan_to_num(f, nan=0.0), bits, bit_off)
    assert ref_cnt == cnt and abs(ref_sum - exact) <= 1e-6 * max(1.0, abs(exact))
    f32 = (rng.standard_normal(n) * 100).astype(np.float32)
    got, cnt = ctx.sum("f32", ctx.to_device(f32, 64), n, mask=m, mask_bit_offset=bit_off)
    exact = math.fsum(f32[valid].astype(np.float64).tolist())
    assert cnt == int(valid.sum()) and (ulps(got, exact) <= ULP_TOL if exact else got == 0)
    mean, cnt = ctx.mean("f32", ctx.to_device(f32, 64), n, mask=m, mask_bit_offset=bit_off)
    if cnt:
        assert ulps(mean, exact / cnt) <= 2
    else:
        assert math.isnan(mean)


def test_mask_edge_patterns(ctx, oracle):
    n = 50_000
    a = np.arange(1, n + 1, dtype=np.int64)
    d = ctx.to_device(a)
    all_set = np.full(n // 8 + 16, 0xFF, dtype=np.uint8)
    none_set = np.zeros(n // 8 + 16, dtype=np.uint8)
    assert ctx.sum("i64", d, n, mask=ctx.to_device(all_set)) == (int(a.sum()), n)
    assert ctx.sum("i64", d, n, mask=ctx.to_device(none_set)) == (0, 0)
    mean, cnt = ctx.mean("i64", d, n, mask=ctx.to_device(none_set))
    assert math.isnan(mean) and cnt == 0
    # cached null count of 0 => the dense kernel is taken and the mask is never read
    # (reference gate: src/kernels/arithmetic/simd.rs:144,454)
    assert ctx.sum("i64", d, n, mask=ctx.to_device(none_set), null_count=0) == (int(a.sum()), n)
    alt = np.full(n // 8 + 16, 0x55, dtype=np.uint8)
    assert ctx.sum("i64", d, n, mask=ctx.to_device(alt)) == oracle.masked_sum(a, alt, 0)


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9])
@pytest.mark.parametrize("bpc", [0, 1, 8])
def test_kernel_variants_agree(ctx, oracle, variant, bpc):
    """Unroll / non-temporal / grid-size variants are the same function. The unroll / load-kind bits are TUNING forms: they exist in
    the tuning build only (MINARROW_HIP_LIB=build/tuning/libminarrow_hip.so); the shipped library refuses them — which is checked
    here — and runs the grid sizes on its one shape."""
    if variant and not tuning_build():
        with pytest.raises(ffi.MinarrowHipError) as e:
            ctx.set_variant(variant)
        assert e.value.status == ffi.MA_ERR_UNSUPPORTED and "TUNING=1" in str(e.value)
        return
    n = 3_000_017
    rng = np.random.default_rng(11)
    a = rng.integers(-(1 << 62), 1 << 62, size=n, dtype=np.int64)
    f = rng.standard_normal(n) * 1e3
    bits = rng.integers(0, 256, size=n // 8 + 16, dtype=np.uint8)
    d, df, m = ctx.to_device(a, 64), ctx.to_device(f, 64), ctx.to_device(bits, 16)
    try:
        ctx.set_variant(variant)
        ctx.set_blocks_per_cu(bpc)
        assert ctx.sum("i64", d, n) == (oracle.sum_scalar(a), n)
        assert ctx.sum("i64", d, n, mask=m, mask_bit_offset=9) == oracle.masked_sum(a, bits, 9)
        got, _ = ctx.sum("f64", df, n)
        assert ulps(got, math.fsum(f.tolist())) <= ULP_TOL
    finally:
        ctx.set_variant(0)
        ctx.set_blocks_per_cu(0)


def test_repeatable_and_async(ctx):
    """Same grid => same bits, run after run; async mode writes results to device-reachable slots."""
    from minarrow_amd.host import PinnedBuffer

    n = 5_000_011
    buf = ctx.alloc(n * 8)
    ctx.synth_splitmix("f64", buf, n, seed=42)
    first = ctx.sum_dd("f64", buf, n)
    for _ in range(5):
        assert ctx.sum_dd("f64", buf, n) == first
    slot = PinnedBuffer(64)
    view = slot.view(np.float64, 8)
    view[:] = 0
    ctx.set_async(True)
    try:
        for _ in range(3):
            ctx.sum_into("f64", buf, n, out_sum=slot.ptr, out_count=slot.ptr + 16, dd_lo=slot.ptr + 8)
        ctx.synchronize()
    finally:
        ctx.set_async(False)
    assert (view[0], view[1], int(slot.view(np.uint64, 8)[2])) == first
    slot.free()


# ---- BASELINE sizes: 1 B rows (config 2 / config 4 shape), size-independent properties ---------------------

@pytest.mark.big
def test_one_billion_rows(ctx):
    """n = 10^9 (benches/benchmark_parallel_simd.rs:39). i64: closed form, bit exact. f64: within 1 ULP
    (= 64 at this magnitude) of the exactly rounded closed form. Checksum-of-checksums: eight 125 M-row
    chunk sums (the 8-GPU row partition of config 4) add up to the whole, also under a 10 % null mask."""
    n = 1_000_000_000
    expect = n * (n - 1) // 2
    buf = ctx.alloc(n * 8)
    ctx.synth_iota("i64", buf, n, 0)
    assert ctx.sum("i64", buf, n) == (expect, n)
    chunk = n // 8
    parts = [ctx.sum("i64", buf.offset(g * chunk * 8), chunk) for g in range(8)]
    assert sum(p[0] for p in parts) == expect and sum(p[1] for p in parts) == n
    mask = ctx.alloc(n // 8 + 64)
    ctx.synth_validity(mask, n, seed=0xC0FFEE, null_every=10)
    total, cnt = ctx.sum("i64", buf, n, mask=mask)
    assert 0.899 * n < cnt < 0.901 * n
    mparts = [ctx.sum("i64", buf.offset(g * chunk * 8), chunk, mask=mask, mask_bit_offset=g * chunk) for g in range(8)]
    assert wrap_i64(sum(p[0] for p in mparts)) == total and sum(p[1] for p in mparts) == cnt
    ctx.synth_iota("f64", buf, n, 0)
    got, c = ctx.sum("f64", buf, n)
    assert c == n and abs(got - float(expect)) <= math.ulp(float(expect))
    mean, _ = ctx.mean("f64", buf, n)
    assert abs(mean - 499_999_999.5) <= math.ulp(499_999_999.5)
    mask.free()
    buf.free()


@pytest.mark.big
def test_more_than_2_to_32_rows(ctx):
    """Row indices are 64-bit end to end: 2^32 + 300 000 007 i32 rows (18.4 GB). data[i] = wrap_i32(i); one full
    cycle of i32 values sums to -2^31, the remainder is a plain arithmetic series."""
    r = 300_000_007
    n = (1 << 32) + r
    buf = ctx.alloc(n * 4)
    ctx.synth_iota("i32", buf, n, 0)
    expect = -(1 << 31) + r * (r - 1) // 2
    assert ctx.sum("i32", buf, n) == (expect, n)
    # the tail window beyond the 2^32nd row
    assert ctx.sum("i32", buf.offset((1 << 32) * 4), r) == (r * (r - 1) // 2, r)
    buf.free()


def test_byte_misaligned_mask_pointer(ctx, oracle):
    """An Arrow validity buffer may start anywhere: the pointer is re-based onto its enclosing 8-byte word and the
    byte delta folded into the bit offset (device and pageable memory alike)."""
    rng = np.random.default_rng(77)
    n = 70_001
    a = rng.integers(-(1 << 50), 1 << 50, size=n, dtype=np.int64)
    bits = rng.integers(0, 256, size=n // 8 + 64, dtype=np.uint8)
    d, m = ctx.to_device(a, 64), ctx.to_device(bits, 16)
    for byte_delta, bit_off in ((1, 0), (3, 5), (7, 63), (13, 2)):
        expect = oracle.masked_sum(a, bits, byte_delta * 8 + bit_off)
        assert ctx.sum("i64", d, n, mask=m.offset(byte_delta), mask_bit_offset=bit_off) == expect
        view = bits[byte_delta:]  # a numpy view: pageable memory at an odd address
        assert ctx.sum("i64", a, n, mask=view, mask_bit_offset=bit_off) == expect


@pytest.mark.parametrize("n", [1, 4097, (1 << 20) + 37, (1 << 24) + 5])
def test_both_publish_forms_of_the_reduction_agree(ctx, oracle, n, monkeypatch):
    """The single-launch reduction publishes per-workgroup partials with write-through stores + sharded arrival tickets
    (default) or with agent-scope release / acquire fences (MINARROW_HIP_FENCED_REDUCE=1 when the context is created): identical
    results, for grids below and above the sharding threshold, dense and masked."""
    rng = np.random.default_rng(n)
    a = rng.integers(-(1 << 62), 1 << 62, size=n, dtype=np.int64)
    f = rng.standard_normal(n) * 1e3
    bits = rng.integers(0, 256, size=n // 8 + 16, dtype=np.uint8)
    da, df, dm = ctx.to_device(a, 64), ctx.to_device(f, 64), ctx.to_device(bits, 16)
    got = {}
    try:
        monkeypatch.setenv("MINARROW_HIP_FENCED_REDUCE", "1")  # read when a context is created
        fenced = Context(0)
        monkeypatch.delenv("MINARROW_HIP_FENCED_REDUCE")
        for variant, c in ((0, ctx), (256, fenced)):
            for grid in (0, 3, 97, 2048):
                c.set_grid(grid)
                got[(variant, grid)] = (c.sum("i64", da, n), c.sum("i64", da, n, mask=dm, mask_bit_offset=5),
                                        c.sum_dd("f64", df, n), c.sum("f64", df, n, mask=dm))
        fenced.close()
    finally:
        ctx.set_grid(0)
    assert got[(0, 0)][0] == (oracle.sum_scalar(a), n)
    assert got[(0, 0)][1] == oracle.masked_sum(a, bits, 5)
    for grid in (0, 3, 97, 2048):
        assert got[(0, grid)] == got[(256, grid)]          # same grid, same fold order: bit-identical, floats included
        assert got[(0, grid)][:2] == got[(0, 0)][:2]       # integers: any grid
    for b in (da, df, dm):
        b.free()


# ---- extended_numeric_types: i8 / u8 / i16 / u16 columns (src/enums/collections/numeric_array.rs:81-99) ---------------
NARROW = [("i8", np.int8), ("u8", np.uint8), ("i16", np.int16), ("u16", np.uint16)]


@pytest.mark.parametrize("tag,dt", NARROW)
@pytest.mark.parametrize("n", [1, 15, 16, 17, 63, 64, 65, 1000, 4095, 4096, 4097, 32768, 100_003, (1 << 22) + 37])
def test_narrow_int_sums_dense_and_masked(ctx, oracle, tag, dt, n):
    """16 / 8 rows per 16-byte load are summed inside 32-bit registers (v_sad_u8, masked half adds; signed types through a
    bias): bit-exact against the oracle's widening scalar loop — extreme values, every validity phase, windows that start
    off the 16-byte boundary, pageable and device operands, sum and mean."""
    rng = np.random.default_rng(n * 7 + len(tag))
    info = np.iinfo(dt)
    for kind in ("random", "extremes"):
        if kind == "random":
            a = rng.integers(info.min, info.max, size=n + 5, endpoint=True).astype(dt)
        else:
            a = rng.choice(np.array([info.min, info.max, 0, -1 if info.min < 0 else 1], dtype=dt), size=n + 5)
        bits = rng.integers(0, 256, size=(n + 200) // 8 + 24, dtype=np.uint8)
        d, m = ctx.to_device(a, 64), ctx.to_device(bits, 16)
        for shift in (0, 1, 3):  # element offsets: the window starts mid-vector
            win = a[shift:shift + n]
            want = oracle.masked_sum(np.ascontiguousarray(win), None, 0)
            assert ctx.sum(tag, d.offset(shift * a.itemsize), n) == want, (tag, kind, n, shift)
            for off in (0, 5, 64, 77):
                want = oracle.masked_sum(np.ascontiguousarray(win), bits, off)
                got = ctx.sum(tag, d.offset(shift * a.itemsize), n, mask=m, mask_bit_offset=off)
                assert got == want, (tag, kind, n, shift, off, got, want)
        # host (pageable) operand, and the mean
        want = oracle.masked_sum(np.ascontiguousarray(a[:n]), bits, 3)
        assert ctx.sum(tag, np.ascontiguousarray(a[:n]), n, mask=bits, mask_bit_offset=3) == want
        mean, cnt = ctx.mean(tag, d, n, mask=m, mask_bit_offset=3)
        signed = info.min < 0
        s = want[0] - (1 << 64) if (signed and want[0] >= 1 << 63) else want[0]
        assert cnt == want[1] and (np.isnan(mean) if cnt == 0 else mean == float(s) / cnt)
        d.free()
        m.free()


@pytest.mark.parametrize("tag,dt", [("i8", np.int8), ("u8", np.uint8)])
def test_gated_byte_sums_in_the_deep_shape_with_a_tight_bitmap(ctx, oracle, tag, dt):
    """Round 5's shape of the Bitmask-gated 1-byte sums: 8 loads per lane, a wave's run of 8192 rows = 128 validity words held
    two per lane, a third load for the funnel partner of the last — clamped, not skipped, at the window's last word. Whole
    numbers of 32 768-row workgroup tiles and ragged tails, every funnel phase incl. 0 (where word 128 is NOT part of the
    window) and 63, the bitmap allocated to the byte (whole words, the reference's convention, and nothing behind them)."""
    rng = np.random.default_rng(len(tag) + 50)
    info = np.iinfo(dt)
    tile = 32768
    for n in (tile, 3 * tile, 5 * tile + 4097, 16 * tile - 1):
        a = rng.integers(info.min, info.max, size=n, endpoint=True).astype(dt)
        d = ctx.to_device(a)
        for off in (0, 1, 13, 63, 64, 127):
            n_words = (off + n + 63) // 64
            bits = rng.integers(0, 256, size=n_words * 8, dtype=np.uint8)
            m = ctx.to_device(bits)  # exactly the window's words
            want = oracle.masked_sum(a, bits, off)
            assert ctx.sum(tag, d, n, mask=m, mask_bit_offset=off) == want, (tag, n, off)
            if tuning_build():  # round 4's shape (2 loads per lane: a tuning form) must agree
                ctx.set_variant(2)
                try:
                    assert ctx.sum(tag, d, n, mask=m, mask_bit_offset=off) == want, (tag, n, off, "round-4 shape")
                finally:
                    ctx.set_variant(0)
            m.free()
        d.free()


@pytest.mark.parametrize("tag,dt", NARROW)
def test_narrow_int_sum_of_a_large_column(ctx, tag, dt):
    """2^31 + 77 rows (more than 32 bits of index for the 1-byte types' byte offsets would need; the per-load 32-bit partial
    sums must not leak between loads): closed form of a repeating pattern, dense and with every third row null."""
    n = (1 << 31) + 77
    item = np.dtype(dt).itemsize
    buf = ctx.alloc(n * item + 64)
    pattern = np.arange(-3, 13, dtype=np.int64).astype(dt)  # 16 values, wraps for unsigned
    # fill by doubling copies on the device: pattern -> 2^k repeats
    ctx.lib.ma_dev_upload(ctx.handle, buf.ptr, pattern.ctypes.data, pattern.nbytes)
    filled = pattern.size
    while filled < n:
        k = min(filled, n - filled)
        ctx.dev_copy(buf.ptr + filled * item, buf.ptr, k * item)
        filled += k
    reps, rem = divmod(n, pattern.size)
    wide = pattern.astype(np.int64 if np.dtype(dt).kind == "i" else np.uint64)
    want = (int(wide.sum()) * reps + int(wide[:rem].sum())) & ((1 << 64) - 1)
    got, cnt = ctx.sum(tag, buf, n)
    assert (got & ((1 << 64) - 1), cnt) == (want, n)
    # every third row null (bit pattern 0b110110..., period 3 bits = 24 bits = 3 bytes)
    mbytes = n // 8 + 64
    mask = ctx.alloc(mbytes)
    period = np.array([0b10110110, 0b01101101, 0b11011011], dtype=np.uint8)  # bit i set iff i % 3 != 0
    ctx.lib.ma_dev_upload(ctx.handle, mask.ptr, period.ctypes.data, 3)
    filled = 3
    while filled < mbytes:
        k = min(filled, mbytes - filled)
        ctx.dev_copy(mask.ptr + filled, mask.ptr, k)
        filled += k
    idx = np.arange(48, dtype=np.int64)  # lcm(16, 3) rows: one period of (value, validity)
    vals = wide[idx % 16]
    valid = (idx % 3) != 0
    reps, rem = divmod(n, 48)
    want = (int(vals[valid].sum()) * reps + int(vals[:rem][valid[:rem]].sum())) & ((1 << 64) - 1)
    want_cnt = int(valid.sum()) * reps + int(valid[:rem].sum())
    got, cnt = ctx.sum(tag, buf, n, mask=mask)
    assert (got & ((1 << 64) - 1), cnt) == (want, want_cnt)
    buf.free()
    mask.free()
