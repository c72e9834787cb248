"""GPU parity tests for the elementwise arithmetic / FMA kernels, called through the C ABI
(ma_apply_int_*, ma_apply_float_*, ma_apply_fma_* and the fused scalar-broadcast forms).

The first block replays the reference's own unit tests (src/kernels/arithmetic/mod.rs:117-537, vectors in
tests/golden/arithmetic_kat.json, routing vectors in tests/golden/routing_kat.json). The second block compares
against the CPU oracle (oracle/minarrow_oracle.c) on seeded random inputs.

Bar: integers and validity bitmaps bit-exact; float Add/Sub/Mul/Div/Rem/FloorDiv/FMA bit-exact (NaN == NaN);
float Power = exp(b * ln(a)) within (4 + 2|b ln a|) ULP of the oracle (two libm calls on each side).
"""
import json
import math
import zlib
from pathlib import Path

import numpy as np
import pytest

from minarrow_amd import ffi

pytestmark = pytest.mark.gpu

GOLD = Path(__file__).resolve().parent / "golden"
ARITH = json.loads((GOLD / "arithmetic_kat.json").read_text())
ROUTE = json.loads((GOLD / "routing_kat.json").read_text())
NP = {"i8": np.int8, "u8": np.uint8, "i16": np.int16, "u16": np.uint16, "i32": np.int32, "u32": np.uint32,
      "i64": np.int64, "u64": np.uint64, "f32": np.float32, "f64": np.float64}
OPS = {"add": 0, "subtract": 1, "multiply": 2, "divide": 3, "remainder": 4, "power": 5, "floordiv": 6}
INT_TAGS = ("i8", "u8", "i16", "u16", "i32", "u32", "i64", "u64")
FLOAT_TAGS = ("f32", "f64")


def seed_of(*parts):
    return zlib.crc32(repr(parts).encode())


def mask_bytes(n):
    return ((n + 63) // 64) * 8


class Gpu:
    """Thin test helper: numpy in, numpy out, everything through the C ABI on device-resident buffers."""

    def __init__(self, ctx):
        self.ctx = ctx

    def apply(self, tag, lhs, rhs, op, mask=None, mask_off=0, device=True, shift=0):
        dt = NP[tag]
        lhs, rhs = np.asarray(lhs, dtype=dt), np.asarray(rhs, dtype=dt)
        n = lhs.size
        op = OPS[op] if isinstance(op, str) else op
        if device:
            sb = shift * np.dtype(dt).itemsize
            dl = self.ctx.to_device(np.concatenate([np.zeros(shift, dt), lhs]), 64)
            dr = self.ctx.to_device(np.concatenate([np.zeros(shift, dt), rhs]), 64)
            do = self.ctx.alloc(max(n, 1) * lhs.itemsize + 64 + sb)
            dm = self.ctx.to_device(mask, 16) if mask is not None else None
            dom = self.ctx.alloc(mask_bytes(n) + 8) if mask is not None else None
            self.ctx.apply(tag, dl.offset(sb), dr.offset(sb), op, do.offset(sb), n, rhs.size, mask=dm,
                           mask_bit_offset=mask_off, out_mask=dom)
            out = do.download(dt, n, sb)
            om = dom.download(np.uint8, mask_bytes(n)) if mask is not None else None
        else:
            out = np.zeros(n, dtype=dt)
            om = np.zeros(mask_bytes(n) + 8, dtype=np.uint8) if mask is not None else None
            self.ctx.apply(tag, lhs, rhs, op, out, n, rhs.size, mask=mask, mask_bit_offset=mask_off, out_mask=om)
        return out, om

    def apply_scalar(self, tag, side, arr, scalar, op, mask=None, mask_off=0):
        dt = NP[tag]
        arr = np.asarray(arr, dtype=dt)
        n = arr.size
        da = self.ctx.to_device(arr, 64)
        do = self.ctx.alloc(max(n, 1) * arr.itemsize + 64)
        dm = self.ctx.to_device(mask, 16) if mask is not None else None
        dom = self.ctx.alloc(mask_bytes(n) + 8) if mask is not None else None
        self.ctx.apply_scalar(tag, side, da, n, scalar, OPS[op] if isinstance(op, str) else op, do, mask=dm,
                              mask_bit_offset=mask_off, out_mask=dom)
        return do.download(dt, n), (dom.download(np.uint8, mask_bytes(n)) if mask is not None else None)

    def fma(self, tag, a, b, c, mask=None, mask_off=0):
        dt = NP[tag]
        a, b, c = (np.asarray(x, dtype=dt) for x in (a, b, c))
        n = a.size
        da, db, dc = (self.ctx.to_device(x, 64) for x in (a, b, c))
        do = self.ctx.alloc(max(n, 1) * a.itemsize + 64)
        dm = self.ctx.to_device(mask, 16) if mask is not None else None
        dom = self.ctx.alloc(mask_bytes(n) + 8) if mask is not None else None
        self.ctx.apply_fma(tag, da, db, dc, do, n, b.size, c.size, mask=dm, mask_bit_offset=mask_off, out_mask=dom)
        return do.download(dt, n), (dom.download(np.uint8, mask_bytes(n)) if mask is not None else None)


@pytest.fixture(scope="module")
def gpu(ctx):
    return Gpu(ctx)


def bits_of(valid):
    return np.packbits(np.asarray(valid, dtype=bool), bitorder="little")


def unpack(bits, n, off=0):
    return np.unpackbits(bits, bitorder="little")[off:off + n].astype(bool)


def assert_float_bits_equal(got, want):
    """Bit-exact including the sign of zero; any NaN equals any NaN (payload/sign of a generated NaN is
    platform-defined: x86 produces the negative default NaN, gfx950 the positive one)."""
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape
    nan_g, nan_w = np.isnan(got), np.isnan(want)
    np.testing.assert_array_equal(nan_g, nan_w)
    ui = np.uint32 if got.dtype == np.float32 else np.uint64
    np.testing.assert_array_equal(got.view(ui)[~nan_g], want.view(ui)[~nan_w])


# ======================================================================================================
# 1. The reference's own tests, replayed through the C ABI
# ======================================================================================================

@pytest.mark.parametrize("tag", INT_TAGS)
def test_ref_int_dense(gpu, tag):
    """int_kernel_suite! $fn_dense — src/kernels/arithmetic/mod.rs:117-178"""
    g = ARITH["int_dense"]
    for op, expect in g["expect"].items():
        out, om = gpu.apply(tag, g["lhs"], g["rhs"], op)
        assert om is None
        np.testing.assert_array_equal(out, np.array(expect, dtype=NP[tag]))
    expect = []
    for a, b in zip(g["lhs"], g["rhs"]):
        acc = 1
        for _ in range(b):
            acc = (acc * a) % (1 << (8 * np.dtype(NP[tag]).itemsize))
        expect.append(acc)
    out, _ = gpu.apply(tag, g["lhs"], g["rhs"], "power")
    np.testing.assert_array_equal(out.astype(object) % (1 << (8 * out.itemsize)), expect)  # same bits, any signedness
    # "Dense integer kernel division by zero must panic" -> MA_ERR_DIVIDE_BY_ZERO
    for op in g["divide_by_zero_ops_must_panic"]:
        with pytest.raises(ffi.MinarrowHipError) as e:
            gpu.apply(tag, g["lhs"], g["divide_by_zero_rhs"], op)
        assert e.value.status == ffi.MA_ERR_DIVIDE_BY_ZERO
        with pytest.raises(ffi.MinarrowHipError) as e:
            gpu.apply(tag, g["lhs"], g["divide_by_zero_rhs"], op, device=False)
        assert e.value.status == ffi.MA_ERR_DIVIDE_BY_ZERO
    # the latch is cleared: the next call succeeds
    out, _ = gpu.apply(tag, g["lhs"], g["rhs"], "divide")
    np.testing.assert_array_equal(out, np.array(g["expect"]["divide"], dtype=NP[tag]))


@pytest.mark.parametrize("tag", INT_TAGS)
def test_ref_int_masked(gpu, tag):
    """int_kernel_suite! $fn_masked — src/kernels/arithmetic/mod.rs:180-220"""
    for case in ARITH["int_masked"]["cases"]:
        for device in (True, False):
            out, om = gpu.apply(tag, case["lhs"], case["rhs"], case["op"], mask=bits_of(case["mask"]), device=device)
            np.testing.assert_array_equal(out, np.array(case["expect"], dtype=NP[tag]))
            np.testing.assert_array_equal(unpack(om, 4), case["expect_mask"])


@pytest.mark.parametrize("tag", INT_TAGS + FLOAT_TAGS)
def test_ref_empty_and_length_mismatch(gpu, ctx, tag):
    """$fn_empty — mod.rs:222-228; confirm_equal_len — src/utils.rs:163-171"""
    out, _ = gpu.apply(tag, [], [], "add")
    assert out.size == 0
    a, b = np.zeros(3, NP[tag]), np.zeros(2, NP[tag])
    with pytest.raises(ffi.MinarrowHipError) as e:
        ctx.apply(tag, a, b, 0, np.zeros(3, NP[tag]), 3, 2)
    assert e.value.status == ffi.MA_ERR_LENGTH_MISMATCH and "length mismatch" in e.value.message


@pytest.mark.parametrize("tag", FLOAT_TAGS)
def test_ref_float_suite(gpu, tag):
    """float_kernel_suite! — src/kernels/arithmetic/mod.rs:293-370"""
    g = ARITH["float"]
    eps = g["eps"][tag]
    for op, expect in g["expect_exact"].items():
        out, _ = gpu.apply(tag, g["lhs"], g["rhs"], op)
        np.testing.assert_array_equal(out, np.array(expect, dtype=NP[tag]))
    out, _ = gpu.apply(tag, g["lhs"], g["rhs"], "remainder")
    assert np.all(np.abs(out) < eps)
    out, _ = gpu.apply(tag, g["lhs"], g["rhs"], "power")
    expect = np.exp(np.array(g["rhs"]) * np.log(np.array(g["lhs"])))
    assert np.all(np.abs(out.astype(np.float64) - expect) < max(eps, 2e-5 if tag == "f32" else eps) * np.maximum(1, expect))
    assert np.all(np.isinf(gpu.apply(tag, g["lhs"], g["divide_by_zero_rhs"], "divide")[0]))
    assert np.all(np.isnan(gpu.apply(tag, g["lhs"], g["divide_by_zero_rhs"], "remainder")[0]))
    m = g["masked"]
    out, om = gpu.apply(tag, g["lhs"], g["rhs"], m["op"], mask=bits_of(m["mask"]))
    np.testing.assert_array_equal(out, np.array(m["expect"], dtype=NP[tag]))
    np.testing.assert_array_equal(unpack(om, 4), m["mask"])


@pytest.mark.parametrize("tag", FLOAT_TAGS)
def test_ref_fma(gpu, ctx, tag):
    """fma_f32 / fma_f64 — src/kernels/arithmetic/mod.rs:372-399"""
    g = ARITH["fma"]
    out, om = gpu.fma(tag, g["lhs"], g["rhs"], g["acc"])
    np.testing.assert_array_equal(out, np.array(g["expect"], dtype=NP[tag]))
    assert om is None
    out, om = gpu.fma(tag, g["lhs"], g["rhs"], g["acc"], mask=bits_of(g["masked"]["mask"]))
    np.testing.assert_array_equal(out, np.array(g["masked"]["expect"], dtype=NP[tag]))
    np.testing.assert_array_equal(unpack(om, 3), g["masked"]["expect_mask"])
    assert gpu.fma(tag, [], [], [])[0].size == 0
    a = np.zeros(3, NP[tag])
    with pytest.raises(ffi.MinarrowHipError) as e:
        ctx.apply_fma(tag, a, a, a[:2].copy(), np.zeros(3, NP[tag]), 3, 3, 2)
    assert e.value.status == ffi.MA_ERR_LENGTH_MISMATCH


def test_ref_datetime(ctx, gpu, oracle):
    """datetime_add / datetime_all_ops / datetime_masked_and_empty / datetime_len_mismatch_panics —
    src/kernels/arithmetic/mod.rs:418-505, through ma_apply_datetime_i64 (dispatch.rs:309-372)."""
    g = ARITH["datetime_i64"]

    def run(lhs, rhs, op, lmask=None, rmask=None, loff=0, roff=0, n=None):
        lhs, rhs = np.asarray(lhs, dtype=np.int64), np.asarray(rhs, dtype=np.int64)
        n = lhs.size - loff if n is None else n
        out = np.zeros(max(n, 1), dtype=np.int64)
        om = np.zeros(mask_bytes(n) + 8, dtype=np.uint8)
        has = ctx.apply_datetime("i64", lhs, loff, n, lmask, rhs, roff, rhs.size - roff if n == lhs.size - loff else n,
                                 rmask, OPS[op], out, om)
        return out[:n], (unpack(om, n) if has else None)

    for case in g["cases"]:
        out, valid = run(case["lhs"], case["rhs"], case["op"])
        np.testing.assert_array_equal(out, case["expect"])
        assert valid is None  # "assert!(out.null_mask.is_none())"
    m = g["masked"]
    out, valid = run(m["lhs"], m["rhs"], m["op"], lmask=np.concatenate([bits_of(m["lhs_mask"]), np.zeros(15, np.uint8)]))
    np.testing.assert_array_equal(out, m["expect"])
    np.testing.assert_array_equal(valid, m["expect_mask"])
    out, valid = run([], [], "add")
    assert out.size == 0
    with pytest.raises(ffi.MinarrowHipError) as e:  # #[should_panic(expected = "apply_datetime: length mismatch")]
        ctx.apply_datetime("i64", np.array([1000, 2000]), 0, 2, None, np.array([10]), 0, 1, None, 0, np.zeros(2, np.int64), None)
    assert e.value.status == ffi.MA_ERR_LENGTH_MISMATCH and "apply_datetime: length mismatch" in e.value.message
    # both sides masked + view offsets: data is windowed, masks are AND-ed from bit 0 (dispatch.rs:321-324)
    rng = np.random.default_rng(8)
    n, loff, roff = 5000, 3, 10
    a = rng.integers(-1000, 1000, size=n + loff).astype(np.int64)
    b = rng.integers(1, 1000, size=n + roff).astype(np.int64)
    ma_, mb_ = rng.integers(0, 256, size=n // 8 + 16, dtype=np.uint8), rng.integers(0, 256, size=n // 8 + 16, dtype=np.uint8)
    out, valid = run(a, b, "multiply", lmask=ma_, rmask=mb_, loff=loff, roff=roff, n=n)
    want_valid = unpack(ma_, n) & unpack(mb_, n)
    np.testing.assert_array_equal(valid, want_valid)
    np.testing.assert_array_equal(out, np.where(want_valid, a[loff:loff + n] * b[roff:roff + n], 0))


@pytest.mark.parametrize("tag", ["i32", "u32", "i64", "u64"])
@pytest.mark.parametrize("n", [1, 63, 64, 65, 4097, 32768 + 5, (1 << 18) + 77])
def test_datetime_two_masks_fused_matches_merge_then_apply(ctx, oracle, tag, n):
    """apply_datetime with BOTH operands masked: the kernel ANDs the two validities in registers (no merged
    temporary). Must equal the reference's two-step form — merge_bitmasks_to_new, then apply_int with the merged
    mask (dispatch.rs:336-365) — for every op, incl. the data-dependent validity of Div / Rem / FloorDiv, on
    device-resident and on host operands."""
    rng = np.random.default_rng(n * 7 + len(tag))
    dt = NP[tag]
    info = np.iinfo(dt)
    a = rng.integers(max(info.min, -(1 << 40)), min(info.max, 1 << 40), size=n).astype(dt)
    b = rng.integers(0, 9, size=n).astype(dt)  # zeros included: they null rows of the division ops
    m1 = rng.integers(0, 256, size=n // 8 + 16, dtype=np.uint8)
    m2 = rng.integers(0, 256, size=n // 8 + 16, dtype=np.uint8)
    merged = oracle.merge_bitmasks(m1, m2, n)
    da, db, dm1, dm2 = ctx.to_device(a, 64), ctx.to_device(b, 64), ctx.to_device(m1, 16), ctx.to_device(m2, 16)
    dout, dom = ctx.alloc(n * a.itemsize + 64), ctx.alloc(mask_bytes(n) + 16)
    for op in ("add", "subtract", "multiply", "divide", "remainder", "floordiv"):
        st, want, want_mask, _ = oracle.apply_int(a, b, op, merged, n)
        assert st == 0
        want_valid = unpack(want_mask, n)
        # device-resident
        assert ctx.apply_datetime(tag, da, 0, n, dm1, db, 0, n, dm2, OPS[op], dout, dom)
        np.testing.assert_array_equal(dout.download(dt, n), want[:n])
        np.testing.assert_array_equal(unpack(dom.download(np.uint8, mask_bytes(n)), n), want_valid)
        # host operands (staged)
        out = np.zeros(n, dtype=dt)
        om = np.zeros(mask_bytes(n) + 8, dtype=np.uint8)
        assert ctx.apply_datetime(tag, a, 0, n, m1, b, 0, n, m2, OPS[op], out, om)
        np.testing.assert_array_equal(out, want[:n])
        np.testing.assert_array_equal(unpack(om, n), want_valid)
    for buf in (da, db, dm1, dm2, dout, dom):
        buf.free()


@pytest.mark.parametrize("tag", ["i8", "u8"])
def test_8bit_division_exhaustive(ctx, oracle, tag):
    """Every operand pair of the 1-byte types (65 536 per type) through Divide / Remainder / FloorDiv, dense (zero
    divisors excluded: they are the reference's panic) and masked (zero divisors null the row): the kernels compute
    8-bit quotients through a float reciprocal, which must be exact everywhere, MIN / -1 included."""
    dt = NP[tag]
    info = np.iinfo(dt)
    vals = np.arange(info.min, info.max + 1, dtype=np.int64)
    a = np.repeat(vals, vals.size).astype(dt)
    b = np.tile(vals, vals.size).astype(dt)
    n = a.size
    nz = b != 0
    an, bn = np.ascontiguousarray(a[nz]), np.ascontiguousarray(b[nz])
    da, db = ctx.to_device(an, 64), ctx.to_device(bn, 64)
    dout = ctx.alloc(an.size + 64)
    for op in ("divide", "remainder", "floordiv"):
        st, want, _, _ = oracle.apply_int(an, bn, op)
        assert st & ~4 == 0  # bit 2: the scalar body's MIN / -1 overflow panic; the value written is the wrapping one
        ctx.apply(tag, da, db, OPS[op], dout, an.size, an.size)
        np.testing.assert_array_equal(dout.download(dt, an.size), want[:an.size], err_msg=f"{tag} {op} dense")
    mask = np.full(mask_bytes(n) + 16, 0xFF, dtype=np.uint8)
    dA, dB, dm = ctx.to_device(a, 64), ctx.to_device(b, 64), ctx.to_device(mask, 16)
    dO, dom = ctx.alloc(n + 64), ctx.alloc(mask_bytes(n) + 16)
    for op in ("divide", "remainder", "floordiv"):
        st, want, want_mask, _ = oracle.apply_int(a, b, op, mask, n)
        assert st & ~4 == 0  # bit 2: the scalar body's MIN / -1 overflow panic; the value written is the wrapping one
        ctx.apply(tag, dA, dB, OPS[op], dO, n, n, mask=dm, out_mask=dom)
        np.testing.assert_array_equal(dO.download(dt, n), want[:n], err_msg=f"{tag} {op} masked")
        np.testing.assert_array_equal(unpack(dom.download(np.uint8, mask_bytes(n)), n), unpack(want_mask, n))
    for buf in (da, db, dout, dA, dB, dm, dO, dom):
        buf.free()


def test_ref_int_power_short_vs_long(gpu):
    """test_int_dense_power_short_vs_long_input_simd — mod.rs:507-537"""
    g = ARITH["int_power_short_vs_long"]
    for n in g["lengths"]:
        out, _ = gpu.apply("u32", np.full(n, g["base"]), np.full(n, g["exp"]), "power")
        assert np.all(out == g["expect_each"])


def test_ref_routing_vectors(gpu):
    """Array(+)Array, Array(+)len-1 Array, Array(+)Scalar — src/kernels/broadcast/array.rs:485-556,685-700,
    src/kernels/broadcast/super_array.rs:480-560, src/kernels/routing/binary_map.rs:76-152. A length-1 side is
    handed to the fused scalar kernel instead of being materialised (routing/broadcast.rs:25-112)."""
    def run(tag, op, lhs, rhs):
        if len(lhs) == 1 and len(rhs) != 1:
            return gpu.apply_scalar(tag, "lhs", rhs, lhs[0], op)[0]
        if len(rhs) == 1 and len(lhs) != 1:
            return gpu.apply_scalar(tag, "rhs", lhs, rhs[0], op)[0]
        return gpu.apply(tag, lhs, rhs, op)[0]

    for c in ROUTE["array_array"]["cases"] + ROUTE["super_array"]["cases"]:
        np.testing.assert_array_equal(run("i32", c["op"], c["lhs"], c["rhs"]), c["expect"])
    c = ROUTE["array_scalar"]
    np.testing.assert_array_equal(run("i32", c["op"], c["lhs"], [c["scalar"]]), c["expect"])
    c = ROUTE["super_array"]["chunked_add"]
    for l, r, e in zip(c["lhs_chunks"], c["rhs_chunks"], c["expect_chunks"]):
        np.testing.assert_array_equal(run("i32", "add", l, r), e)
    for c in ROUTE["binary_map_f64"]["cases"]:
        lhs = [float(x) for x in c.get("lhs", c.get("lhs_i32"))]
        rhs = c.get("rhs", [c.get("scalar")])
        np.testing.assert_array_equal(run("f64", c["op"], lhs, rhs), c["expect"])


# ======================================================================================================
# 2. Seeded random parity against the oracle
# ======================================================================================================

SIZES = [1, 5, 63, 64, 65, 1000, 4095, 4096, 4097, 8193, 100_003]


def rand_ints(rng, tag, n, small=False):
    info = np.iinfo(NP[tag])
    if small:
        lo, hi = (0, 40) if info.min == 0 else (-20, 40)
        return rng.integers(lo, hi, size=n).astype(NP[tag])
    a = rng.integers(info.min, info.max, size=n, dtype=NP[tag], endpoint=True)
    # sprinkle the edge values
    edges = np.array([info.min, info.max, 0, 1, info.max - 1] + ([-1] if info.min < 0 else [2]), dtype=NP[tag])
    idx = rng.integers(0, n, size=max(1, n // 16))
    a[idx] = edges[rng.integers(0, edges.size, size=idx.size)]
    return a


@pytest.mark.parametrize("tag", INT_TAGS)
@pytest.mark.parametrize("op", list(OPS))
def test_int_dense_random(gpu, oracle, tag, op):
    rng = np.random.default_rng(seed_of(tag, op))
    for n in SIZES:
        lhs = rand_ints(rng, tag, n)
        rhs = rand_ints(rng, tag, n, small=(op == "power"))
        if op in ("divide", "remainder", "floordiv"):
            rhs[rhs == 0] = 3  # the dense kernels panic on a zero divisor: covered by test_ref_int_dense
        st, want, _, _ = oracle.apply_int(oracle.aligned_copy(lhs), oracle.aligned_copy(rhs), op)
        assert st & ~oracle.PANIC_OVERFLOW == 0  # MIN / -1: wrapping value, as the reference's SIMD lanes give
        for shift in (0, 1):
            got, _ = gpu.apply(tag, lhs, rhs, op, shift=shift)
            np.testing.assert_array_equal(got, want, err_msg=f"{tag} {op} n={n} shift={shift}")


@pytest.mark.parametrize("tag", INT_TAGS)
@pytest.mark.parametrize("op", list(OPS))
def test_int_masked_random(gpu, oracle, tag, op):
    rng = np.random.default_rng(seed_of(tag, op, "m"))
    for n in SIZES:
        lhs = rand_ints(rng, tag, n)
        rhs = rand_ints(rng, tag, n, small=(op == "power"))
        rhs[rng.integers(0, n, size=max(1, n // 7))] = 0  # zero divisors become nulls
        for mask_off in (0, 3, 64, 77):
            bits = rng.integers(0, 256, size=(mask_off + n + 7) // 8 + 8, dtype=np.uint8)
            window = oracle.pad_bits(np.packbits(unpack(bits, n, mask_off), bitorder="little"), n)
            st, want, want_mask = oracle.int_body("masked_std", lhs, rhs, op, mask=window)
            assert st & ~oracle.PANIC_OVERFLOW == 0
            st2, want2, want_mask2 = oracle.int_body("masked_simd", oracle.aligned_copy(lhs), oracle.aligned_copy(rhs), op,
                                                     mask=window, lanes=8)
            np.testing.assert_array_equal(want, want2)
            got, got_mask = gpu.apply(tag, lhs, rhs, op, mask=bits, mask_off=mask_off)
            np.testing.assert_array_equal(got, want, err_msg=f"{tag} {op} n={n} off={mask_off}")
            np.testing.assert_array_equal(got_mask, want_mask[:mask_bytes(n)], err_msg=f"{tag} {op} n={n} off={mask_off} (validity)")
            np.testing.assert_array_equal(want_mask[:mask_bytes(n)], want_mask2[:mask_bytes(n)])


def rand_floats(rng, tag, n, positive=False):
    dt = NP[tag]
    a = (rng.standard_normal(n) * 10.0 ** rng.integers(-6, 7, size=n)).astype(dt)
    if positive:
        a = np.abs(a) + dt(1e-3)
    else:
        specials = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, np.finfo(dt).tiny / 4, -np.finfo(dt).tiny / 8,
                             np.finfo(dt).max, 1.0, -1.0], dtype=dt)
        idx = rng.integers(0, n, size=max(1, n // 10))
        a[idx] = specials[rng.integers(0, specials.size, size=idx.size)]
    return a


@pytest.mark.parametrize("tag", FLOAT_TAGS)
@pytest.mark.parametrize("op", ["add", "subtract", "multiply", "divide", "remainder", "floordiv"])
def test_float_exact_ops_random(gpu, oracle, tag, op):
    rng = np.random.default_rng(seed_of(tag, op))
    for n in SIZES:
        lhs, rhs = rand_floats(rng, tag, n), rand_floats(rng, tag, n)
        st, want, _, _ = oracle.apply_float(oracle.aligned_copy(lhs), oracle.aligned_copy(rhs), op)
        assert st == 0
        for shift in (0, 1):
            got, _ = gpu.apply(tag, lhs, rhs, op, shift=shift)
            assert_float_bits_equal(got, want)
        bits = rng.integers(0, 256, size=(5 + n + 7) // 8 + 8, dtype=np.uint8)
        window = oracle.pad_bits(np.packbits(unpack(bits, n, 5), bitorder="little"), n)
        st, want, want_mask = oracle.float_body("masked_std", lhs, rhs, op, mask=window)
        got, got_mask = gpu.apply(tag, lhs, rhs, op, mask=bits, mask_off=5)
        assert_float_bits_equal(got, want)
        np.testing.assert_array_equal(got_mask, want_mask[:mask_bytes(n)])


@pytest.mark.parametrize("tag", FLOAT_TAGS)
def test_float_power_random(gpu, oracle, tag):
    rng = np.random.default_rng(17)
    dt = NP[tag]
    for n in (1000, 8193):
        lhs = rand_floats(rng, tag, n, positive=True)
        rhs = (rng.standard_normal(n) * 3).astype(dt)
        st, want, _, _ = oracle.apply_float(oracle.aligned_copy(lhs), oracle.aligned_copy(rhs), "power")
        got, _ = gpu.apply(tag, lhs, rhs, "power")
        x = np.abs(rhs.astype(np.float64) * np.log(lhs.astype(np.float64)))
        finite = np.isfinite(want)
        ulp = np.spacing(np.abs(want).astype(dt)).astype(np.float64)
        with np.errstate(invalid="ignore"):  # inf - inf on the non-finite entries, which are compared separately
            diff = np.abs(got.astype(np.float64) - want.astype(np.float64))
        if tag == "f32":
            # ln, product and exp are rounded to f32 at the same three points as the reference, and the device's two
            # transcendental steps ARE the correctly rounded ones (tests/test_gpu_pow_series.py: 100 % of 60 000 + 67 500
            # inputs); the host libm's logf / expf are too in all but rare cases. Where one of them is not, the 1-ULP ln
            # difference is amplified by |b ln a|.
            assert np.mean(diff[finite] == 0) > 0.99
            assert np.all(diff[finite] <= (1 + 2 * x[finite]) * ulp[finite])
        else:
            # f64: the device's ln is within 0.52 ULP of exact, glibc's within 0.61 (same test file): they disagree on <1 % of
            # inputs, by 1 ULP, which exp amplifies by |b ln a|; the two exp implementations are < 1 ULP each.
            # (a 1-ULP ln difference moves the rounded product by up to 2 of ITS ulps, each worth up to |b ln a| result ulps)
            assert np.all(diff[finite] <= (2 + 2 * x[finite]) * ulp[finite])
            assert np.mean(diff[finite] <= ulp[finite]) > 0.98
        np.testing.assert_array_equal(got[~finite], want[~finite])
    # ln of a negative base is NaN, of zero -inf: same special-case structure as the reference
    got, _ = gpu.apply(tag, [-2.0, 0.0, 0.0, 1.0], [2.0, 2.0, -1.0, 1e30], "power")
    assert np.isnan(got[0]) and got[1] == 0.0 and got[2] == np.inf and got[3] == 1.0


@pytest.mark.parametrize("tag", INT_TAGS + FLOAT_TAGS)
@pytest.mark.parametrize("op", ["add", "subtract", "multiply", "divide", "floordiv"])
def test_scalar_broadcast_equals_materialised(gpu, oracle, tag, op):
    """Fused scalar kernels == the reference's vec64![x; n] + two-array kernel (routing/broadcast.rs:25-112)."""
    rng = np.random.default_rng(seed_of(tag, op, "s"))
    is_f = tag in FLOAT_TAGS
    for n in (7, 4097, 50_001):
        arr = rand_floats(rng, tag, n) if is_f else rand_ints(rng, tag, n)
        scalar = NP[tag](2.5) if is_f else NP[tag](7)
        full = np.full(n, scalar, dtype=NP[tag])
        fn = oracle.apply_float if is_f else oracle.apply_int
        if not is_f and op in ("divide", "floordiv"):
            arr[arr == 0] = 5
        for side in ("rhs", "lhs"):
            l, r = (arr, full) if side == "rhs" else (full, arr)
            st, want, _, _ = fn(oracle.aligned_copy(l), oracle.aligned_copy(r), op)
            assert st & ~oracle.PANIC_OVERFLOW == 0
            got, _ = gpu.apply_scalar(tag, side, arr, scalar, op)
            (assert_float_bits_equal if is_f else np.testing.assert_array_equal)(got, want)
        # masked
        bits = rng.integers(0, 256, size=n // 8 + 16, dtype=np.uint8)
        window = oracle.pad_bits(np.packbits(unpack(bits, n, 9), bitorder="little"), n)
        body = oracle.float_body if is_f else oracle.int_body
        st, want, want_mask = body("masked_std", arr, full, op, mask=window)
        got, got_mask = gpu.apply_scalar(tag, "rhs", arr, scalar, op, mask=bits, mask_off=9)
        (assert_float_bits_equal if is_f else np.testing.assert_array_equal)(got, want)
        np.testing.assert_array_equal(got_mask, want_mask[:mask_bytes(n)])


@pytest.mark.parametrize("tag", FLOAT_TAGS)
def test_fma_random(gpu, oracle, tag):
    rng = np.random.default_rng(23)
    for n in SIZES:
        a, b, c = (rand_floats(rng, tag, n) for _ in range(3))
        st, want, _ = oracle.apply_fma(oracle.aligned_copy(a), oracle.aligned_copy(b), oracle.aligned_copy(c))
        got, _ = gpu.fma(tag, a, b, c)
        assert_float_bits_equal(got, want)  # fused on both sides (simd.rs:620, std.rs:210)
        bits = rng.integers(0, 256, size=n // 8 + 16, dtype=np.uint8)
        window = oracle.pad_bits(np.packbits(unpack(bits, n, 2), bitorder="little"), n)
        st, want, want_mask = oracle.apply_fma(oracle.aligned_copy(a), oracle.aligned_copy(b), oracle.aligned_copy(c), mask=window)
        got, got_mask = gpu.fma(tag, a, b, c, mask=bits, mask_off=2)
        assert_float_bits_equal(got, want)
        np.testing.assert_array_equal(got_mask, want_mask[:mask_bytes(n)])


@pytest.mark.parametrize("tag", ["i8", "u16", "i32", "i64", "f32", "f64"])
def test_mixed_phase_operands(ctx, oracle, tag):
    """lhs, rhs and out on different 16-byte phases (views sliced at different offsets, routing/arithmetic.rs:273-285):
    the stores stay 16-byte aligned after the head rows, the inputs are read with element-aligned vector loads.
    Results must equal the oracle's, dense and masked, incl. data-dependent validity for integer division."""
    dt = np.dtype({"i8": np.int8, "u16": np.uint16, "i32": np.int32, "i64": np.int64, "f32": np.float32, "f64": np.float64}[tag])
    n = 150_001
    rng = np.random.default_rng(dt.itemsize)
    pad = 40
    a = rng.integers(1, 100, size=n + pad).astype(dt)
    b = rng.integers(0, 7, size=n + pad).astype(dt)  # zeros: masked integer division nulls those rows
    da, db = ctx.to_device(a, 64), ctx.to_device(b, 64)
    do = ctx.alloc((n + pad) * dt.itemsize + 64)
    bits = rng.integers(0, 256, size=n // 8 + 16, dtype=np.uint8)
    dm, dom = ctx.to_device(bits, 16), ctx.alloc(n // 8 + 64)
    is_float = tag in ("f32", "f64")
    ref = oracle.apply_float if is_float else oracle.apply_int
    for la, lb, lo in ((1, 0, 0), (0, 3, 1), (5, 2, 7), (1, 1, 2)):
        x, y = np.ascontiguousarray(a[la:la + n]), np.ascontiguousarray(b[lb:lb + n])
        pa, pb, po = da.ptr + la * dt.itemsize, db.ptr + lb * dt.itemsize, do.ptr + lo * dt.itemsize
        for op in (0, 2) + ((3,) if is_float else ()):
            ctx.apply(tag, pa, pb, op, po, n, n)
            st, want, _, _ = ref(x, y, op)
            assert st == 0
            got = do.download(dt, n, lo * dt.itemsize)
            np.testing.assert_array_equal(got.view(np.uint8), want.view(np.uint8))
        # masked, with validity decided by the data for integer division
        ctx.apply(tag, pa, pb, 3, po, n, n, mask=dm, out_mask=dom)
        st, want, want_mask, _ = ref(x, y, 3, mask=bits)
        assert st == 0
        got = do.download(dt, n, lo * dt.itemsize)
        np.testing.assert_array_equal(got.view(np.uint8), want.view(np.uint8))
        nb = ((n + 63) // 64) * 8
        np.testing.assert_array_equal(dom.download(np.uint8, nb), want_mask[:nb])
        # scalar side
        ctx.apply_scalar(tag, "rhs", pa, n, 3, 2, po)
        np.testing.assert_array_equal(do.download(dt, n, lo * dt.itemsize), (x * dt.type(3)).astype(dt))
    if tag in ("f32", "f64"):
        c = rng.integers(1, 9, size=n + pad).astype(dt)
        dc = ctx.to_device(c, 64)
        ctx.apply_fma(tag, da.ptr + dt.itemsize, db.ptr + 2 * dt.itemsize, dc.ptr, do.ptr + 3 * dt.itemsize, n, n, n)
        want = (a[1:1 + n].astype(np.float64) * b[2:2 + n] + c[:n]).astype(dt)  # exact in these ranges
        np.testing.assert_array_equal(do.download(dt, n, 3 * dt.itemsize), want)


def test_async_divide_by_zero_is_reported_at_synchronize(ctx):
    n = 5000
    a = ctx.to_device(np.arange(1, n + 1, dtype=np.int64))
    z = ctx.to_device(np.zeros(n, dtype=np.int64))
    out = ctx.alloc(n * 8)
    ctx.set_async(True)
    try:
        ctx.apply("i64", a, z, OPS["divide"], out, n, n)  # enqueues; the latch is read later
        with pytest.raises(ffi.MinarrowHipError) as e:
            ctx.synchronize()
        assert e.value.status == ffi.MA_ERR_DIVIDE_BY_ZERO
        ctx.synchronize()  # latch cleared
    finally:
        ctx.set_async(False)


# ======================================================================================================
# 3. BASELINE size (config 3): 1 B-row f64 add / mul, array (+) array and array (+) scalar
# ======================================================================================================

@pytest.mark.big
def test_one_billion_rows_f64_add_mul(ctx):
    """a[i] = i, b[i] = n - i. a + b == n everywhere (sum == n^2, exact); a * 2.5 and a * b are checked on
    windows against numpy and through the verified sum kernel (linearity: sum(a * 2.5) == 2.5 * sum(a))."""
    n = 1_000_000_000
    a, b, out = ctx.alloc(n * 8), ctx.alloc(n * 8), ctx.alloc(n * 8)
    ctx.synth_iota("f64", a, n, 0)
    # b = n - a  (scalar on the left: Scalar - Array)
    ctx.apply_scalar("f64", "lhs", a, n, float(n), OPS["subtract"], b)
    ctx.apply("f64", a, b, OPS["add"], out, n, n)
    s, c = ctx.sum("f64", out, n)
    assert c == n and s == float(n) * float(n)
    for start in (0, 123_456_789, n - 4096):
        np.testing.assert_array_equal(out.download(np.float64, 4096, start * 8), np.full(4096, float(n)))
    ctx.apply_scalar("f64", "rhs", a, n, 2.5, OPS["multiply"], out)
    s, _ = ctx.sum("f64", out, n)
    exact = 5 * (n * (n - 1) // 2) / 2  # every product i * 2.5 is exact, so the sum is 2.5 * sum(i)
    assert abs(s - exact) <= math.ulp(exact)
    for start in (0, 500_000_000, n - 4096):
        i = np.arange(start, start + 4096, dtype=np.float64)
        np.testing.assert_array_equal(out.download(np.float64, 4096, start * 8), i * 2.5)
    ctx.apply("f64", a, b, OPS["multiply"], out, n, n)
    for start in (0, 500_000_000, n - 4096):
        i = np.arange(start, start + 4096, dtype=np.float64)
        np.testing.assert_array_equal(out.download(np.float64, 4096, start * 8), i * (n - i))
    for buf in (a, b, out):
        buf.free()


@pytest.mark.big
def test_more_than_2_to_32_rows_elementwise(ctx):
    """64-bit row indexing in the elementwise kernels: 2^32 + 1 000 003 u8 rows, array (+) scalar, checked on windows
    at both ends and across the 2^32 boundary (values are (i mod 251) so every window is predictable)."""
    n = (1 << 32) + 1_000_003
    src = ctx.alloc(n + 64)
    out = ctx.alloc(n + 64)
    # fill with a 251-periodic pattern by consolidating repeats of one 251 * 4096-byte block (exercises concat too)
    block = np.tile(np.arange(251, dtype=np.uint8), 4096)
    d_block = ctx.to_device(block)
    reps = n // block.size
    rem = n - reps * block.size
    chunks = [d_block] * reps + ([d_block] if rem else [])
    lens = [block.size] * reps + ([rem] if rem else [])
    ctx.consolidate_column(1, chunks, lens, src)
    ctx.apply_scalar("u8", "rhs", src, n, 7, OPS["add"], out)
    for start in (0, (1 << 32) - 1000, n - 5000):
        i = np.arange(start, start + 2000, dtype=np.int64)
        want = ((i % 251) + 7).astype(np.uint8)
        np.testing.assert_array_equal(out.download(np.uint8, 2000, start), want)
    src.free()
    out.free()


def test_output_bitmap_must_be_word_aligned(ctx):
    n = 100
    a = ctx.to_device(np.arange(n, dtype=np.int64), 64)
    out = ctx.alloc(n * 8)
    m = ctx.to_device(np.full(64, 0xFF, dtype=np.uint8))
    om = ctx.alloc(64)
    with pytest.raises(ffi.MinarrowHipError) as e:
        ctx.apply("i64", a, a, 0, out, n, n, mask=m, out_mask=om.offset(3))
    assert e.value.status == ffi.MA_ERR_INVALID_ARGUMENT
    with pytest.raises(ffi.MinarrowHipError) as e:
        ctx.apply("i64", a, a, 0, out, n, n, mask=m, out_mask=None)  # Some(mask) needs an output bitmap
    assert e.value.status == ffi.MA_ERR_INVALID_ARGUMENT
    with pytest.raises(ffi.MinarrowHipError) as e:
        ctx.apply("i64", a, a, 9, out, n, n)  # not an ArithmeticOperator
    assert e.value.status == ffi.MA_ERR_INVALID_ARGUMENT
    ctx.apply("i64", a, a, 0, out, n, n, mask=m.offset(3), mask_bit_offset=1, out_mask=om)  # input masks may be anywhere
    np.testing.assert_array_equal(out.download(np.int64, n), 2 * np.arange(n))
