"""GPU parity tests for the bitmask kernels (src/kernels/bitmask/*), called through the C ABI.

Block 1 replays the reference's own unit tests (tests/golden/bitmask_kat.json, from src/kernels/bitmask/
simd.rs:797-955, std.rs:367-563, and merge_masks from src/kernels/arithmetic/mod.rs:401-409).
Block 2 compares with the CPU oracle on seeded random bitmaps, including windows with sub-byte / sub-word
offsets where the reference's addressing granularity matters. Everything is bit-exact.
"""
import json
from pathlib import Path

import numpy as np
import pytest

from minarrow_amd.host import live_variants

from minarrow_amd import ffi

pytestmark = pytest.mark.gpu

GOLD = Path(__file__).resolve().parent / "golden"
BITS = json.loads((GOLD / "bitmask_kat.json").read_text())
ARITH = json.loads((GOLD / "arithmetic_kat.json").read_text())


def pack(valid, pad=16):
    b = np.packbits(np.asarray(valid, dtype=bool), bitorder="little")
    return np.concatenate([b, np.zeros(pad, dtype=np.uint8)])


def unpack(bits, n, off=0):
    return np.unpackbits(np.ascontiguousarray(bits), bitorder="little")[off:off + n].astype(bool)


def nbytes(n):
    return ((n + 63) // 64) * 8


class G:
    def __init__(self, ctx):
        self.ctx = ctx

    def words(self, name, a, ao, b, bo, n):
        da, db = self.ctx.to_device(a, 16), self.ctx.to_device(b, 16)
        out = self.ctx.alloc(nbytes(n) + 8)
        self.ctx.mask_words_op(name, da, ao, db, bo, n, out)
        return out.download(np.uint8, nbytes(n))

    def unary(self, name, a, ao, n):
        da = self.ctx.to_device(a, 16)
        out = self.ctx.alloc(nbytes(n) + 8)
        self.ctx.mask_unary_op(name, da, ao, n, out)
        return out.download(np.uint8, nbytes(n))


@pytest.fixture(scope="module")
def g(ctx):
    return G(ctx)


# ---- 1. the reference's tests ---------------------------------------------------------------------------

def test_ref_simd_suite(g, ctx):
    s = BITS["simd_suite"]
    a, b = s["and_or_xor"]["a"], s["and_or_xor"]["b"]
    for name, fn in (("and_masks", np.logical_and), ("or_masks", np.logical_or), ("xor_masks", np.logical_xor)):
        np.testing.assert_array_equal(unpack(g.words(name, pack(a), 0, pack(b), 0, 8), 8), fn(a, b))
    np.testing.assert_array_equal(unpack(g.unary("not_mask", pack(s["not"]["a"]), 0, 4), 4), s["not"]["expect"])
    for c in s["in_mask"]:
        n = len(c["lhs"])
        if n:
            np.testing.assert_array_equal(unpack(g.words("in_mask", pack(c["lhs"]), 0, pack(c["rhs"]), 0, n), n), c["expect"])
        else:
            ctx.mask_words_op("in_mask", None, 0, None, 0, 0, None)  # empty window: OK, nothing written
    c = s["not_in_mask"]
    np.testing.assert_array_equal(unpack(g.words("not_in_mask", pack(c["lhs"]), 0, pack(c["rhs"]), 0, 4), 4), c["expect"])
    c = s["eq_ne"]
    np.testing.assert_array_equal(unpack(g.words("eq_mask", pack(c["a"]), 0, pack(c["b"]), 0, 4), 4), c["expect_eq"])
    np.testing.assert_array_equal(unpack(g.words("ne_mask", pack(c["a"]), 0, pack(c["b"]), 0, 4), 4), c["expect_ne"])
    c = s["all_eq"]
    pa = pack(c["a"])
    assert ctx.mask_all("all_eq", pa, 0, pa.copy(), 0, 8) is True
    flipped = list(c["a"])
    flipped[0] = not flipped[0]
    assert ctx.mask_all("all_eq", pa, 0, pack(flipped), 0, 8) is False
    c = s["all_ne"]
    assert ctx.mask_all("all_ne", pack(c["a"]), 0, pack(c["b"]), 0, 3) is True
    assert ctx.mask_all("all_ne", pack(c["a"]), 0, pack(c["a"]), 0, 3) is False
    assert ctx.popcount_mask(pack(s["popcount"]["a"]), 0, 8) == s["popcount"]["expect"]
    for lanes in (8, 16, 32, 64):
        n = 64 * lanes
        ones = pack(np.ones(n, dtype=bool))
        assert ctx.all_true_mask(ones, n) and not ctx.all_false_mask(ones, n)
        ones[0] &= ~np.uint8(1 << 3)
        assert not ctx.all_true_mask(ones, n)
        assert ctx.all_false_mask(pack(np.zeros(n, dtype=bool)), n)


def test_ref_std_suite(g, ctx):
    s = BITS["std_suite"]
    for name in ("and", "or", "xor"):
        c = s[name]
        n = len(c["a"])
        np.testing.assert_array_equal(unpack(g.words(f"{name}_masks", pack(c["a"]), 0, pack(c["b"]), 0, n), n), c["expect"])
    for c in s["in_mask"]:
        np.testing.assert_array_equal(unpack(g.words("in_mask", pack(c["lhs"]), 0, pack(c["rhs"]), 0, 3), 3), c["expect"])
    c = s["not_in_mask"]
    np.testing.assert_array_equal(unpack(g.words("not_in_mask", pack(c["lhs"]), 0, pack(c["rhs"]), 0, 2), 2), c["expect"])
    np.testing.assert_array_equal(unpack(g.words("eq_mask", pack(s["eq"]["a"]), 0, pack(s["eq"]["b"]), 0, 3), 3), s["eq"]["expect"])
    np.testing.assert_array_equal(unpack(g.words("ne_mask", pack(s["ne"]["a"]), 0, pack(s["ne"]["b"]), 0, 3), 3), s["ne"]["expect"])
    for c in s["all_eq"]:
        assert ctx.mask_all("all_eq", pack(c["a"]), 0, pack(c["b"]), 0, len(c["a"])) == c["expect"]
    for c in s["all_ne"]:
        assert ctx.mask_all("all_ne", pack(c["a"]), 0, pack(c["b"]), 0, len(c["a"])) == c["expect"]
    assert ctx.popcount_mask(pack(s["popcount"]["a"]), 0, 6) == s["popcount"]["expect"]
    for c in s["all_true"]:
        assert ctx.all_true_mask(pack(c["a"]), len(c["a"])) == c["expect"]
    for c in s["all_false"]:
        assert ctx.all_false_mask(pack(c["a"]), len(c["a"])) == c["expect"]
    # clear_trailing_bits: len 9 with byte 1 forced to 0xFF -> only bit 8 survives an op's output
    src = np.array([0xFF, 0xFF] + [0] * 14, dtype=np.uint8)
    out = g.words("and_masks", src, 0, src, 0, 9)
    assert out[1] == s["clear_trailing_bits"]["expect_byte1_after"] and out[0] == 0xFF


def test_ref_merge_masks(ctx):
    """merge_masks_correctness — src/kernels/arithmetic/mod.rs:401-409"""
    c = ARITH["merge_masks"]
    out = np.zeros(16, dtype=np.uint8)
    assert ctx.merge_bitmasks(pack(c["a"]), pack(c["b"]), 4, out) is True
    np.testing.assert_array_equal(unpack(out, 4), c["expect"])
    assert ctx.merge_bitmasks(None, None, 4, out) is False
    assert ctx.merge_bitmasks(pack(c["a"]), None, 4, out) is True
    np.testing.assert_array_equal(unpack(out, 4), c["a"])


# ---- 2. random parity with the oracle, including the reference's window-granularity rules ---------------------

LENS = [1, 7, 8, 63, 64, 65, 127, 128, 129, 1000, 4096, 100_003]


@pytest.mark.parametrize("name,op", [("and_masks", "and"), ("or_masks", "or"), ("xor_masks", "xor")])
def test_binop_random(g, oracle, name, op):
    rng = np.random.default_rng(1)
    for n in LENS:
        for lo, ro in ((0, 0), (8, 16), (64, 128), (3, 0), (13, 70), (9, 9)):
            a = rng.integers(0, 256, size=(n + max(lo, ro)) // 8 + 24, dtype=np.uint8)
            b = rng.integers(0, 256, size=a.size, dtype=np.uint8)
            want = oracle.bitmask_binop(op, a, lo, b, ro, n)[:nbytes(n)]
            np.testing.assert_array_equal(g.words(name, a, lo, b, ro, n), want, err_msg=f"{op} n={n} lo={lo} ro={ro}")
            if lo % 8 == 0 and ro % 8 == 0:  # byte-aligned windows are plain bit-for-bit logic
                fn = {"and": np.logical_and, "or": np.logical_or, "xor": np.logical_xor}[op]
                np.testing.assert_array_equal(unpack(want, n), fn(unpack(a, n, lo), unpack(b, n, ro)))


def test_not_and_slice_random(g, oracle):
    rng = np.random.default_rng(2)
    for n in LENS:
        for off in (0, 8, 64, 5, 77):
            a = rng.integers(0, 256, size=(n + off) // 8 + 24, dtype=np.uint8)
            np.testing.assert_array_equal(g.unary("not_mask", a, off, n), oracle.bitmask_not(a, off, n)[:nbytes(n)])
            sl = g.unary("bitmask_slice", a, off, n)
            np.testing.assert_array_equal(unpack(sl, n), unpack(a, n, off))
            assert not unpack(sl, nbytes(n) * 8 - n, n).any()  # trailing bits are zero


@pytest.mark.parametrize("n", [256, 8191, 64 * 64 * 2 + 1, (1 << 18) + 64, (1 << 22) + 77, 3 * (1 << 21) + 12345])
def test_slice_at_every_sub_byte_offset_through_the_vector_path(g, n):
    """bitmask_slice re-bases a window that starts at ANY bit. From 256 bits on the kernel moves word pairs (16-byte
    accesses) and funnels the 1..7 leftover bits in registers — across lanes within a wave instruction, across the
    accesses of a wave's run, and into the scalar tail: sizes around each of those seams, every sub-byte offset, checked
    bit for bit against numpy."""
    rng = np.random.default_rng(n)
    a = rng.integers(0, 256, size=n // 8 + 64, dtype=np.uint8)
    bits = np.unpackbits(a, bitorder="little")
    for off in (1, 2, 3, 4, 5, 6, 7, 13, 63, 64 + 9, 8 * 5 + 3):
        sl = g.unary("bitmask_slice", a, off, n)
        np.testing.assert_array_equal(unpack(sl, n), bits[off:off + n].astype(bool), err_msg=f"n={n} off={off}")
        assert not unpack(sl, nbytes(n) * 8 - n, n).any()  # trailing bits are zero


def test_popcount_and_predicates_random(ctx, oracle):
    rng = np.random.default_rng(3)
    for n in LENS + [10_000_019]:
        a = rng.integers(0, 256, size=n // 8 + 40, dtype=np.uint8)
        d = ctx.to_device(a, 16)
        for off in (0, 64, 128, 5, 70):
            assert ctx.popcount_mask(d, off, n) == oracle.bitmask_popcount(a, off, n)
        assert ctx.popcount_mask(d, 0, n) == int(unpack(a, n).sum())
        assert ctx.all_true_mask(d, n) == oracle.all_true(a, n, None) == bool(unpack(a, n).all())
        assert ctx.all_false_mask(d, n) == oracle.all_false(a, n, None)
        ones = np.full(n // 8 + 40, 0xFF, dtype=np.uint8)
        assert ctx.all_true_mask(ones, n) and not ctx.all_false_mask(ones, n)
        ones[(n - 1) >> 3] &= ~np.uint8(1 << ((n - 1) & 7))  # clear the last logical bit only
        assert not ctx.all_true_mask(ones, n)
        zeros = np.zeros(n // 8 + 40, dtype=np.uint8)
        zeros[n >> 3] |= np.uint8(1 << (n & 7))  # a set bit just past the end does not count
        assert ctx.all_false_mask(zeros, n)


def test_eq_ne_all_eq_random(g, ctx, oracle):
    rng = np.random.default_rng(4)
    for n in LENS:
        a = rng.integers(0, 256, size=n // 8 + 40, dtype=np.uint8)
        b = a.copy()
        for ao, bo in ((0, 0), (64, 128)):
            flips = rng.integers(0, n, size=3)
            b2 = b.copy()
            for f in flips:
                b2[(bo + f) >> 3] ^= np.uint8(1 << ((bo + f) & 7))
            panics, want = oracle.bitmask_eq(a, ao, b2, bo, n)
            assert not panics
            np.testing.assert_array_equal(g.words("eq_mask", a, ao, b2, bo, n), want[:nbytes(n)])
            _, want_ne = oracle.bitmask_eq(a, ao, b2, bo, n, negate=True)
            np.testing.assert_array_equal(g.words("ne_mask", a, ao, b2, bo, n), want_ne[:nbytes(n)])
            assert ctx.mask_all("all_eq", a, ao, b2, bo, n) == bool(oracle.bitmask_all_eq(a, ao, b2, bo, n))
            assert ctx.mask_all("all_ne", a, ao, b2, bo, n) == bool(oracle.bitmask_all_ne(a, ao, b2, bo, n))
        assert ctx.mask_all("all_eq", a, 0, a.copy(), 0, n)
    # the reference panics on offsets that are not multiples of 64
    a = np.zeros(64, dtype=np.uint8)
    assert oracle.bitmask_eq(a, 3, a, 0, 100)[0]
    for name in ("eq_mask", "ne_mask"):
        with pytest.raises(ffi.MinarrowHipError) as e:
            g.words(name, a, 3, a, 0, 100)
        assert e.value.status == ffi.MA_ERR_INVALID_ARGUMENT
    assert oracle.bitmask_all_eq(a, 3, a, 0, 100) == -1
    with pytest.raises(ffi.MinarrowHipError):
        ctx.mask_all("all_eq", a, 3, a, 0, 100)
    assert ctx.mask_all("all_eq", a, 3, a, 0, 20) == bool(oracle.bitmask_all_eq(a, 3, a, 0, 20))  # len < 64: allowed


def test_in_not_in_random(g, oracle):
    rng = np.random.default_rng(5)
    for n in (5, 64, 65, 1000, 70_001):
        lhs = rng.integers(0, 256, size=n // 8 + 40, dtype=np.uint8)
        for kind in ("both", "ones", "zeros"):
            rhs = {"both": rng.integers(0, 256, size=lhs.size, dtype=np.uint8),
                   "ones": np.full(lhs.size, 0xFF, dtype=np.uint8), "zeros": np.zeros(lhs.size, dtype=np.uint8)}[kind]
            for lo, ro in ((0, 0), (3, 64), (64, 0)):
                want = oracle.bitmask_in(lhs, lo, rhs, ro, n)[:nbytes(n)]
                np.testing.assert_array_equal(g.words("in_mask", lhs, lo, rhs, ro, n), want, err_msg=f"{kind} {n} {lo} {ro}")
                want = oracle.bitmask_not_in(lhs, lo, rhs, ro, n)[:nbytes(n)]
                np.testing.assert_array_equal(g.words("not_in_mask", lhs, lo, rhs, ro, n), want)


def test_merge_random(ctx, oracle):
    rng = np.random.default_rng(6)
    for n in LENS:
        a = rng.integers(0, 256, size=n // 8 + 24, dtype=np.uint8)
        b = rng.integers(0, 256, size=n // 8 + 24, dtype=np.uint8)
        out = ctx.alloc(nbytes(n) + 8)
        assert ctx.merge_bitmasks(ctx.to_device(a, 16), ctx.to_device(b, 16), n, out)
        np.testing.assert_array_equal(out.download(np.uint8, nbytes(n)), oracle.merge_bitmasks(a, b, n)[:nbytes(n)])


@pytest.mark.parametrize("tag,dt", [("u8", np.uint8), ("u16", np.uint16), ("u32", np.uint32), ("u64", np.uint64)])
def test_simd_eq_mask_random(ctx, oracle, tag, dt):
    rng = np.random.default_rng(7)
    for n in (1, 63, 64, 65, 1000, 100_003):
        data = rng.integers(0, 16, size=n).astype(dt)
        out = ctx.alloc(nbytes(n) + 8)
        ctx.simd_eq_mask(tag, ctx.to_device(data, 64), n, 0x7, 0x3, out)
        got = out.download(np.uint8, nbytes(n))
        np.testing.assert_array_equal(got, oracle.simd_eq_mask(data, 0x7, 0x3)[:nbytes(n)])
        np.testing.assert_array_equal(unpack(got, n), (data & 0x7) == 0x3)
    # data that does not start on a 16-byte boundary (ballot path) and a host-resident column, > 1 vector tile
    n = 300_001
    data = rng.integers(0, 256, size=n + 1).astype(dt)
    dev = ctx.to_device(data, 64)
    out = ctx.alloc(nbytes(n) + 8)
    ctx.simd_eq_mask(tag, dev.ptr + data.itemsize, n, 0xF, 0x9, out)
    np.testing.assert_array_equal(unpack(out.download(np.uint8, nbytes(n)), n), (data[1:] & 0xF) == 0x9)
    host_out = np.zeros(nbytes(n) + 8, dtype=np.uint8)
    ctx.simd_eq_mask(tag, data[:n].copy(), n, 0xF, 0x9, host_out)
    np.testing.assert_array_equal(unpack(host_out, n), (data[:n] & 0xF) == 0x9)


@pytest.mark.parametrize("tag,dt", [("u8", np.uint8), ("u16", np.uint16), ("u32", np.uint32), ("u64", np.uint64)])
@pytest.mark.parametrize("grid", [0, 1, 3])
def test_simd_eq_mask_full_range_many_trips(ctx, tag, dt, grid):
    """Full-range elements (the 1- and 2-byte types compare four / two rows per 32-bit word: top bits, 0x7F / 0x80 / 0xFF
    bytes and carries between neighbours must not leak), field masks with high bits, and grids of 1 and 3 workgroups so that
    every wave makes many trips — its result words are staged in LDS and stored one trip late, the last trip after the
    loop. Both load depths (variant 2048 = 4 loads per lane)."""
    rng = np.random.default_rng(ord(tag[1]) + grid)
    info = np.iinfo(dt)
    n = 20 * 64 * (16 // np.dtype(dt).itemsize) * 8 * 4 + 777  # 20 tiles of the 8-deep shape + a ragged tail
    data = rng.integers(0, info.max, size=n, dtype=dt, endpoint=True)
    special = np.array([0, 1, 0x7F, 0x80, 0xFF, info.max, info.max >> 1, (info.max >> 1) + 1], dtype=dt)
    data[rng.integers(0, n, size=n // 3)] = special[rng.integers(0, special.size, size=n // 3)]
    dev = ctx.to_device(data, 64)
    out = ctx.alloc(nbytes(n) + 8)
    for fm, tg in ((info.max, 0x80), (info.max, info.max), (0x80, 0x80), (info.max ^ 1, 0x7E), (0xFF, 0), (info.max, 0)):
        fm, tg = dt(fm & info.max), dt(tg & info.max)
        for variant in live_variants((0, 2048)):  # 2048: a tuning form (tuning build only)
            ctx.set_variant(variant)
            ctx.set_grid(grid)
            try:
                ctx.dev_memset(out, 0xA5, nbytes(n) + 8)
                ctx.simd_eq_mask(tag, dev, n, int(fm), int(tg), out)
            finally:
                ctx.set_grid(0)
                ctx.set_variant(0)
            got = unpack(out.download(np.uint8, nbytes(n)), n)
            np.testing.assert_array_equal(got, (data & fm) == tg, err_msg=f"mask {fm:#x} target {tg:#x} variant {variant}")


def test_one_billion_bit_masks(ctx):
    """Config-4-sized validity (10^9 bits, ~10 % nulls): popcount == the masked sum's valid count;
    NOT flips exactly; AND with its own NOT is empty (size-independent properties)."""
    n = 1_000_000_000
    m = ctx.alloc(n // 8 + 64)
    inv = ctx.alloc(n // 8 + 64)
    both = ctx.alloc(n // 8 + 64)
    ctx.synth_validity(m, n, seed=0xC0FFEE, null_every=10)
    pop = ctx.popcount_mask(m, 0, n)
    assert 0.899 * n < pop < 0.901 * n
    data = ctx.alloc(n * 8)
    ctx.synth_iota("i64", data, n, 0)
    assert ctx.sum("i64", data, n, mask=m)[1] == pop
    data.free()
    ctx.mask_unary_op("not_mask", m, 0, n, inv)
    assert ctx.popcount_mask(inv, 0, n) == n - pop
    ctx.mask_words_op("and_masks", m, 0, inv, 0, n, both)
    assert ctx.all_false_mask(both, n) and ctx.popcount_mask(both, 0, n) == 0
    ctx.mask_words_op("or_masks", m, 0, inv, 0, n, both)
    assert ctx.all_true_mask(both, n)
    assert ctx.mask_all("all_ne", m, 0, inv, 0, n) and not ctx.mask_all("all_eq", m, 0, inv, 0, n)


def test_bitmask_struct_vectors(ctx, oracle):
    """The reference's Bitmask struct tests (src/structs/bitmask.rs:919-1108) through the C ABI."""
    import json
    from pathlib import Path

    k = json.loads((Path(__file__).resolve().parent / "golden" / "bitmask_struct_kat.json").read_text())

    def dev(bools):
        return ctx.to_device(oracle.pad_bits(oracle.pack_bits(bools), len(bools)), 16)

    for c in k["count_and_all"]["cases"]:
        n = len(c["bits"])
        d = dev(c["bits"])
        assert ctx.popcount_mask(d, 0, n) == c["count_ones"]
        assert ctx.all_true_mask(d, n) == c["all_set"] and ctx.all_false_mask(d, n) == c["all_unset"]
    c = k["invert_union_intersect"]
    a, b, out = dev(c["a"]), dev(c["b"]), ctx.alloc(64)
    for name, key in (("or_masks", "union"), ("and_masks", "intersect")):
        ctx.mask_words_op(name, a, 0, b, 0, 8, out)
        assert unpack(out.download(np.uint8, 8), 8).tolist() == c[key]
    ctx.mask_unary_op("not_mask", a, 0, 8, out)
    assert unpack(out.download(np.uint8, 8), 8).tolist() == c["invert_a"]
    c = k["union_opt"]
    ctx.mask_words_op("or_masks", dev(c["a"]), 0, dev(c["b"]), 0, 4, out)
    assert unpack(out.download(np.uint8, 8), 4).tolist() == c["expect"]
    c = k["slice_clone"]
    ctx.mask_unary_op("bitmask_slice", dev(c["bits"]), c["offset"], c["len"], out)
    assert unpack(out.download(np.uint8, 8), c["len"]).tolist() == c["expect"]
    c = k["concatenate"]
    assert ctx.consolidate_boolean_column([(dev(c["m1"]), 0, 5), (dev(c["m2"]), 0, 4)], out) is False
    assert unpack(out.download(np.uint8, 8), 9).tolist() == c["expect"]


def test_scans_back_to_back_leave_no_state_behind(ctx):
    """The scan kernels keep their accumulator and ticket on the device between launches (the last workgroup to arrive
    zeroes both): thousands of scans of different sizes, grids and answers in a row, from two threads sharing the
    context (lanes have their own scratch), must each return their own result."""
    import threading

    rng = np.random.default_rng(77)
    cases = []
    for n in (1, 63, 64, 1000, 65_537, 1 << 20, (1 << 23) + 5):
        bits = rng.integers(0, 256, size=n // 8 + 16, dtype=np.uint8)
        want = int(np.unpackbits(bits, bitorder="little")[:n].sum())
        ones = np.full(n // 8 + 16, 0xFF, dtype=np.uint8)
        cases.append((n, ctx.to_device(bits, 16), want, ctx.to_device(ones, 16)))
    errors = []

    def work(seed):
        try:
            order = np.random.default_rng(seed).integers(0, len(cases), size=1500)
            for k in order:
                n, dbits, want, dones = cases[int(k)]
                assert ctx.popcount_mask(dbits, 0, n) == want
                assert ctx.all_true_mask(dones, n) is True
                assert ctx.all_false_mask(dones, n) is False
                assert ctx.popcount_mask(dones, 0, n) == n
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(s,)) for s in (1, 2)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors[:3]
