"""GPU parity for ma_route_super_array_broadcast (src/kernels/broadcast/super_array.rs:180-251): the reference's
own vectors (tests/golden/routing_kat.json, super_array.rs:520-560) and the OR-union mask rule against the oracle."""
import json
from pathlib import Path

import numpy as np
import pytest

from minarrow_amd import ffi

pytestmark = pytest.mark.gpu
ROUTE = json.loads((Path(__file__).resolve().parent / "golden" / "routing_kat.json").read_text())


def nbytes(n):
    return ((n + 63) // 64) * 8


def test_ref_chunked_add(ctx):
    c = ROUTE["super_array"]["chunked_add"]
    lhs = [np.array(x, dtype=np.int32) for x in c["lhs_chunks"]]
    rhs = [np.array(x, dtype=np.int32) for x in c["rhs_chunks"]]
    outs = [np.zeros(3, dtype=np.int32) for _ in lhs]
    has = ctx.route_super_array_broadcast("i", 0, lhs, rhs, [3, 3], [3, 3], outs)
    assert has == [False, False]
    for o, e in zip(outs, c["expect_chunks"]):
        np.testing.assert_array_equal(o, e)
    with pytest.raises(ffi.MinarrowHipError) as e:
        ctx.route_super_array_broadcast("i", 0, lhs, rhs, [3, 3], [3, 2], outs)
    assert e.value.status == ffi.MA_ERR_LENGTH_MISMATCH and "Super Array broadcasting error" in e.value.message


@pytest.mark.parametrize("fmt,dt", [("l", np.int64), ("g", np.float64), ("i", np.int32)])
def test_mask_union_rule_and_many_chunks(ctx, oracle, fmt, dt):
    rng = np.random.default_rng(2)
    lens = [5, 64, 1000, 4097, 70_001, 0, 129]
    k = len(lens)
    lhs = [rng.integers(1, 100, size=n).astype(dt) for n in lens]
    rhs = [rng.integers(1, 100, size=n).astype(dt) for n in lens]
    lm = [rng.integers(0, 256, size=n // 8 + 16, dtype=np.uint8) if i % 2 == 0 else None for i, n in enumerate(lens)]
    rm = [rng.integers(0, 256, size=n // 8 + 16, dtype=np.uint8) if i % 3 != 1 else None for i, n in enumerate(lens)]
    d = lambda xs: [ctx.to_device(x, 64) if x is not None else None for x in xs]  # noqa: E731
    outs = [ctx.alloc(max(n, 1) * np.dtype(dt).itemsize + 64) for n in lens]
    oms = [ctx.alloc(nbytes(n) + 8) for n in lens]
    for op, name in ((2, "multiply"), (3, "divide")):
        has = ctx.route_super_array_broadcast(fmt, op, d(lhs), d(rhs), lens, lens, outs, d(lm), d(rm), oms)
        for i, n in enumerate(lens):
            if lm[i] is not None and rm[i] is not None:
                common = oracle.bitmask_union(oracle.pad_bits(lm[i], n), oracle.pad_bits(rm[i], n), n)  # OR, not AND
            else:
                common = lm[i] if lm[i] is not None else rm[i]
            assert has[i] == (common is not None)
            if n == 0:
                continue
            if common is None:
                fn = oracle.apply_float if np.dtype(dt).kind == "f" else oracle.apply_int
                st, want, _, _ = fn(oracle.aligned_copy(lhs[i]), oracle.aligned_copy(rhs[i]), name)
                np.testing.assert_array_equal(outs[i].download(dt, n), want)
            else:
                body = oracle.float_body if np.dtype(dt).kind == "f" else oracle.int_body
                st, want, want_mask = body("masked_std", lhs[i], rhs[i], name, mask=oracle.pad_bits(common, n))
                np.testing.assert_array_equal(outs[i].download(dt, n), want)
                np.testing.assert_array_equal(oms[i].download(np.uint8, nbytes(n)), want_mask[:nbytes(n)])
    # null_mask_override replaces every chunk's common mask
    n = lens[3]
    ov = rng.integers(0, 256, size=max(lens) // 8 + 16, dtype=np.uint8)
    has = ctx.route_super_array_broadcast(fmt, 0, d(lhs[3:4]), d(rhs[3:4]), [n], [n], outs[3:4], d(lm[3:4]), d(rm[3:4]), oms[3:4],
                                          override=ctx.to_device(ov, 16))
    assert has == [True]
    body = oracle.float_body if np.dtype(dt).kind == "f" else oracle.int_body
    st, want, want_mask = body("masked_std", lhs[3], rhs[3], "add", mask=oracle.pad_bits(ov, n))
    np.testing.assert_array_equal(outs[3].download(dt, n), want)


@pytest.mark.parametrize("aligned", [False, True])
@pytest.mark.parametrize("k,variant,max_len", [(700, 0, 9000), (700, 256, 9000), (700, 256 | 32, 9000), (700, 64, 9000),
                                               (9500, 0, 300), (9500, 128, 300), (9500, 1024, 300), (9500, 128 | 1024 | 64, 300)])
def test_many_small_chunks_one_launch(ctx, oracle, aligned, k, variant, max_len):
    """RechunkStrategy::Auto = 8192 rows (src/structs/chunked/super_array.rs:51-59): thousands of chunk pairs, odd
    lengths, mixed mask presence, run as one batched launch; compared chunk by chunk with the oracle. aligned: every
    chunk's output starts on a 16-byte boundary (the kernel then writes the output bitmaps itself); otherwise chunks are
    packed back to back, half of them start mid-vector, and the bitmaps come from the second launch.
    variant: 0 = the library's own choice (700 chunks: the tile-search kernel on an uploaded table; 9500: the
    chunk-per-workgroup kernel reading its table from pinned host memory — unless a masked chunk's output starts mid-vector,
    the unaligned case, which keeps the tile form); 256 / 128 force either form, + 32 = 8 x 16-byte tiles; 64 = output
    bitmaps by the second launch. 9500 chunks go in three SEGMENTS
    (4096 + 8192 + the rest: one table and one launch each, the host describing segment k + 1 while segment k runs);
    1024 = the same list as one segment."""
    rng = np.random.default_rng(12)
    lens = [int(x) for x in rng.integers(0, max_len, size=k)]
    lens[:4] = [8192, 8192, 1, 0]
    starts = []
    pos = 0
    for n in lens:
        starts.append(pos)
        pos += n + (n & 1 if aligned else 0)
    total = pos
    a = rng.integers(-1000, 1000, size=total).astype(np.int64)
    b = rng.integers(1, 1000, size=total).astype(np.int64)
    da, db, do = ctx.to_device(a, 64), ctx.to_device(b, 64), ctx.alloc(total * 8 + 64)
    offs = np.array(starts + [total], dtype=np.int64)
    # per-chunk bitmaps live in one device arena, each starting on an 8-byte boundary
    lm_host, rm_host, m_offs, pos = [], [], [], 0
    for i, n in enumerate(lens):
        nb = ((n + 63) // 64) * 8 + 8
        lm_host.append(rng.integers(0, 256, size=nb, dtype=np.uint8) if i % 3 != 0 else None)
        rm_host.append(rng.integers(0, 256, size=nb, dtype=np.uint8) if i % 4 == 1 else None)
        m_offs.append(pos)
        pos += nb
    arena_l = np.zeros(pos + 8, dtype=np.uint8)
    arena_r = np.zeros(pos + 8, dtype=np.uint8)
    for i in range(k):
        if lm_host[i] is not None:
            arena_l[m_offs[i]:m_offs[i] + lm_host[i].size] = lm_host[i]
        if rm_host[i] is not None:
            arena_r[m_offs[i]:m_offs[i] + rm_host[i].size] = rm_host[i]
    dl, dr, dom = ctx.to_device(arena_l), ctx.to_device(arena_r), ctx.alloc(pos + 8)
    lhs = [da.offset(int(offs[i]) * 8) for i in range(k)]
    rhs = [db.offset(int(offs[i]) * 8) for i in range(k)]
    outs = [do.offset(int(offs[i]) * 8) for i in range(k)]
    lms = [dl.offset(m_offs[i]) if lm_host[i] is not None else None for i in range(k)]
    rms = [dr.offset(m_offs[i]) if rm_host[i] is not None else None for i in range(k)]
    oms = [dom.offset(m_offs[i]) for i in range(k)]
    for op, name in ((2, "multiply"), (0, "add")):
        ctx.set_variant(variant)
        try:
            has = ctx.route_super_array_broadcast("l", op, lhs, rhs, lens, lens, outs, lms, rms, oms)
            if name == "add":  # and the dense form of the same call: no bitmaps at all
                dense_out = ctx.alloc(total * 8 + 64)
                douts = [dense_out.offset(int(offs[i]) * 8) for i in range(k)]
                assert not any(ctx.route_super_array_broadcast("l", op, lhs, rhs, lens, lens, douts))
                dense = dense_out.download(np.int64, total)
                for i, n in enumerate(lens):
                    sl = slice(int(offs[i]), int(offs[i]) + n)
                    np.testing.assert_array_equal(dense[sl], a[sl] + b[sl], err_msg=f"dense chunk {i}")
                dense_out.free()
        finally:
            ctx.set_variant(0)
        got = do.download(np.int64, total)
        got_masks = dom.download(np.uint8, pos)
        for i, n in enumerate(lens):
            sl = slice(int(offs[i]), int(offs[i]) + n)
            if lm_host[i] is not None and rm_host[i] is not None:
                common = oracle.bitmask_union(oracle.pad_bits(lm_host[i], n), oracle.pad_bits(rm_host[i], n), n)
            else:
                common = lm_host[i] if lm_host[i] is not None else rm_host[i]
            assert has[i] == (common is not None)
            if n == 0:
                continue
            if common is None:
                want = (a[sl] * b[sl]) if name == "multiply" else (a[sl] + b[sl])
                np.testing.assert_array_equal(got[sl], want)
            else:
                st, want, want_mask = oracle.int_body("masked_std", a[sl], b[sl], name, mask=oracle.pad_bits(common, n))
                np.testing.assert_array_equal(got[sl], want, err_msg=f"chunk {i}")
                nb = ((n + 63) // 64) * 8
                np.testing.assert_array_equal(got_masks[m_offs[i]:m_offs[i] + nb], want_mask[:nb], err_msg=f"chunk {i} validity")
    # dense division by zero in one chunk is reported
    zero_b = b.copy()
    zero_b[int(offs[5])] = 0
    db2 = ctx.to_device(zero_b, 64)
    rhs2 = [db2.offset(int(offs[i]) * 8) for i in range(k)]
    ctx.set_variant(variant)
    try:
        with pytest.raises(ffi.MinarrowHipError) as e:
            ctx.route_super_array_broadcast("l", 3, lhs, rhs2, lens, lens, outs)
    finally:
        ctx.set_variant(0)
    assert e.value.status == ffi.MA_ERR_DIVIDE_BY_ZERO


@pytest.mark.parametrize("fmt,dt", [("l", np.int64), ("i", np.int32), ("I", np.uint32), ("L", np.uint64)])
def test_masked_integer_division_with_zero_divisors(ctx, oracle, fmt, dt):
    """Masked integer Div / Rem / FloorDiv go chunk by chunk (validity depends on the data: a zero divisor clears its
    bit, simd.rs:319-326). Chunks carrying nulls on both sides use lhs.union(rhs) formed inside the kernel."""
    rng = np.random.default_rng(31)
    lens = [1, 63, 64, 65, 1000, 4097, 70_001]
    hi = 100 if np.dtype(dt).kind == "u" else 50
    lo = 0 if np.dtype(dt).kind == "u" else -50
    lhs = [rng.integers(lo, hi, size=n).astype(dt) for n in lens]
    rhs = [rng.integers(0, 4, size=n).astype(dt) for n in lens]  # a quarter of the divisors are zero
    lm = [rng.integers(0, 256, size=n // 8 + 16, dtype=np.uint8) if i % 3 != 1 else None for i, n in enumerate(lens)]
    rm = [rng.integers(0, 256, size=n // 8 + 16, dtype=np.uint8) if i % 3 != 2 else None for i, n in enumerate(lens)]
    d = lambda xs: [ctx.to_device(x, 64) if x is not None else None for x in xs]  # noqa: E731
    outs = [ctx.alloc(n * np.dtype(dt).itemsize + 64) for n in lens]
    oms = [ctx.alloc(nbytes(n) + 8) for n in lens]
    for op, name in ((3, "divide"), (4, "remainder"), (6, "floordiv")):
        has = ctx.route_super_array_broadcast(fmt, op, d(lhs), d(rhs), lens, lens, outs, d(lm), d(rm), oms)
        assert has == [True] * len(lens)
        for i, n in enumerate(lens):
            if lm[i] is not None and rm[i] is not None:
                common = oracle.bitmask_union(oracle.pad_bits(lm[i], n), oracle.pad_bits(rm[i], n), n)
            else:
                common = lm[i] if lm[i] is not None else rm[i]
            st, want, want_mask = oracle.int_body("masked_std", lhs[i], rhs[i], name, mask=oracle.pad_bits(common, n))
            np.testing.assert_array_equal(outs[i].download(dt, n), want, err_msg=f"{name} chunk {i}")
            np.testing.assert_array_equal(oms[i].download(np.uint8, nbytes(n)), want_mask[:nbytes(n)], err_msg=f"{name} chunk {i} validity")
