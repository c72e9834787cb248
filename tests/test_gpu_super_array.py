"""GPU parity for ma_route_super_array_broadcast (src/kernels/broadcast/super_array.rs:180-251): the reference's
own vectors (tests/golden/routing_kat.json, super_array.rs:520-560) and the OR-union mask rule against the oracle."""
import json
from pathlib import Path

import numpy as np
import pytest

from minarrow_amd.host import live_variants

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
@pytest.mark.parametrize("k,variant,max_len", [p for p in [(700, 0, 9000), (700, 256, 9000), (700, 256 | 32, 9000), (700, 64, 9000),
                                                          (9500, 0, 300), (9500, 128, 300), (9500, 1024, 300), (9500, 128 | 1024 | 64, 300)]
                                               if live_variants([p[1]])])  # 64 and 1024 are tuning forms (tuning build only)
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


def test_ref_scalar_to_superarray(ctx):
    """test_scalar_to_superarray (src/kernels/broadcast/scalar.rs:1119-1150): 10 + [[1,2,3],[4,5,6]]; test_scalar_to_superarrayview
    (:1154-1193): 5 * the two 3-row slices of [10..60]."""
    chunks = [np.array([1, 2, 3], dtype=np.int32), np.array([4, 5, 6], dtype=np.int32)]
    outs = [np.zeros(3, dtype=np.int32) for _ in chunks]
    assert ctx.broadcast_super_array_scalar("i", 0, np.int32(10), chunks, [3, 3], outs, scalar_is_lhs=True) == [False, False]
    np.testing.assert_array_equal(outs[0], [11, 12, 13])
    np.testing.assert_array_equal(outs[1], [14, 15, 16])
    arr = np.array([10, 20, 30, 40, 50, 60], dtype=np.int32)
    d = ctx.to_device(arr, 64)
    do = ctx.alloc(6 * 4 + 64)
    ctx.broadcast_super_array_scalar("i", 2, np.int32(5), [d, d.offset(12)], [3, 3], [do, do.offset(12)], scalar_is_lhs=True)
    np.testing.assert_array_equal(do.download(np.int32, 6), [50, 100, 150, 200, 250, 300])


@pytest.mark.parametrize("fmt,dt", [("i", np.int32), ("I", np.uint32), ("l", np.int64), ("L", np.uint64), ("f", np.float32),
                                    ("g", np.float64)])
@pytest.mark.parametrize("k,variant,max_len", [(40, 0, 70_000), (3000, 256, 600), (3000, 128, 600), (9500, 0, 300)])
def test_super_array_scalar_both_sides(ctx, oracle, fmt, dt, k, variant, max_len):
    """SuperArray (op) Scalar and Scalar (op) SuperArray (super_array.rs:87-116, scalar.rs:214-243) in one launch against the
    oracle's array kernels on (chunk, scalar broadcast to the chunk's length) — what maybe_broadcast_scalar_array makes of
    the reference's length-1 operand. Ragged chunk lengths packed back to back (outputs start mid-vector), every operator,
    both sides, both kernel forms; then with validity on two thirds of the chunks (the ABI's optional gating)."""
    rng = np.random.default_rng(k + variant)
    dt = np.dtype(dt)
    lens = [int(x) for x in rng.integers(0, max_len, size=k)]
    lens[:3] = [8192, 1, 0]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    total = int(offs[-1])
    if dt.kind == "f":
        a = (rng.standard_normal(total) * 10).astype(dt)
        scalars = [dt.type(2.5), dt.type(-0.75)]
    else:
        a = rng.integers(1 if dt.kind == "u" else -1000, 1000, size=total).astype(dt)
        a[a == 0] = 1  # the column is also the divisor (scalar / chunk): a dense zero there is an error, tested below
        scalars = [dt.type(7), dt.type(3)]
    da, do = ctx.to_device(a, 64), ctx.alloc(total * dt.itemsize + 64)
    chunks = [da.offset(int(offs[i]) * dt.itemsize) for i in range(k)]
    outs = [do.offset(int(offs[i]) * dt.itemsize) for i in range(k)]
    fn = oracle.apply_float if dt.kind == "f" else oracle.apply_int
    names = ["add", "subtract", "multiply", "divide", "remainder", "power", "floordiv"]
    ops = [0, 1, 2, 3, 4, 6] if dt.kind != "f" else [0, 1, 2, 3, 5]
    ctx.set_variant(variant)
    try:
        for op in ops:
            for side, s in ((False, scalars[0]), (True, scalars[1])):
                has = ctx.broadcast_super_array_scalar(fmt, op, s, chunks, lens, outs, scalar_is_lhs=side)
                assert not any(has)
                got = do.download(dt, total)
                full = np.full(total, s, dtype=dt)
                l, r = (full, a) if side else (a, full)
                st, want, _, _ = fn(oracle.aligned_copy(l), oracle.aligned_copy(r), names[op])
                if dt.kind == "f" and op == 5:
                    ok = np.isclose(got, want, rtol=1e-5 if dt.itemsize == 4 else 1e-12, equal_nan=True)
                    assert ok.all()
                else:
                    np.testing.assert_array_equal(got, want, err_msg=f"op {names[op]} scalar_is_lhs={side}")
    finally:
        ctx.set_variant(0)
    if dt.kind != "f":  # dense integer division by a zero scalar is reported, like a zero in a dense divisor chunk
        with pytest.raises(ffi.MinarrowHipError) as e:
            ctx.broadcast_super_array_scalar(fmt, 3, dt.type(0), chunks, lens, outs)
        assert e.value.status == ffi.MA_ERR_DIVIDE_BY_ZERO
    # optional validity per chunk: gates like the mask argument of the array kernels, and is copied to the chunk's output bitmap
    masks_h, m_offs, pos = [], [], 0
    for i, n in enumerate(lens):
        nb = ((n + 63) // 64) * 8 + 8
        masks_h.append(rng.integers(0, 256, size=nb, dtype=np.uint8) if i % 3 != 0 else None)
        m_offs.append(pos)
        pos += nb
    arena = np.zeros(pos + 8, dtype=np.uint8)
    for i in range(k):
        if masks_h[i] is not None:
            arena[m_offs[i]:m_offs[i] + masks_h[i].size] = masks_h[i]
    dm, dom = ctx.to_device(arena), ctx.alloc(pos + 8)
    ms = [dm.offset(m_offs[i]) if masks_h[i] is not None else None for i in range(k)]
    oms = [dom.offset(m_offs[i]) for i in range(k)]
    # outputs of masked chunks must start on 16-byte boundaries for the chunk form: use per-chunk aligned outputs
    starts, p = [], 0
    for n in lens:
        starts.append(p)
        p += (n * dt.itemsize + 15) // 16 * 16
    do2 = ctx.alloc(p + 64)
    outs2 = [do2.offset(starts[i]) for i in range(k)]
    body = oracle.float_body if dt.kind == "f" else oracle.int_body
    ctx.set_variant(variant)
    try:
        for op, name, side in ((2, "multiply", False), (1, "subtract", True), (3, "divide", False)):
            s = scalars[0]
            has = ctx.broadcast_super_array_scalar(fmt, op, s, chunks, lens, outs2, ms, oms, scalar_is_lhs=side)
            got_masks = dom.download(np.uint8, pos)
            for i in list(range(0, min(k, 60))) + list(range(60, k, 97)):
                n = lens[i]
                assert has[i] == (masks_h[i] is not None)
                if n == 0:
                    continue
                sl = slice(int(offs[i]), int(offs[i]) + n)
                got = do2.download(dt, n, starts[i])
                full = np.full(n, s, dtype=dt)
                l, r = (full, a[sl]) if side else (a[sl], full)
                if masks_h[i] is None:
                    st, want, _, _ = fn(oracle.aligned_copy(l), oracle.aligned_copy(r), name)
                    np.testing.assert_array_equal(got, want)
                else:
                    st, want, want_mask = body("masked_std", l, r, name, mask=oracle.pad_bits(masks_h[i], n))
                    np.testing.assert_array_equal(got, want, err_msg=f"{name} chunk {i}")
                    nb = ((n + 63) // 64) * 8
                    np.testing.assert_array_equal(got_masks[m_offs[i]:m_offs[i] + nb], want_mask[:nb], err_msg=f"{name} chunk {i} validity")
    finally:
        ctx.set_variant(0)


@pytest.mark.parametrize("fmt,dt", [("i", np.int32), ("l", np.int64), ("I", np.uint32), ("L", np.uint64)])
@pytest.mark.parametrize("k,variant,max_len", [(2500, 0, 9000), (2500, 256, 9000), (2500, 128, 9000), (5000, 0, 600), (5000, 256 | 32, 600)])
def test_masked_integer_division_over_chunk_lists_in_one_launch(ctx, oracle, fmt, dt, k, variant, max_len):
    """Masked integer Div / Rem / FloorDiv over thousands of chunk pairs: the output validity depends on the data (a zero
    divisor clears the row's bit, simd.rs:319-326), and the batched kernels produce it themselves — full runs pack the result
    bits of their lanes, a chunk's ragged last run goes row by row with one ballot per 64 rows. Ragged lengths, nulls on
    one or both sides or neither (a dense chunk with a zero divisor would raise: dense chunks get non-zero divisors here),
    a quarter of the masked chunks' divisors zero; both kernel forms; the scalar forms on both sides; and one call whose
    masked outputs start mid-vector (the chunk-by-chunk path)."""
    rng = np.random.default_rng(k + variant + ord(fmt))
    dt = np.dtype(dt)
    lens = [int(x) for x in rng.integers(0, max_len, size=k)]
    lens[:4] = [8192, 4096, 1, 0]
    lo, hi = (0, 100) if dt.kind == "u" else (-50, 50)
    lhs = [rng.integers(lo, hi, size=n).astype(dt) for n in lens]
    mb = lambda n: ((n + 63) // 64) * 8 + 8  # noqa: E731 — bytes of a chunk's bitmap slot
    lm = [rng.integers(0, 256, size=mb(n), dtype=np.uint8) if i % 3 != 1 else None for i, n in enumerate(lens)]
    rm = [rng.integers(0, 256, size=mb(n), dtype=np.uint8) if i % 4 == 0 else None for i, n in enumerate(lens)]
    rhs = [(rng.integers(0, 4, size=n) if (lm[i] is not None or rm[i] is not None) else rng.integers(1, 5, size=n)).astype(dt)
           for i, n in enumerate(lens)]
    # every chunk's buffers start on 16-byte boundaries inside shared arenas
    starts, p = [], 0
    for n in lens:
        starts.append(p)
        p += (n * dt.itemsize + 15) // 16 * 16
    arena_l, arena_r = np.zeros(p + 64, dtype=np.uint8), np.zeros(p + 64, dtype=np.uint8)
    m_starts, q = [], 0
    for n in lens:
        m_starts.append(q)
        q += ((n + 63) // 64) * 8 + 8
    am_l, am_r = np.zeros(q + 8, dtype=np.uint8), np.zeros(q + 8, dtype=np.uint8)
    for i, n in enumerate(lens):
        arena_l[starts[i]:starts[i] + n * dt.itemsize] = lhs[i].view(np.uint8)
        arena_r[starts[i]:starts[i] + n * dt.itemsize] = rhs[i].view(np.uint8)
        if lm[i] is not None:
            am_l[m_starts[i]:m_starts[i] + lm[i].size] = lm[i]
        if rm[i] is not None:
            am_r[m_starts[i]:m_starts[i] + rm[i].size] = rm[i]
    dl, dr, do = ctx.to_device(arena_l, 64), ctx.to_device(arena_r, 64), ctx.alloc(p + 128)
    dml, dmr, dom = ctx.to_device(am_l, 16), ctx.to_device(am_r, 16), ctx.alloc(q + 64)
    L = [dl.offset(s) for s in starts]
    R = [dr.offset(s) for s in starts]
    O = [do.offset(s) for s in starts]
    LM = [dml.offset(m_starts[i]) if lm[i] is not None else None for i in range(k)]
    RM = [dmr.offset(m_starts[i]) if rm[i] is not None else None for i in range(k)]
    OM = [dom.offset(s) for s in m_starts]

    def common(i, n):
        if lm[i] is not None and rm[i] is not None:
            return oracle.bitmask_union(oracle.pad_bits(lm[i], n), oracle.pad_bits(rm[i], n), n)
        return lm[i] if lm[i] is not None else rm[i]

    def check(name, get_l, get_r, has, sample):
        got, got_masks = do.download(np.uint8, p), dom.download(np.uint8, q)
        for i in sample:
            n = lens[i]
            c = common(i, n)
            assert has[i] == (c is not None)
            if n == 0:
                continue
            g = got[starts[i]:starts[i] + n * dt.itemsize].view(dt)
            if c is None:
                st, want, _, _ = oracle.apply_int(oracle.aligned_copy(get_l(i)), oracle.aligned_copy(get_r(i)), name)
                np.testing.assert_array_equal(g, want, err_msg=f"{name} dense chunk {i}")
            else:
                st, want, want_mask = oracle.int_body("masked_std", get_l(i), get_r(i), name, mask=oracle.pad_bits(c, n))
                np.testing.assert_array_equal(g, want, err_msg=f"{name} chunk {i}")
                nb = ((n + 63) // 64) * 8
                np.testing.assert_array_equal(got_masks[m_starts[i]:m_starts[i] + nb], want_mask[:nb], err_msg=f"{name} chunk {i} validity")

    sample = list(range(0, 40)) + list(range(40, k, 61))
    ctx.set_variant(variant)
    try:
        for op, name in ((3, "divide"), (4, "remainder"), (6, "floordiv")):
            ctx.dev_memset(dom, 0xA5, q + 64)
            has = ctx.route_super_array_broadcast(fmt, op, L, R, lens, lens, O, LM, RM, OM)
            check(name, lambda i: lhs[i], lambda i: rhs[i], has, sample)
        # chunk / scalar and scalar / chunk with the chunks' own validity (the ABI's optional gating); a zero scalar divisor
        # nulls every row of a masked chunk
        for side, s in ((False, dt.type(3)), (True, dt.type(7))):
            cols = rhs if side else lhs  # scalar / chunk divides by the chunk: the masked chunks' rhs has zeros
            arena = dr if side else dl
            cm = [LM[i] if LM[i] is not None else RM[i] for i in range(k)]
            cm_h = [lm[i] if lm[i] is not None else rm[i] for i in range(k)]
            C_ = [arena.offset(st_) for st_ in starts]
            ctx.dev_memset(dom, 0xA5, q + 64)
            has = ctx.broadcast_super_array_scalar(fmt, 3, s, C_, lens, O, cm, OM, scalar_is_lhs=side)
            got, got_masks = do.download(np.uint8, p), dom.download(np.uint8, q)
            for i in sample:
                n = lens[i]
                assert has[i] == (cm_h[i] is not None)
                if n == 0:
                    continue
                full = np.full(n, s, dtype=dt)
                l_, r_ = (full, cols[i]) if side else (cols[i], full)
                g = got[starts[i]:starts[i] + n * dt.itemsize].view(dt)
                if cm_h[i] is None:
                    st, want, _, _ = oracle.apply_int(oracle.aligned_copy(l_), oracle.aligned_copy(r_), "divide")
                    np.testing.assert_array_equal(g, want)
                else:
                    st, want, want_mask = oracle.int_body("masked_std", l_, r_, "divide", mask=oracle.pad_bits(cm_h[i], n))
                    np.testing.assert_array_equal(g, want, err_msg=f"scalar side={side} chunk {i}")
                    nb = ((n + 63) // 64) * 8
                    np.testing.assert_array_equal(got_masks[m_starts[i]:m_starts[i] + nb], want_mask[:nb])
    finally:
        ctx.set_variant(0)
    # masked outputs that start mid-vector: the chunk-by-chunk path, same results
    few = list(range(0, 12))
    O2 = [do.offset(starts[i] + dt.itemsize) for i in few]
    ctx.dev_memset(dom, 0xA5, q + 64)
    has = ctx.route_super_array_broadcast(fmt, 3, [L[i] for i in few], [R[i] for i in few], [max(lens[i] - 1, 0) for i in few],
                                          [max(lens[i] - 1, 0) for i in few], O2, [LM[i] for i in few], [RM[i] for i in few], [OM[i] for i in few])
    got, got_masks = do.download(np.uint8, p), dom.download(np.uint8, q)
    for j, i in enumerate(few):
        n = max(lens[i] - 1, 0)
        c = common(i, lens[i])
        if n == 0:
            continue
        g = got[starts[i] + dt.itemsize:starts[i] + dt.itemsize + n * dt.itemsize].view(dt)
        if c is None:
            st, want, _, _ = oracle.apply_int(oracle.aligned_copy(lhs[i][:n]), oracle.aligned_copy(rhs[i][:n]), "divide")
            np.testing.assert_array_equal(g, want)
        else:
            st, want, want_mask = oracle.int_body("masked_std", lhs[i][:n], rhs[i][:n], "divide", mask=oracle.pad_bits(c, n))
            np.testing.assert_array_equal(g, want, err_msg=f"mid-vector chunk {i}")
            nb = ((n + 63) // 64) * 8
            np.testing.assert_array_equal(got_masks[m_starts[i]:m_starts[i] + nb], want_mask[:nb])
