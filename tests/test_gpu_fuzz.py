"""Property-based GPU parity (hypothesis): shapes, element offsets (16-byte phases), bit offsets and null densities
drawn at random around the boundaries the kernels care about (64 rows = one validity word, one wave tile, one
workgroup tile, the segment size of ma_sum_columns). Every example is checked bit-exactly against numpy / the oracle.
`MA_FUZZ_EXAMPLES` scales the campaign (default 40 examples per property; a 20 000-example-per-property campaign —
120 000 cases — passed on an MI355X in round 1)."""
import math
import os

import numpy as np
import pytest

from minarrow_amd.host import live_variants
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

pytestmark = pytest.mark.gpu

N_EXAMPLES = int(os.environ.get("MA_FUZZ_EXAMPLES", "40"))
# MA_FUZZ_RANDOM=1: fresh random examples on every run (campaign mode) instead of the reproducible default set
COMMON = dict(max_examples=N_EXAMPLES, deadline=None, derandomize=os.environ.get("MA_FUZZ_RANDOM", "0") != "1", database=None,
              suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow, HealthCheck.data_too_large])

NP = {"i8": np.int8, "u8": np.uint8, "i16": np.int16, "u16": np.uint16, "i32": np.int32, "u32": np.uint32,
      "i64": np.int64, "u64": np.uint64, "f32": np.float32, "f64": np.float64}
# lengths clustered around the kernel boundaries
EDGES = [0, 1, 2, 15, 16, 17, 63, 64, 65, 127, 128, 129, 255, 256, 257, 1023, 1024, 1025, 4095, 4096, 4097, 8191, 8192,
         16383, 16384, 16385, 32767, 32768, 65535, 65536, 65537]
lengths = st.one_of(st.sampled_from(EDGES), st.integers(0, 70_000), st.integers(100_000, 300_000))


def unpack(bits, off, n):
    return np.unpackbits(bits, bitorder="little")[off:off + n].astype(bool)


def rand_values(rng, dt, n, small=False):
    dt = np.dtype(dt)
    if dt.kind == "f":
        return (rng.standard_normal(n) * 100).astype(dt)
    info = np.iinfo(dt)
    if small:
        return rng.integers(max(info.min, -50), min(info.max, 50), size=n, endpoint=True).astype(dt)
    return rng.integers(info.min, info.max, size=n, dtype=dt, endpoint=True)


@settings(**COMMON)
@given(tag=st.sampled_from(["i32", "u32", "i64", "u64", "f32", "f64", "i8", "u8", "i16", "u16"]), n=lengths, elem_off=st.integers(0, 7),
       mask_off=st.integers(0, 130), masked=st.booleans(), null_pct=st.sampled_from([0, 1, 10, 50, 100]),
       seed=st.integers(0, 2**31))
def test_sum(ctx, tag, n, elem_off, mask_off, masked, null_pct, seed):
    rng = np.random.default_rng(seed)
    dt = np.dtype(NP[tag])
    vals = rand_values(rng, dt, n + elem_off)
    dev = ctx.to_device(vals, 64)
    x = vals[elem_off:]
    bits = None
    valid = np.ones(n, dtype=bool)
    dmask = None
    if masked:
        nb = (mask_off + n) // 8 + 16
        bits = (rng.random(nb * 8) >= null_pct / 100.0)
        bits = np.packbits(bits, bitorder="little")
        valid = unpack(bits, mask_off, n)
        dmask = ctx.to_device(bits, 16)
    s, c = ctx.sum(tag, dev.ptr + elem_off * dt.itemsize, n, mask=dmask, mask_bit_offset=mask_off)
    assert c == int(valid.sum())
    sel = x[valid]
    if dt.kind == "f":
        exact = math.fsum(sel.astype(np.float64).tolist())
        assert abs(s - exact) <= math.ulp(exact)
    else:
        want = int(sel.astype(object).sum()) if sel.size else 0
        assert (int(s) - want) % (1 << 64) == 0


@settings(**COMMON)
@given(tag=st.sampled_from(list(NP)), op=st.sampled_from([0, 1, 2, 3, 4, 6]), kind=st.sampled_from(["aa", "as", "sa"]),
       n=lengths, offs=st.tuples(st.integers(0, 5), st.integers(0, 5), st.integers(0, 5)), masked=st.booleans(),
       mask_off=st.integers(0, 70), seed=st.integers(0, 2**31))
def test_apply(ctx, oracle, tag, op, kind, n, offs, masked, mask_off, seed):
    rng = np.random.default_rng(seed)
    dt = np.dtype(NP[tag])
    is_float = dt.kind == "f"
    la, lb, lo = offs
    a = rand_values(rng, dt, n + la, small=True)
    b = rand_values(rng, dt, n + lb, small=True)
    if not is_float and not masked and op in (3, 4, 6):
        b[b == 0] = 1  # a dense integer zero divisor is an error status, covered elsewhere
    if dt.kind == "i" and op in (3, 4, 6):
        b[b == -1] = 1  # MIN / -1: lane vs scalar-tail behaviour differs inside the reference itself
    sc = dt.type(3)
    da, db = ctx.to_device(a, 64), ctx.to_device(b, 64)
    do = ctx.alloc((n + lo) * dt.itemsize + 64)
    pa, pb, po = da.ptr + la * dt.itemsize, db.ptr + lb * dt.itemsize, do.ptr + lo * dt.itemsize
    x, y = np.ascontiguousarray(a[la:]), np.ascontiguousarray(b[lb:])
    bits = dm = dom = None
    if masked:
        # the oracle gates from bit 0: give it the window, the GPU the un-windowed bitmap + offset
        full = np.packbits(rng.random(((mask_off + n) // 8 + 16) * 8) >= 0.3, bitorder="little")
        win = np.packbits(unpack(full, mask_off, n), bitorder="little") if n else np.zeros(0, np.uint8)
        bits = np.concatenate([win, np.zeros(24, np.uint8)])
        dm, dom = ctx.to_device(full, 16), ctx.alloc(n // 8 + 64)
    if kind == "aa":
        lhs, rhs = x, y
        ctx.apply(tag, pa, pb, op, po, n, n, mask=dm, mask_bit_offset=mask_off, out_mask=dom)
    elif kind == "as":
        lhs, rhs = x, np.full(n, sc, dtype=dt)
        ctx.apply_scalar(tag, "rhs", pa, n, sc, op, po, mask=dm, mask_bit_offset=mask_off, out_mask=dom)
    else:
        lhs, rhs = np.full(n, sc, dtype=dt), y
        ctx.apply_scalar(tag, "lhs", pb, n, sc, op, po, mask=dm, mask_bit_offset=mask_off, out_mask=dom)
    if n == 0:
        return
    ref = oracle.apply_float if is_float else oracle.apply_int
    status, want, want_mask, _ = ref(np.ascontiguousarray(lhs), np.ascontiguousarray(rhs), op, mask=bits)
    assert status == 0
    got = do.download(dt, n, lo * dt.itemsize)
    if is_float:
        same = (got.view(np.uint8).reshape(n, -1) == want[:n].view(np.uint8).reshape(n, -1)).all(axis=1)
        both_nan = np.isnan(got) & np.isnan(want[:n])
        assert (same | both_nan).all()
    else:
        np.testing.assert_array_equal(got, want[:n])
    if masked:
        nb = ((n + 63) // 64) * 8
        np.testing.assert_array_equal(dom.download(np.uint8, nb), want_mask[:nb])


@settings(**COMMON)
@given(op=st.sampled_from(["and_masks", "or_masks", "xor_masks", "not_mask", "bitmask_slice"]), n=lengths,
       lo=st.integers(0, 200), ro=st.integers(0, 200), out_off=st.sampled_from([0, 8, 16]), seed=st.integers(0, 2**31))
def test_bitmask_word_ops(ctx, op, n, lo, ro, out_off, seed):
    if n == 0:
        return
    rng = np.random.default_rng(seed)
    nb = (max(lo, ro) + n) // 8 + 24
    a = rng.integers(0, 256, size=nb, dtype=np.uint8)
    b = rng.integers(0, 256, size=nb, dtype=np.uint8)
    da, db = ctx.to_device(a, 16), ctx.to_device(b, 16)
    out = ctx.alloc(n // 8 + 64)
    po = out.ptr + out_off
    if op in ("not_mask", "bitmask_slice"):
        ctx.mask_unary_op(op, da, lo, n, po)
        # not_mask addresses its window at byte granularity (bitmask/mod.rs:124-128), slice at bit granularity
        start = (lo // 8) * 8 if op == "not_mask" else lo
        src = unpack(a, start, n)
        want = ~src if op == "not_mask" else src
    else:
        ctx.mask_words_op(op, da, lo, db, ro, n, po)
        x, y = unpack(a, (lo // 8) * 8, n), unpack(b, (ro // 8) * 8, n)
        want = {"and_masks": x & y, "or_masks": x | y, "xor_masks": x ^ y}[op]
    nbytes = ((n + 63) // 64) * 8
    got = out.download(np.uint8, nbytes, out_off)
    np.testing.assert_array_equal(unpack(got, 0, n), want)
    assert not unpack(got, 0, nbytes * 8)[n:].any()  # trailing bits cleared


@settings(**COMMON)
@given(tag=st.sampled_from(["u8", "u16", "u32", "u64"]), n=lengths, elem_off=st.integers(0, 17), seed=st.integers(0, 2**31))
def test_simd_eq_mask(ctx, tag, n, elem_off, seed):
    if n == 0:
        return
    rng = np.random.default_rng(seed)
    dt = np.dtype(NP[tag])
    data = rng.integers(0, 16, size=n + elem_off).astype(dt)
    dev = ctx.to_device(data, 64)
    out = ctx.alloc(n // 8 + 64)
    ctx.simd_eq_mask(tag, dev.ptr + elem_off * dt.itemsize, n, 0x6, 0x4, out)
    nbytes = ((n + 63) // 64) * 8
    got = out.download(np.uint8, nbytes)
    np.testing.assert_array_equal(unpack(got, 0, n), (data[elem_off:] & 0x6) == 0x4)
    assert not unpack(got, 0, nbytes * 8)[n:].any()


@settings(**COMMON)
@given(elem=st.sampled_from([1, 2, 4, 8]), lens=st.lists(st.one_of(st.sampled_from(EDGES[:24]), st.integers(0, 40_000)), min_size=1, max_size=9),
       starts=st.lists(st.integers(0, 9), min_size=9, max_size=9), with_masks=st.booleans(), seed=st.integers(0, 2**31),
       variant=st.sampled_from([0, 0, 256, 128]))
def test_consolidate(ctx, elem, lens, starts, with_masks, seed, variant):
    """variant 256: the chunk-per-workgroup kernel on pinned-host descriptors (4- and 8-byte columns), 128: the tile form."""
    rng = np.random.default_rng(seed)
    dt = {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[elem]
    chunks = [rand_values(rng, dt, n + s) for n, s in zip(lens, starts)]
    devs = [ctx.to_device(c, 64) for c in chunks]
    ptrs = [d.ptr + s * elem for d, s in zip(devs, starts)]
    total = sum(lens)
    masks = offs = dmasks = None
    if with_masks:
        masks, offs, dmasks = [], [], []
        for i, n in enumerate(lens):
            if i % 3 == 2:
                masks.append(None)
                offs.append(0)
                dmasks.append(None)
            else:
                off = int(rng.integers(0, 100))
                m = rng.integers(0, 256, size=(off + n) // 8 + 16, dtype=np.uint8)
                masks.append(m)
                offs.append(off)
                dmasks.append(ctx.to_device(m, 16))
    out = ctx.alloc(max(total, 1) * elem + 64)
    om = ctx.alloc(total // 8 + 64)
    ctx.set_variant(variant)
    try:
        has = ctx.consolidate_column(elem, ptrs, lens, out, dmasks, offs, om)
    finally:
        ctx.set_variant(0)
    want = np.concatenate([c[s:] for c, s in zip(chunks, starts)]) if total else np.zeros(0, dt)
    np.testing.assert_array_equal(out.download(dt, total), want)
    any_mask = with_masks and any(m is not None for m in masks)
    assert has == any_mask
    if any_mask and total:
        want_valid = np.concatenate([unpack(m, o, n) if m is not None else np.ones(n, bool) for m, o, n in zip(masks, offs, lens)])
        nbytes = ((total + 63) // 64) * 8
        got = om.download(np.uint8, nbytes)
        np.testing.assert_array_equal(unpack(got, 0, total), want_valid)
        assert not unpack(got, 0, nbytes * 8)[total:].any()


@settings(**COMMON)
@given(fmt=st.sampled_from(["i", "I", "l", "L", "f", "g", "c", "C", "s", "S"]),
       lens=st.lists(st.one_of(st.sampled_from(EDGES), st.integers(0, 150_000)), min_size=1, max_size=12),
       starts=st.lists(st.integers(0, 5), min_size=12, max_size=12), with_masks=st.booleans(), seed=st.integers(0, 2**31))
def test_sum_columns(ctx, fmt, lens, starts, with_masks, seed):
    rng = np.random.default_rng(seed)
    dt = np.dtype({"i": np.int32, "I": np.uint32, "l": np.int64, "L": np.uint64, "f": np.float32, "g": np.float64,
                   "c": np.int8, "C": np.uint8, "s": np.int16, "S": np.uint16}[fmt])
    cols = [rand_values(rng, dt, n + s) for n, s in zip(lens, starts)]
    devs = [ctx.to_device(c, 64) for c in cols]
    ptrs = [d.ptr + s * dt.itemsize for d, s in zip(devs, starts)]
    masks = offs = dmasks = None
    if with_masks:
        masks, offs, dmasks = [], [], []
        for i, n in enumerate(lens):
            if i % 2:
                masks.append(None); offs.append(0); dmasks.append(None)
            else:
                off = int(rng.integers(0, 100))
                m = rng.integers(0, 256, size=(off + n) // 8 + 16, dtype=np.uint8)
                masks.append(m); offs.append(off); dmasks.append(ctx.to_device(m, 16))
    f, i64, cnt = ctx.sum_columns(fmt, ptrs, lens, dmasks, offs)
    for k, (c, s, n) in enumerate(zip(cols, starts, lens)):
        x = c[s:]
        valid = unpack(masks[k], offs[k], n) if with_masks and masks[k] is not None else np.ones(n, bool)
        assert cnt[k] == valid.sum()
        sel = x[valid]
        if dt.kind == "f":
            exact = math.fsum(sel.astype(np.float64).tolist())
            assert abs(f[k] - exact) <= math.ulp(exact)
        else:
            want = int(sel.astype(object).sum()) if sel.size else 0
            assert (int(i64[k]) - want) % (1 << 64) == 0


SHORT = [0, 0, 1, 2, 15, 16, 17, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 2047, 2048, 2049, 4095,
         4096, 4097, 8191, 8192]


@settings(**COMMON)
@given(fmt=st.sampled_from(["i", "I", "l", "L", "f", "g", "c", "C", "s", "S"]), n_cols=st.integers(256, 700),
       gap=st.integers(0, 3), mask_mode=st.sampled_from(["none", "some", "all"]), grid=st.sampled_from([0, 0, 1, 2, 7]),
       variant=st.sampled_from(live_variants([0, 0, 4, 6, 32768])), seed=st.integers(0, 2**31))
def test_sum_short_columns(ctx, fmt, n_cols, gap, mask_mode, grid, variant, seed):
    """From 256 columns of a segment or less each: a WAVE per column with the next tile requested ahead, across columns
    (`column_waves_kernel`) — per column (ma_sum_columns) and as one total (ma_sum_chunks), on grids so small that a wave
    walks hundreds of columns, at both tile depths, every 16-byte phase, validity at any bit offset on none / some / all."""
    rng = np.random.default_rng(seed)
    dt = np.dtype({"i": np.int32, "I": np.uint32, "l": np.int64, "L": np.uint64, "f": np.float32, "g": np.float64,
                   "c": np.int8, "C": np.uint8, "s": np.int16, "S": np.uint16}[fmt])
    lens = [int(x) for x in rng.choice(SHORT, size=n_cols)]
    starts, pos = [], int(rng.integers(0, 4))
    for n in lens:
        starts.append(pos)
        pos += n + gap
    arena = rand_values(rng, dt, pos + 16)
    dev = ctx.to_device(arena, 64)
    ptrs = [dev.ptr + s * dt.itemsize for s in starts]
    mbytes = rng.integers(0, 256, size=sum((n + 7) // 8 + 24 for n in lens) + 64, dtype=np.uint8)
    dmask = ctx.to_device(mbytes, 16)
    mptrs, offs, valid, mpos = [], [], [], 0
    for i, n in enumerate(lens):
        masked = n > 0 and (mask_mode == "all" or (mask_mode == "some" and i % 3 != 0))
        off = int(rng.integers(0, 130)) if masked else 0
        mptrs.append(dmask.ptr + mpos if masked else None)
        offs.append(off)
        valid.append(np.unpackbits(mbytes[mpos:mpos + (off + n + 7) // 8 + 1], bitorder="little")[off:off + n].astype(bool)
                     if masked else np.ones(n, bool))
        mpos += (n + 7) // 8 + 24
    sel = [arena[s:s + n][v] for s, n, v in zip(starts, lens, valid)]
    try:
        ctx.set_grid(grid)
        ctx.set_variant(variant)
        f, i64, cnt = ctx.sum_columns(fmt, ptrs, lens, mptrs if mask_mode != "none" else None, offs if mask_mode != "none" else None)
        tf, ti, tc = ctx.sum_chunks(fmt, ptrs, lens, mptrs if mask_mode != "none" else None, offs if mask_mode != "none" else None)
    finally:
        ctx.set_grid(0)
        ctx.set_variant(0)
    np.testing.assert_array_equal(cnt, np.array([v.sum() for v in valid], dtype=np.uint64))
    assert tc == sum(int(v.sum()) for v in valid)
    if dt.kind == "f":
        for k, x in enumerate(sel):
            exact = math.fsum(x.astype(np.float64).tolist())
            assert abs(f[k] - exact) <= math.ulp(exact), k
        exact = math.fsum(np.concatenate(sel).astype(np.float64).tolist())
        assert abs(tf - exact) <= math.ulp(exact)
    else:
        want = [int(x.astype(object).sum()) if x.size else 0 for x in sel]
        assert all((int(a) - w) % (1 << 64) == 0 for a, w in zip(i64, want))
        assert (int(ti) - sum(want)) % (1 << 64) == 0


@settings(**COMMON)
@given(tag=st.sampled_from(["f32", "f64"]), n=lengths, offs=st.tuples(st.integers(0, 3), st.integers(0, 3), st.integers(0, 3), st.integers(0, 3)),
       masked=st.booleans(), seed=st.integers(0, 2**31))
def test_fma(ctx, oracle, tag, n, offs, masked, seed):
    if n == 0:
        return
    rng = np.random.default_rng(seed)
    dt = np.dtype(NP[tag])
    la, lb, lc, lo = offs
    a, b, c = (rand_values(rng, dt, n + k) for k in (la, lb, lc))
    da, db, dc = ctx.to_device(a, 64), ctx.to_device(b, 64), ctx.to_device(c, 64)
    do = ctx.alloc((n + lo) * dt.itemsize + 64)
    bits = dm = dom = None
    if masked:
        bits = np.concatenate([np.packbits(rng.random(((n + 63) // 64) * 64) >= 0.25, bitorder="little"), np.zeros(24, np.uint8)])
        dm, dom = ctx.to_device(bits, 16), ctx.alloc(n // 8 + 64)
    ctx.apply_fma(tag, da.ptr + la * dt.itemsize, db.ptr + lb * dt.itemsize, dc.ptr + lc * dt.itemsize,
                  do.ptr + lo * dt.itemsize, n, n, n, mask=dm, out_mask=dom)
    # 64-byte aligned copies: the reference's dispatch takes its SIMD body (fused mul_add, simd.rs:620) only for
    # aligned inputs; its scalar fallback is UNFUSED (dispatch.rs:266,280) — SURVEY a15. The GPU is always fused.
    status, want, want_mask = oracle.apply_fma(oracle.aligned_copy(a[la:]), oracle.aligned_copy(b[lb:]),
                                               oracle.aligned_copy(c[lc:]), mask=bits)
    assert status == 0
    got = do.download(dt, n, lo * dt.itemsize)
    np.testing.assert_array_equal(got.view(np.uint8), want[:n].view(np.uint8))
    if masked:
        nb = ((n + 63) // 64) * 8
        np.testing.assert_array_equal(dom.download(np.uint8, nb), want_mask[:nb])


@settings(**COMMON)
@given(pair=st.sampled_from([("i32", "f64"), ("f64", "i32"), ("i32", "f32"), ("f32", "i32")]), op=st.sampled_from([0, 1, 2, 3]),
       n=lengths, offs=st.tuples(st.integers(0, 5), st.integers(0, 5), st.integers(0, 5)), masked=st.booleans(),
       seed=st.integers(0, 2**31))
def test_promote(ctx, pair, op, n, offs, masked, seed):
    if n == 0:
        return
    rng = np.random.default_rng(seed)
    lt, rt = np.dtype(NP[pair[0]]), np.dtype(NP[pair[1]])
    ot = lt if lt.kind == "f" else rt
    la, lb, lo = offs
    a, b = rand_values(rng, lt, n + la, small=True), rand_values(rng, rt, n + lb, small=True)
    da, db = ctx.to_device(a, 64), ctx.to_device(b, 64)
    do = ctx.alloc((n + lo) * ot.itemsize + 64)
    valid = np.ones(n, bool)
    dm = dom = None
    if masked:
        bits = np.packbits(rng.random(((n + 63) // 64) * 64 + 64) >= 0.25, bitorder="little")
        valid = unpack(bits, 0, n)
        dm, dom = ctx.to_device(bits, 16), ctx.alloc(n // 8 + 64)
    ctx.apply_promote(pair[0], pair[1], da.ptr + la * lt.itemsize, db.ptr + lb * rt.itemsize, op, do.ptr + lo * ot.itemsize, n, n,
                      mask=dm, out_mask=dom)
    x, y = a[la:].astype(ot), b[lb:].astype(ot)  # `x as f64` — routing/arithmetic.rs:244-269
    with np.errstate(all="ignore"):
        want = [x + y, x - y, x * y, x / y][op].astype(ot)
    want = np.where(valid, want, ot.type(0))
    got = do.download(ot, n, lo * ot.itemsize)
    same = got.view(np.uint8).reshape(n, -1) == want.view(np.uint8).reshape(n, -1)
    assert (same.all(axis=1) | (np.isnan(got) & np.isnan(want))).all()
    if masked:
        np.testing.assert_array_equal(unpack(dom.download(np.uint8, ((n + 63) // 64) * 8), 0, n), valid)


@settings(**COMMON)
@given(lens=st.lists(st.one_of(st.sampled_from(EDGES[:22]), st.integers(0, 100_000)), min_size=1, max_size=9),
       with_masks=st.booleans(), seed=st.integers(0, 2**31))
def test_consolidate_boolean(ctx, lens, with_masks, seed):
    rng = np.random.default_rng(seed)
    if sum(lens) == 0:
        return
    chunks, dchunks, masks, dmasks = [], [], [], []
    for i, n in enumerate(lens):
        off = int(rng.integers(0, 130))
        src = rng.integers(0, 256, size=(off + n) // 8 + 24, dtype=np.uint8)
        chunks.append((src, off, n))
        dchunks.append((ctx.to_device(src, 16), off, n))
        if with_masks and i % 2 == 0:
            moff = int(rng.integers(0, 130))
            m = rng.integers(0, 256, size=(moff + n) // 8 + 24, dtype=np.uint8)
            masks.append((m, moff))
            dmasks.append((ctx.to_device(m, 16), moff))
        else:
            masks.append(None)
            dmasks.append(None)
    total = sum(lens)
    nbytes = ((total + 63) // 64) * 8
    out, om = ctx.alloc(nbytes + 8), ctx.alloc(nbytes + 8)
    has = ctx.consolidate_boolean_column(dchunks, out, dmasks, om)
    want = np.concatenate([unpack(s, o, n) for s, o, n in chunks])
    got = out.download(np.uint8, nbytes)
    np.testing.assert_array_equal(unpack(got, 0, total), want)
    assert not unpack(got, 0, nbytes * 8)[total:].any()
    any_mask = any(m is not None for m in masks)
    assert has == any_mask
    if any_mask:
        want_valid = np.concatenate([unpack(m[0], m[1], n) if m is not None else np.ones(n, bool) for m, (_, _, n) in zip(masks, chunks)])
        np.testing.assert_array_equal(unpack(om.download(np.uint8, nbytes), 0, total), want_valid)


@settings(**COMMON)
@given(fmt=st.sampled_from(["i", "l", "f", "g"]), op=st.sampled_from([0, 1, 2]),
       lens=st.lists(st.one_of(st.sampled_from(EDGES[:24]), st.integers(0, 50_000)), min_size=1, max_size=10),
       mask_mode=st.sampled_from(["none", "mixed", "override"]), seed=st.integers(0, 2**31),
       variant=st.sampled_from(live_variants([0, 32, 64, 96, 256, 256 | 32, 256 | 16])), out_off=st.integers(0, 3))
def test_route_super_array_broadcast(ctx, fmt, op, lens, mask_mode, seed, variant, out_off):
    """variant: 32 = the 8 x 16-byte tile also for short chunks, 64 = output bitmaps by the second launch instead of the
    computing wave, 256 = the chunk-per-workgroup kernel (RechunkStrategy-sized chunks) with either tile width; out_off: outputs start out_off elements past a 16-byte boundary (a masked chunk off the boundary
    sends the whole call to the second-launch path)."""
    rng = np.random.default_rng(seed)
    dt = np.dtype({"i": np.int32, "l": np.int64, "f": np.float32, "g": np.float64}[fmt])
    k = len(lens)
    L = [rand_values(rng, dt, n, small=True) for n in lens]
    R = [rand_values(rng, dt, n, small=True) for n in lens]
    dL, dR = [ctx.to_device(x, 64) for x in L], [ctx.to_device(x, 64) for x in R]
    bO = [ctx.alloc(n * dt.itemsize + 128) for n in lens]
    dO = [b.ptr + (out_off if i % 2 else 0) * dt.itemsize for i, b in enumerate(bO)]
    dOM = [ctx.alloc(n // 8 + 64) for n in lens]
    lm = rm = [None] * k
    override = None
    if mask_mode == "mixed":
        lm = [np.packbits(rng.random(((n + 63) // 64) * 64 + 64) >= 0.3, bitorder="little") if i % 3 != 0 else None for i, n in enumerate(lens)]
        rm = [np.packbits(rng.random(((n + 63) // 64) * 64 + 64) >= 0.3, bitorder="little") if i % 2 == 0 else None for i, n in enumerate(lens)]
    elif mask_mode == "override":
        nmax = max(lens)
        override = np.packbits(rng.random(((nmax + 63) // 64) * 64 + 64) >= 0.3, bitorder="little")
    dlm = [ctx.to_device(m, 16) if m is not None else None for m in lm]
    drm = [ctx.to_device(m, 16) if m is not None else None for m in rm]
    dov = ctx.to_device(override, 16) if override is not None else None
    ctx.set_variant(variant)
    try:
        has = ctx.route_super_array_broadcast(fmt, op, dL, dR, lens, lens, dO, dlm, drm, dOM, dov)
    finally:
        ctx.set_variant(0)
    for i, n in enumerate(lens):
        if override is not None:
            valid = unpack(override, 0, n)
            want_has = True
        elif lm[i] is not None or rm[i] is not None:
            a = unpack(lm[i], 0, n) if lm[i] is not None else np.zeros(n, bool)
            b = unpack(rm[i], 0, n) if rm[i] is not None else np.zeros(n, bool)
            valid = a | b  # Bitmask::union (super_array.rs:224)
            want_has = True
        else:
            valid = np.ones(n, bool)
            want_has = False
        assert has[i] == want_has
        if n == 0:
            continue
        with np.errstate(all="ignore"):
            res = [L[i] + R[i], L[i] - R[i], L[i] * R[i]][op].astype(dt)
        want = np.where(valid, res, dt.type(0))
        got = bO[i].download(dt, n, (out_off if i % 2 else 0) * dt.itemsize)
        np.testing.assert_array_equal(got.view(np.uint8), want.view(np.uint8))
        if want_has:
            bits = dOM[i].download(np.uint8, ((n + 63) // 64) * 8)
            np.testing.assert_array_equal(unpack(bits, 0, n), valid)
            assert not unpack(bits, n, ((n + 63) // 64) * 64 - n).any()  # trailing bits of the last word are zero


@settings(**COMMON)
@given(fmt=st.sampled_from(["i", "I", "l", "L", "f", "g"]), op=st.sampled_from([0, 1, 2]), side=st.booleans(),
       lens=st.lists(st.one_of(st.sampled_from(EDGES[:24]), st.integers(0, 50_000)), min_size=1, max_size=10),
       masked=st.booleans(), seed=st.integers(0, 2**31), variant=st.sampled_from(live_variants([0, 32, 64, 256, 256 | 32])),
       in_off=st.integers(0, 3), out_off=st.integers(0, 3))
def test_broadcast_super_array_scalar(ctx, fmt, op, side, lens, masked, seed, variant, in_off, out_off):
    """SuperArray (op) Scalar / Scalar (op) SuperArray (super_array.rs:87-116, scalar.rs:214-243) in one launch: chunks on
    any element phase, outputs off the 16-byte boundary on every other chunk, optional validity per chunk, both kernel forms."""
    rng = np.random.default_rng(seed)
    dt = np.dtype({"i": np.int32, "I": np.uint32, "l": np.int64, "L": np.uint64, "f": np.float32, "g": np.float64}[fmt])
    k = len(lens)
    C_ = [rand_values(rng, dt, n + in_off, small=True) for n in lens]
    s = rand_values(rng, dt, 1, small=True)[0]
    dC = [ctx.to_device(x, 64) for x in C_]
    pC = [d.ptr + in_off * dt.itemsize for d in dC]
    bO = [ctx.alloc(n * dt.itemsize + 128) for n in lens]
    offs = [(out_off if (i % 2 and not masked) else 0) for i in range(k)]  # masked chunks of the chunk form start on 16 bytes
    dO = [b.ptr + o * dt.itemsize for b, o in zip(bO, offs)]
    ms = dms = doms = None
    if masked:
        ms = [np.packbits(rng.random(((n + 63) // 64) * 64 + 64) >= 0.3, bitorder="little") if i % 3 != 0 else None for i, n in enumerate(lens)]
        dms = [ctx.to_device(m, 16) if m is not None else None for m in ms]
        doms = [ctx.alloc(n // 8 + 64) for n in lens]
    ctx.set_variant(variant)
    try:
        has = ctx.broadcast_super_array_scalar(fmt, op, s, pC, lens, dO, dms, doms, scalar_is_lhs=side)
    finally:
        ctx.set_variant(0)
    for i, n in enumerate(lens):
        want_has = bool(masked and ms[i] is not None)
        assert has[i] == want_has
        if n == 0:
            continue
        x = C_[i][in_off:]
        full = np.full(n, s, dtype=dt)
        l, r = (full, x) if side else (x, full)
        with np.errstate(all="ignore"):
            res = [l + r, l - r, l * r][op].astype(dt)
        valid = unpack(ms[i], 0, n) if want_has else np.ones(n, bool)
        want = np.where(valid, res, dt.type(0))
        got = bO[i].download(dt, n, offs[i] * dt.itemsize)
        np.testing.assert_array_equal(got.view(np.uint8), want.view(np.uint8))
        if want_has:
            bits = doms[i].download(np.uint8, ((n + 63) // 64) * 8)
            np.testing.assert_array_equal(unpack(bits, 0, n), valid)


# ---- host-resident operands through the tiled staging pipeline (ma_pipeline.hip) -------------------------------------
TILE = 256 << 10  # bytes per operand per tile while these properties run


@settings(**COMMON)
@given(tag=st.sampled_from(["i8", "u16", "i32", "i64", "f32", "f64"]), op=st.sampled_from([0, 1, 2, 3, 6]),
       kind=st.sampled_from(["aa", "as", "sa"]), tiles=st.floats(2.0, 5.5), ragged=st.integers(0, 70),
       where=st.tuples(st.booleans(), st.booleans(), st.booleans()), masked=st.booleans(), mask_off=st.integers(0, 70),
       seed=st.integers(0, 2**31))
def test_apply_host_resident(ctx, oracle, tag, op, kind, tiles, ragged, where, masked, mask_off, seed):
    """Any mix of host (numpy, pageable) and device operands, lengths straddling tile seams: the oracle's bits."""
    rng = np.random.default_rng(seed)
    dt = np.dtype(NP[tag])
    is_float = dt.kind == "f"
    n = int(TILE / dt.itemsize * tiles) + ragged
    a, b = rand_values(rng, dt, n, small=True), rand_values(rng, dt, n, small=True)
    if not is_float and not masked and op in (3, 6):
        b[b == 0] = 1
    if dt.kind == "i" and op in (3, 6):
        b[b == -1] = 1
    sc = dt.type(3)
    host_a, host_b, host_o = where
    if not (host_a or host_b or host_o):
        host_o = True
    pa = a if host_a else ctx.to_device(a, 64)
    pb = b if host_b else ctx.to_device(b, 64)
    out_host = np.zeros(n, dtype=dt)
    po = out_host if host_o else ctx.alloc(n * dt.itemsize + 64)
    bits = full = om = None
    if masked:
        full = np.packbits(rng.random(((mask_off + n) // 8 + 16) * 8) >= 0.3, bitorder="little")
        bits = np.concatenate([np.packbits(unpack(full, mask_off, n), bitorder="little"), np.zeros(24, np.uint8)])
        om = np.zeros(n // 8 + 64, dtype=np.uint8)
    ctx.set_staging_tile(TILE)
    try:
        if kind == "aa":
            lhs, rhs = a, b
            ctx.apply(tag, pa, pb, op, po, n, n, mask=full, mask_bit_offset=mask_off, out_mask=om)
        elif kind == "as":
            lhs, rhs = a, np.full(n, sc, dtype=dt)
            ctx.apply_scalar(tag, "rhs", pa, n, sc, op, po, mask=full, mask_bit_offset=mask_off, out_mask=om)
        else:
            lhs, rhs = np.full(n, sc, dtype=dt), b
            ctx.apply_scalar(tag, "lhs", pb, n, sc, op, po, mask=full, mask_bit_offset=mask_off, out_mask=om)
    finally:
        ctx.set_staging_tile(32 << 20)
    ref = oracle.apply_float if is_float else oracle.apply_int
    status, want, want_mask, _ = ref(np.ascontiguousarray(lhs), np.ascontiguousarray(rhs), op, mask=bits)
    assert status == 0
    got = out_host if host_o else po.download(dt, n)
    if is_float:
        same = (got.view(np.uint8).reshape(n, -1) == want[:n].view(np.uint8).reshape(n, -1)).all(axis=1)
        assert (same | (np.isnan(got) & np.isnan(want[:n]))).all()
    else:
        np.testing.assert_array_equal(got, want[:n])
    if masked:
        nb = ((n + 63) // 64) * 8
        np.testing.assert_array_equal(om[:nb], want_mask[:nb])


@settings(**COMMON)
@given(tag=st.sampled_from(["i32", "u64", "f32", "f64"]), tiles=st.floats(2.0, 6.0), ragged=st.integers(0, 70),
       masked=st.booleans(), mask_off=st.integers(0, 130), null_pct=st.sampled_from([0, 10, 100]), seed=st.integers(0, 2**31))
def test_sum_host_resident(ctx, tag, tiles, ragged, masked, mask_off, null_pct, seed):
    rng = np.random.default_rng(seed)
    dt = np.dtype(NP[tag])
    n = int(TILE / dt.itemsize * tiles) + ragged
    x = rand_values(rng, dt, n)
    bits, valid = None, np.ones(n, dtype=bool)
    if masked:
        bits = np.packbits(rng.random(((mask_off + n) // 8 + 16) * 8) >= null_pct / 100.0, bitorder="little")
        valid = unpack(bits, mask_off, n)
    ctx.set_staging_tile(TILE)
    try:
        s, c = ctx.sum(tag, x, n, mask=bits, mask_bit_offset=mask_off)
    finally:
        ctx.set_staging_tile(32 << 20)
    assert c == int(valid.sum())
    sel = x[valid]
    if dt.kind == "f":
        exact = math.fsum(sel.astype(np.float64).tolist())
        assert abs(s - exact) <= math.ulp(exact)
    else:
        assert (int(s) - (int(sel.astype(object).sum()) if sel.size else 0)) % (1 << 64) == 0


# ---- whole-table consolidation into one arena (ma_consolidate_table_arena) ------------------------------------------
@settings(**COMMON)
@given(n_batches=st.integers(1, 7), n_cols=st.integers(1, 6), seed=st.integers(0, 2**31), pageable=st.booleans())
def test_consolidate_table_arena(ctx, oracle, n_batches, n_cols, seed, pageable):
    from test_gpu_consolidate import _arena_call

    rng = np.random.default_rng(seed)
    batch_rows = [int(rng.choice([0, 1, 63, 64, 65, 127, 128, 129, 4096, 4097, int(rng.integers(0, 20_000))])) for _ in range(n_batches)]
    if sum(batch_rows) == 0:
        batch_rows[-1] = 1 + int(rng.integers(0, 300))
    cols = []
    for _ in range(n_cols):
        dt = NP[str(rng.choice(list(NP)))]
        chunks = [rand_values(rng, dt, r) for r in batch_rows]
        kind = int(rng.integers(0, 3))
        if kind == 0:
            cols.append((chunks, None, None))
            continue
        offs = [int(rng.integers(0, 70)) for _ in batch_rows]
        masks = [None if (kind == 2 and rng.random() < 0.5) else rng.integers(0, 256, size=(o + r + 7) // 8 + 16, dtype=np.uint8)
                 for r, o in zip(batch_rows, offs)]
        cols.append((chunks, masks, offs))
    arena, d_off, m_off, used = _arena_call(ctx, cols, batch_rows, pageable=pageable)
    want, wd, wm, wu = oracle.consolidate_table_arena(cols)
    assert (d_off, m_off, used) == (wd, wm, wu)
    np.testing.assert_array_equal(arena, want)
