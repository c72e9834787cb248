"""GPU parity tests of ma_sum_columns: per-column {sum, valid count} of many columns in two launches. Every column
must equal what the single-column entry point (and the oracle) gives: bit-exact for integers, <= 1 ULP of the exactly
rounded sum for floats, for ragged lengths, empty columns, masks at odd bit offsets, device- and host-resident data."""
import math

import numpy as np
import pytest

from minarrow_amd import ffi
from minarrow_amd.host import live_variants, tuning_build

pytestmark = pytest.mark.gpu

NP = {"i": np.int32, "I": np.uint32, "l": np.int64, "L": np.uint64, "f": np.float32, "g": np.float64,
      "c": np.int8, "C": np.uint8, "s": np.int16, "S": np.uint16}
TAG = {"i": "i32", "I": "u32", "l": "i64", "L": "u64", "f": "f32", "g": "f64", "c": "i8", "C": "u8", "s": "i16", "S": "u16"}


def make_columns(rng, fmt, lens):
    dt = NP[fmt]
    cols = []
    for n in lens:
        if fmt in "fg":
            cols.append((rng.standard_normal(n) * 1e3).astype(dt))
        else:
            info = np.iinfo(dt)
            lo, hi = (info.min, info.max) if dt().itemsize <= 2 else (info.min // 2, info.max // 2)
            cols.append(rng.integers(lo, hi, size=n, dtype=dt, endpoint=dt().itemsize <= 2))
    return cols


@pytest.mark.parametrize("fmt", list(NP))
@pytest.mark.parametrize("device", [True, False])
def test_ragged_columns_match_single_column_sums(ctx, fmt, device):
    rng = np.random.default_rng(ord(fmt))
    lens = [0, 1, 5, 63, 64, 65, 1000, 4095, 4096, 4097, 65_535, 65_536, 65_537, 200_003, 0, 17]
    if fmt in "cCsS":  # their segments are 524 288 / 262 144 rows (512 KiB)
        lens += [262_143, 262_145, 524_287, 524_289, 1_100_003]
    cols = make_columns(rng, fmt, lens)
    masks, offs = [], []
    for i, n in enumerate(lens):
        if i % 3 == 0:
            masks.append(None)
            offs.append(0)
        else:
            off = [0, 3, 64, 77][i % 4]
            masks.append(rng.integers(0, 256, size=(off + n) // 8 + 16, dtype=np.uint8))
            offs.append(off)
    if device:
        # sub-allocate at odd element offsets so that segment heads are exercised
        d_cols = [ctx.to_device(np.concatenate([np.zeros(1, c.dtype), c]), 64) for c in cols]
        ptrs = [d.ptr + c.itemsize for d, c in zip(d_cols, cols)]
        d_masks = [ctx.to_device(m, 16) if m is not None else None for m in masks]
    else:
        ptrs, d_masks = cols, masks
    f, i64, cnt = ctx.sum_columns(fmt, ptrs, lens, d_masks, offs)
    for k, (c, n) in enumerate(zip(cols, lens)):
        valid = (np.unpackbits(masks[k], bitorder="little")[offs[k]:offs[k] + n].astype(bool)
                 if masks[k] is not None else np.ones(n, dtype=bool))
        assert cnt[k] == valid.sum(), k
        sel = c[valid]
        if fmt in "fg":
            exact = math.fsum(sel.astype(np.float64).tolist())
            assert abs(f[k] - exact) <= math.ulp(exact), (k, n)
        else:
            want = int(sel.astype(object).sum()) if n else 0
            assert (int(i64[k]) - want) % (1 << 64) == 0, (k, n)
        # identical to the single-column entry point
        if n:
            s1, c1 = ctx.sum(TAG[fmt], ptrs[k], n, mask=d_masks[k], mask_bit_offset=offs[k])
            assert c1 == cnt[k]
            if fmt in "fg":
                assert abs(s1 - f[k]) <= math.ulp(f[k])
            else:
                assert (int(s1) - int(i64[k])) % (1 << 64) == 0


@pytest.mark.parametrize("fmt", ["l", "g", "i", "f", "C", "s"])
def test_many_short_columns_take_the_wave_per_column_form(ctx, fmt):
    """From 256 columns of a segment or less each (a chunked column handed over chunk by chunk) a WAVE sums a column, the
    32-byte descriptors are read where the host built them and partial c is column c: 3000 columns of ragged lengths incl.
    empty ones and whole segments, sub-allocated at odd element offsets, validity at odd bit offsets on two thirds."""
    rng = np.random.default_rng(ord(fmt) + 1)
    dt = NP[fmt]
    seg = 65_536 if np.dtype(dt).itemsize >= 4 else (1 << 19) // np.dtype(dt).itemsize
    lens = [int(x) for x in rng.choice([0, 1, 63, 64, 65, 1000, 4097, 8192, 8191, 20_001], size=3000)]
    lens[5], lens[17] = seg, seg - 1
    cols = make_columns(rng, fmt, lens)
    arena = np.concatenate([np.concatenate([np.zeros(1, c.dtype), c]) for c in cols])
    starts = np.cumsum([0] + [c.size + 1 for c in cols[:-1]]) + 1
    dev = ctx.to_device(arena, 64)
    ptrs = [dev.ptr + int(s) * arena.itemsize for s in starts]
    masks, offs, d_masks = [], [], []
    for i, n in enumerate(lens):
        if i % 3 == 0 or n == 0:
            masks.append(None); offs.append(0); d_masks.append(None)
        else:
            off = [0, 3, 64, 77][i % 4]
            m = rng.integers(0, 256, size=(off + n) // 8 + 16, dtype=np.uint8)
            masks.append(m); offs.append(off); d_masks.append(ctx.to_device(m, 16))
    f, i64, cnt = ctx.sum_columns(fmt, ptrs, lens, d_masks, offs)
    for k, (c, n) in enumerate(zip(cols, lens)):
        valid = (np.unpackbits(masks[k], bitorder="little")[offs[k]:offs[k] + n].astype(bool)
                 if masks[k] is not None else np.ones(n, dtype=bool))
        assert cnt[k] == valid.sum(), k
        sel = c[valid]
        if fmt in "fg":
            exact = math.fsum(sel.astype(np.float64).tolist())
            assert abs(f[k] - exact) <= math.ulp(exact), (k, n)
        else:
            want = int(sel.astype(object).sum()) if n else 0
            assert (int(i64[k]) - want) % (1 << 64) == 0, (k, n)
    dense_f, dense_i, dense_c = ctx.sum_columns(fmt, ptrs, lens)
    np.testing.assert_array_equal(dense_c, np.array(lens, dtype=np.uint64))


@pytest.mark.parametrize("fmt", ["i", "I"])
def test_whole_segments_of_extreme_32_bit_values_cannot_overflow_the_half_word_accumulators(ctx, fmt):
    """The wave-per-column form adds 32-bit integers in halves (two 32-bit words per accumulator, widened once per column): 300
    columns of 65 536 rows — the longest a column of this form gets — of nothing but the type's minimum, maximum, -1 / 2^16 - 1
    patterns, dense and with validity, per column and as ONE total (where the words are widened per chunk)."""
    dt = NP[fmt]
    info = np.iinfo(dt)
    fills = [info.min, info.max, dt(-1) if fmt == "i" else dt(0xFFFF), dt(0x7FFF8000 if fmt == "i" else 0xFFFF0000), dt(0x8000)]
    rng = np.random.default_rng(77)
    n, k = 65_536, 300
    lens = [n] * k
    lens[7], lens[100] = n - 1, n - 2049
    cols = [np.full(m, fills[i % len(fills)], dtype=dt) for i, m in enumerate(lens)]
    arena = np.concatenate(cols)
    starts = np.cumsum([0] + [c.size for c in cols[:-1]])
    dev = ctx.to_device(arena, 64)
    ptrs = [dev.ptr + int(s) * arena.itemsize for s in starts]
    bits = rng.integers(0, 256, size=n // 8 + 32, dtype=np.uint8)
    bits[:64] = 0xFF
    d_bits = ctx.to_device(bits, 16)
    valid = np.unpackbits(bits, bitorder="little")
    for masks, offs in ((None, None), ([d_bits] * k, [5] * k)):
        f, i64, cnt = ctx.sum_columns(fmt, ptrs, lens, masks, offs)
        total = 0
        for j, (c, m) in enumerate(zip(cols, lens)):
            v = valid[5:5 + m].astype(bool) if masks else np.ones(m, dtype=bool)
            want = int(c[0]) * int(v.sum())
            total += want
            assert cnt[j] == v.sum(), j
            assert (int(i64[j]) - want) % (1 << 64) == 0, (j, int(i64[j]), want)
            assert f[j] == float(want), j
        tf, ti, tc = ctx.sum_chunks(fmt, ptrs, lens, masks, offs)
        assert (int(ti) - total) % (1 << 64) == 0, (ti, total)
        assert tc == (sum(int(valid[5:5 + m].sum()) for m in lens) if masks else sum(lens))


@pytest.mark.parametrize("total", [False, True])
def test_back_to_back_async_calls_on_long_chunk_lists_each_read_their_own_table(ctx, total):
    """From 8192 short columns the descriptor table goes to a device buffer on the context's upload stream (four buffers used in turn,
    pieces copied while the host still writes the rest): twelve asynchronous calls in a row, each over a DIFFERENT selection of
    20 000-40 000 chunks of one arena with validity on every other call, results left in device memory — every call's results must be
    those of its own table, whatever the host and the copies were doing meanwhile."""
    import ctypes as C
    rng = np.random.default_rng(99)
    per, k_all = 1024, 40_000
    arena = rng.integers(-(1 << 30), 1 << 30, size=per * k_all, dtype=np.int32)
    bits = rng.integers(0, 256, size=per * k_all // 8 + 64, dtype=np.uint8)
    dev = ctx.to_device(arena, 64)
    d_bits = ctx.to_device(bits, 64)
    col_sums = arena.reshape(k_all, per).astype(np.int64).sum(axis=1)
    valid = np.unpackbits(bits, bitorder="little")[:per * k_all].reshape(k_all, per).astype(bool)
    masked_sums = np.where(valid, arena.reshape(k_all, per).astype(np.int64), 0).sum(axis=1)
    masked_cnts = valid.sum(axis=1)
    calls, keep = [], []
    n_calls = 12
    outs = ctx.alloc(n_calls * 3 * 8 * k_all)
    ctx.set_async(True)
    try:
        for c in range(n_calls):
            k = 20_000 + 1_667 * c
            sel = rng.permutation(k_all)[:k]
            masked = c % 2 == 1
            ptrs = (C.c_void_p * k)(*[dev.ptr + int(j) * per * 4 for j in sel])
            lens = (C.c_size_t * k)(*([per] * k))
            mks = (C.c_void_p * k)(*[d_bits.ptr + int(j) * (per // 8) for j in sel]) if masked else None
            keep.append((ptrs, lens, mks))
            base = outs.ptr + c * 3 * 8 * k_all
            fn = ctx.lib.ma_sum_chunks if total else ctx.lib.ma_sum_columns
            ffi.check(fn(ctx.handle, ord("i"), k, C.cast(ptrs, C.c_void_p), C.cast(lens, C.c_void_p),
                         C.cast(mks, C.c_void_p) if mks is not None else None, None, base, base + 8 * k_all, base + 16 * k_all))
            calls.append((sel, masked, k))
        ctx.synchronize()
    finally:
        ctx.set_async(False)
    for c, (sel, masked, k) in enumerate(calls):
        n_out = 1 if total else k
        got_i = outs.download(np.int64, n_out, c * 3 * 8 * k_all + 8 * k_all)
        got_c = outs.download(np.uint64, n_out, c * 3 * 8 * k_all + 16 * k_all)
        want_i = (masked_sums if masked else col_sums)[sel]
        want_c = masked_cnts[sel] if masked else np.full(k, per)
        if total:
            assert int(got_i[0]) == int(want_i.sum()) and int(got_c[0]) == int(want_c.sum()), c
        else:
            np.testing.assert_array_equal(got_i, want_i, err_msg=f"call {c}")
            np.testing.assert_array_equal(got_c, want_c.astype(np.uint64), err_msg=f"call {c}")


@pytest.mark.parametrize("k", [8191, 8192, 8193, 16384 + 8191, 16384 + 8192, 50_001])
def test_chunk_lists_either_side_of_the_upload_threshold_and_of_a_piece(ctx, k):
    """Below 8192 columns the descriptor table is read where it was built, from there on it is uploaded in pieces of 16 384
    descriptors (a last piece of less than half a piece rides with the one before): the sizes either side of both edges, ragged
    lengths incl. empty columns, validity on every third column, twice in a row on the same context."""
    rng = np.random.default_rng(k)
    lens = rng.choice([0, 1, 63, 64, 100, 257, 1024], size=k).astype(np.int64)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    arena = rng.integers(-(1 << 31), 1 << 31, size=int(lens.sum()) + 8, dtype=np.int32)
    bits = rng.integers(0, 256, size=arena.size // 8 + 64, dtype=np.uint8)
    dev, d_bits = ctx.to_device(arena, 64), ctx.to_device(bits, 64)
    valid = np.unpackbits(bits, bitorder="little")[:arena.size].astype(bool)
    ptrs = [dev.ptr + int(s) * 4 for s in starts]
    masks = [d_bits if i % 3 == 0 else None for i in range(k)]
    offs = [int(s) if i % 3 == 0 else 0 for i, s in enumerate(starts)]
    csum = np.concatenate([[0], np.cumsum(arena.astype(np.int64))])
    gated = np.concatenate([[0], np.cumsum(np.where(valid, arena.astype(np.int64), 0))])
    vcnt = np.concatenate([[0], np.cumsum(valid)])
    e = starts + lens
    third = np.arange(k) % 3 == 0
    want_i = np.where(third, gated[e] - gated[starts], csum[e] - csum[starts])
    want_c = np.where(third, vcnt[e] - vcnt[starts], lens)
    for _ in range(2):
        f, i64, cnt = ctx.sum_columns("i", ptrs, [int(n) for n in lens], masks, offs)
        np.testing.assert_array_equal(i64, want_i)
        np.testing.assert_array_equal(cnt, want_c.astype(np.uint64))
        np.testing.assert_array_equal(f, want_i.astype(np.float64))


def test_a_long_chunk_list_in_pageable_host_memory_is_staged_while_its_table_is_being_uploaded(ctx):
    """9000 chunks that live in ordinary host memory: every chunk is staged through the call's scope while the descriptor table's
    upload has begun — the two must not step on each other's staging."""
    rng = np.random.default_rng(5)
    cols = [rng.integers(-1000, 1000, size=int(n), dtype=np.int64) for n in rng.integers(1, 200, size=9000)]
    f, i64, cnt = ctx.sum_columns("l", cols, [c.size for c in cols])
    np.testing.assert_array_equal(i64, np.array([int(c.sum()) for c in cols], dtype=np.int64))
    np.testing.assert_array_equal(cnt, np.array([c.size for c in cols], dtype=np.uint64))
    tf, ti, tc = ctx.sum_chunks("l", cols, [c.size for c in cols])
    assert ti == sum(int(c.sum()) for c in cols) and tc == sum(c.size for c in cols)


def test_thousand_small_columns(ctx):
    """1000 columns of 1000 rows (the launch-bound shape): iota data, closed forms."""
    k, n = 1000, 1000
    data = ctx.alloc(k * n * 8)
    ctx.synth_iota("i64", data, k * n, 0)
    f, i64, cnt = ctx.sum_columns("l", [data.ptr + i * n * 8 for i in range(k)], [n] * k)
    base = np.arange(k, dtype=np.int64) * n
    np.testing.assert_array_equal(i64, base * n + n * (n - 1) // 2)
    np.testing.assert_array_equal(cnt, np.full(k, n, dtype=np.uint64))
    np.testing.assert_array_equal(f, i64.astype(np.float64))


def test_config5_per_column_reduce_in_one_call(ctx):
    """8 chunks x 2^24 rows with 10 % nulls (config 5 scaled): per-chunk sums in one call == one call per chunk, and
    their fold == the sum of the physically consolidated column."""
    k, n = 8, 1 << 24
    chunks = [ctx.alloc(n * 8) for _ in range(k)]
    masks = [ctx.alloc(n // 8 + 64) for _ in range(k)]
    for c in range(k):
        ctx.synth_iota("f64", chunks[c], n, c)
        ctx.synth_validity(masks[c], n, seed=0x55 + c, null_every=10)
    f, _, cnt = ctx.sum_columns("g", chunks, [n] * k, masks, [0] * k)
    for c in range(k):
        s, v = ctx.sum("f64", chunks[c], n, mask=masks[c])
        assert v == cnt[c] and abs(s - f[c]) <= math.ulp(s)
    out, om = ctx.alloc(k * n * 8), ctx.alloc(k * n // 8 + 64)
    assert ctx.consolidate_column(8, chunks, [n] * k, out, masks, [0] * k, om)
    s, v = ctx.sum("f64", out, k * n, mask=om)
    assert v == int(cnt.sum())
    assert abs(s - math.fsum(f.tolist())) <= 2 * math.ulp(s)


def test_errors(ctx):
    a = np.arange(10, dtype=np.int64)
    with pytest.raises(ffi.MinarrowHipError) as e:
        ctx.sum_columns("e", [a], [10])  # float16: not a numeric array type of the reference
    assert e.value.status == ffi.MA_ERR_UNSUPPORTED
    f, i64, cnt = ctx.sum_columns("l", [], [])
    assert len(f) == 0


@pytest.mark.parametrize("fmt", ["l", "L", "g", "f", "i", "C", "s"])
@pytest.mark.parametrize("shape", ["few_long", "many_short", "mixed", "empty"])
def test_sum_chunks_is_the_sum_of_the_consolidated_column(ctx, fmt, shape):
    """ma_sum_chunks: ONE {sum, count} over a column held as a chunk list == the single-column sum of the concatenated
    chunks — bit-exact for integers, within 1 ULP of the exactly rounded sum for floats whatever the number of chunks
    (7 000 chunks: the two-level fold of the partials), validity at odd bit offsets on two thirds of the chunks."""
    rng = np.random.default_rng(ord(fmt) * 7 + len(shape))
    dt = NP[fmt]
    if shape == "few_long":
        lens = [200_003, 65_536, 1, 0, 70_001]
    elif shape == "many_short":
        lens = [int(x) for x in rng.choice([0, 1, 63, 64, 65, 1000, 8192, 8191], size=7000)]
    elif shape == "mixed":
        lens = [int(x) for x in rng.choice([5, 8192, 300_000], size=300, p=[0.3, 0.6, 0.1])]
    else:
        lens = []
    cols = make_columns(rng, fmt, lens)
    masks, offs, d_cols, d_masks = [], [], [], []
    for i, n in enumerate(lens):
        d_cols.append(ctx.to_device(cols[i], 64))
        if i % 3 == 0 or n == 0:
            masks.append(None); offs.append(0); d_masks.append(None)
        else:
            off = [0, 3, 64, 77][i % 4]
            m = rng.integers(0, 256, size=(off + n) // 8 + 16, dtype=np.uint8)
            masks.append(m); offs.append(off); d_masks.append(ctx.to_device(m, 16))
    f, i64, cnt = ctx.sum_chunks(fmt, d_cols, lens, d_masks if lens else None, offs if lens else None)
    sel = [c[np.unpackbits(m, bitorder="little")[o:o + n].astype(bool)] if m is not None else c
           for c, m, o, n in zip(cols, masks, offs, lens)]
    allv = np.concatenate(sel) if sel else np.zeros(0, dtype=dt)
    assert cnt == allv.size
    if fmt in "fg":
        exact = math.fsum(allv.astype(np.float64).tolist())
        assert abs(f - exact) <= math.ulp(exact) if allv.size else f == 0.0
        assert i64 is None
    else:
        want = int(allv.astype(object).sum()) if allv.size else 0
        assert (int(i64) - want) % (1 << 64) == 0
    f2, i2, c2 = ctx.sum_chunks(fmt, d_cols, lens)  # dense
    assert c2 == sum(lens)
    if fmt in "fg" and lens:
        exact = math.fsum(np.concatenate(cols).astype(np.float64).tolist())
        assert abs(f2 - exact) <= math.ulp(exact)


@pytest.mark.parametrize("fmt", ["l", "g", "i", "f", "s", "C"])
def test_short_columns_with_few_waves_walk_many_columns_each(ctx, fmt):
    """The wave-per-column kernel keeps the next tile requested while it works on this one, ACROSS column boundaries: with
    the grid cut to 3 or 1 workgroups every wave walks hundreds of columns (full tiles, no full tile, empty, misaligned,
    dense next to masked), at both tile depths, per column and as one total — and the first shape of the round (variant
    bit 4096) must give the same integers / floats within 1 ULP of the exact sum."""
    rng = np.random.default_rng(ord(fmt) + 99)
    lens = [int(x) for x in rng.choice([0, 1, 63, 64, 65, 1000, 2048, 4096, 4097, 8192, 8191, 12_345], size=1500)]
    cols = make_columns(rng, fmt, lens)
    arena = np.concatenate([np.concatenate([np.zeros(1, c.dtype), c]) for c in cols])
    starts = np.cumsum([0] + [c.size + 1 for c in cols[:-1]]) + 1
    dev = ctx.to_device(arena, 64)
    ptrs = [dev.ptr + int(s) * arena.itemsize for s in starts]
    masks, offs, d_masks = [], [], []
    for i, n in enumerate(lens):
        if i % 3 == 0 or n == 0:
            masks.append(None); offs.append(0); d_masks.append(None)
        else:
            off = [0, 3, 64, 77][i % 4]
            m = rng.integers(0, 256, size=(off + n) // 8 + 16, dtype=np.uint8)
            masks.append(m); offs.append(off); d_masks.append(ctx.to_device(m, 16))
    valid = [np.unpackbits(m, bitorder="little")[o:o + n].astype(bool) if m is not None else np.ones(n, dtype=bool)
             for m, o, n in zip(masks, offs, lens)]
    want_cnt = np.array([v.sum() for v in valid], dtype=np.uint64)
    sel = [c[v] for c, v in zip(cols, valid)]
    try:
        shapes = ((3, 0), (1, 4), (3, 6), (2, 32768), (0, 4096), (2, 4096))  # 32768: every wave from tile 0 (variants: tuning forms)
        for grid, variant in [(g, v) for g, v in shapes if v in live_variants([v])] + ([] if tuning_build() else [(1, 0), (2, 0), (0, 0)]):
            ctx.set_grid(grid)
            ctx.set_variant(variant)
            f, i64, cnt = ctx.sum_columns(fmt, ptrs, lens, d_masks, offs)
            np.testing.assert_array_equal(cnt, want_cnt)
            tf, ti, tc = ctx.sum_chunks(fmt, ptrs, lens, d_masks, offs)
            assert tc == int(want_cnt.sum())
            if fmt in "fg":
                for k, s in enumerate(sel):
                    exact = math.fsum(s.astype(np.float64).tolist())
                    assert abs(f[k] - exact) <= math.ulp(exact), (grid, variant, k)
                exact = math.fsum(np.concatenate(sel).astype(np.float64).tolist())
                assert abs(tf - exact) <= math.ulp(exact), (grid, variant)
            else:
                want = [int(s.astype(object).sum()) if s.size else 0 for s in sel]
                assert all((int(a) - w) % (1 << 64) == 0 for a, w in zip(i64, want)), (grid, variant)
                assert (int(ti) - sum(want)) % (1 << 64) == 0, (grid, variant)
    finally:
        ctx.set_grid(0)
        ctx.set_variant(0)


@pytest.mark.parametrize("fmt", ["s", "S", "C", "c"])
def test_long_columns_of_a_large_table_are_cut_into_pieces_on_the_device(ctx, fmt):
    """From 16 384 pieces of 64 KiB a table of many LONG 1- or 2-byte columns is described per column, cut into pieces by a kernel and summed
    by the wave kernel: 405 ragged columns of 17 000 pieces in all — empty ones, one of a single row, starts off the 16-byte
    grid, validity at odd bit offsets on a third — per column and as one total, against numpy, and against the segment
    form (variant bit 4096)."""
    rng = np.random.default_rng(ord(fmt) + 5)
    dt = NP[fmt]
    isz = np.dtype(dt).itemsize
    piece = 65_536 // isz
    total_rows = 17_000 * piece
    lens = [int(x) for x in rng.integers(1, 2 * total_rows // 400, size=400)]
    lens[3], lens[20], lens[35] = 0, 1, piece  # empty, one row, exactly one piece
    lens += [0, 7 * piece + 13, 0, 5, piece - 1]
    info = np.iinfo(dt)
    arena = rng.integers(info.min, int(info.max) + 1, size=sum(lens) + len(lens), dtype=dt)
    starts = np.cumsum([0] + [n + 1 for n in lens[:-1]]) + 1  # one element between columns: every phase of the 16-byte grid
    dev = ctx.to_device(arena, 64)
    ptrs = [dev.ptr + int(s) * isz for s in starts]
    masks, offs, d_masks = [], [], []
    for i, n in enumerate(lens):
        if i % 3 != 1 or n == 0:
            masks.append(None); offs.append(0); d_masks.append(None)
        else:
            off = [3, 64, 77][i % 3]
            m = rng.integers(0, 256, size=(off + n) // 8 + 16, dtype=np.uint8)
            masks.append(m); offs.append(off); d_masks.append(ctx.to_device(m, 16))
    want_sum, want_cnt = [], []
    for s0, n, m, o in zip(starts, lens, masks, offs):
        c = arena[int(s0):int(s0) + n].astype(np.int64)
        v = np.unpackbits(m, bitorder="little")[o:o + n].astype(bool) if m is not None else np.ones(n, dtype=bool)
        want_sum.append(int(c[v].sum()))
        want_cnt.append(int(v.sum()))
    try:
        for variant in live_variants((0, 4096)):  # 4096: round 3's first shape (a tuning form)
            ctx.set_variant(variant)
            f, i64, cnt = ctx.sum_columns(fmt, ptrs, lens, d_masks, offs)
            np.testing.assert_array_equal(cnt, np.array(want_cnt, dtype=np.uint64))
            tf, ti, tc = ctx.sum_chunks(fmt, ptrs, lens, d_masks, offs)
            assert tc == sum(want_cnt)
            assert [int(x) for x in i64] == want_sum, variant
            assert int(ti) == sum(want_sum)
    finally:
        ctx.set_variant(0)


def test_segment_form_finds_its_column_among_thousands(ctx):
    """One column longer than a segment sends the whole table through the workgroup-per-segment form, whose column search
    probes 64 entries at a time: 9000 columns (three rounds of probes), empty ones in runs (they share their first segment
    with the next column), two long ones at the ends and one in the middle."""
    rng = np.random.default_rng(77)
    lens = [int(x) for x in rng.choice([0, 0, 1, 5, 64, 300, 1000], size=9000)]
    lens[0], lens[4500], lens[-1] = 70_000, 200_001, 65_537
    lens[10:30] = [0] * 20
    lens[-5:-1] = [0] * 4
    cols = [rng.integers(-(1 << 40), 1 << 40, size=n, dtype=np.int64) for n in lens]
    arena = np.concatenate(cols)
    starts = np.cumsum([0] + lens[:-1])
    dev = ctx.to_device(arena, 64)
    ptrs = [dev.ptr + int(s) * 8 for s in starts]
    f, i64, cnt = ctx.sum_columns("l", ptrs, lens)
    np.testing.assert_array_equal(cnt, np.array(lens, dtype=np.uint64))
    np.testing.assert_array_equal(i64, np.array([int(c.sum()) for c in cols], dtype=np.int64))
    tf, ti, tc = ctx.sum_chunks("l", ptrs, lens)
    assert tc == sum(lens) and int(ti) == int(arena.sum())


@pytest.mark.parametrize("fmt", ["l", "L", "g"])
@pytest.mark.parametrize("n_cols", [1, 4, 5, 9])
def test_few_long_columns_take_the_fused_scan(ctx, fmt, n_cols):
    """Up to 16 device-resident 8-byte columns of >= 2^21 rows each: ma_sum_columns / ma_sum_chunks run the fused scan of
    ma_reduce_fused.hip, four columns per launch (dense or with validity: a tile requested ahead), and fold its per-column
    partials. Results must be what the general path (ctx variant 16384) gives: integers bit-exact, floats within 1 ULP of
    the exactly rounded sum; columns start on odd 8-byte offsets (a one-row head), validity at odd bit offsets on two thirds
    of them, ragged lengths; then the same list dense."""
    rng = np.random.default_rng(ord(fmt) * 31 + n_cols)
    lens = [(1 << 21) + [5, 4097, 0, 70_001, 123, 8191, 1, 300_000, 64][i] for i in range(n_cols)]
    cols = make_columns(rng, fmt, lens)
    d_cols = [ctx.to_device(np.concatenate([np.zeros(1, c.dtype), c]), 64) for c in cols]
    ptrs = [d.ptr + 8 for d in d_cols]
    masks, offs, d_masks = [], [], []
    for i, n in enumerate(lens):
        if i % 3 == 0:
            masks.append(None); offs.append(0); d_masks.append(None)
        else:
            off = [0, 3, 64, 77][i % 4]
            m = rng.integers(0, 256, size=(off + n) // 8 + 16, dtype=np.uint8)
            masks.append(m); offs.append(off); d_masks.append(ctx.to_device(m, 16))
    valid = [np.unpackbits(m, bitorder="little")[o:o + n].astype(bool) if m is not None else np.ones(n, dtype=bool)
             for m, o, n in zip(masks, offs, lens)]

    def check(f, i64, cnt):
        for k in range(n_cols):
            assert cnt[k] == valid[k].sum(), k
            sel = cols[k][valid[k]]
            if fmt == "g":
                exact = math.fsum(sel.tolist())
                assert abs(f[k] - exact) <= math.ulp(exact), k
            else:
                assert (int(i64[k]) - int(sel.astype(object).sum())) % (1 << 64) == 0, k

    ctx.set_variant(65536)  # several launches' worth of columns take the fused scan from ~384 MiB per launch only: forced here
    try:
        got = ctx.sum_columns(fmt, ptrs, lens, d_masks, offs)
    finally:
        ctx.set_variant(0)
    check(*got)
    ctx.set_variant(16384)  # the general path: a workgroup per 65 536-row segment
    try:
        ref = ctx.sum_columns(fmt, ptrs, lens, d_masks, offs)
        ref_total = ctx.sum_chunks(fmt, ptrs, lens, d_masks, offs)
    finally:
        ctx.set_variant(0)
    check(*ref)
    np.testing.assert_array_equal(got[2], ref[2])
    if fmt != "g":
        np.testing.assert_array_equal(got[1], ref[1])
        np.testing.assert_array_equal(got[0], ref[0])  # the f64 view of an integer sum
    # the same list as ONE column
    ctx.set_variant(65536)
    try:
        f, i64, cnt = ctx.sum_chunks(fmt, ptrs, lens, d_masks, offs)
        f2, i2, c2 = ctx.sum_columns(fmt, ptrs, lens)  # dense, no mask table at all: per column and as ONE column
        t2 = ctx.sum_chunks(fmt, ptrs, lens)
    finally:
        ctx.set_variant(0)
    assert cnt == sum(int(v.sum()) for v in valid) == ref_total[2]
    if fmt == "g":
        exact = math.fsum(np.concatenate([c[v] for c, v in zip(cols, valid)]).tolist())
        assert abs(f - exact) <= math.ulp(exact) and abs(ref_total[0] - exact) <= math.ulp(exact)
    else:
        assert i64 == ref_total[1] and f == ref_total[0]
    # the dense results against the general path too
    ctx.set_variant(16384)
    try:
        g2 = ctx.sum_columns(fmt, ptrs, lens)
        gt = ctx.sum_chunks(fmt, ptrs, lens)
    finally:
        ctx.set_variant(0)
    np.testing.assert_array_equal(c2, np.array(lens, dtype=np.uint64))
    np.testing.assert_array_equal(c2, g2[2])
    assert t2[2] == gt[2] == sum(lens)
    for k in range(n_cols):
        if fmt == "g":
            exact = math.fsum(cols[k].tolist())
            assert abs(f2[k] - exact) <= math.ulp(exact) and abs(g2[0][k] - exact) <= math.ulp(exact)
        else:
            assert (int(i2[k]) - int(cols[k].astype(object).sum())) % (1 << 64) == 0
            assert i2[k] == g2[1][k] and f2[k] == g2[0][k]
    if fmt == "g":
        exact = math.fsum(np.concatenate(cols).tolist())
        assert abs(t2[0] - exact) <= math.ulp(exact) and abs(gt[0] - exact) <= math.ulp(exact)
    else:
        assert t2[1] == gt[1] and t2[0] == gt[0]
        assert (int(t2[1]) - sum(int(c.astype(object).sum()) for c in cols)) % (1 << 64) == 0
