"""GPU parity tests for ma_consolidate_column (BASELINE config 5: SuperTable consolidate + per-column reduce).
Reference: src/structs/chunked/super_table.rs:657-743, src/traits/consolidate.rs:80-207, src/structs/arena.rs:264-308.
Bit-exact against the reference's own test vectors and the CPU oracle."""
import json
import math
from pathlib import Path

import numpy as np
import pytest

from minarrow_amd.host import live_variants

from minarrow_amd import ffi

pytestmark = pytest.mark.gpu

KAT = json.loads((Path(__file__).resolve().parent / "golden" / "consolidate_kat.json").read_text())
NP = {"i32": np.int32, "f64": np.float64}


def nbytes(n):
    return ((n + 63) // 64) * 8


def run(ctx, chunks, masks=None, offs=None, device=True):
    dt = chunks[0].dtype
    total = sum(c.size for c in chunks)
    if device:
        d_chunks = [ctx.to_device(c, 64) for c in chunks]
        d_masks = [ctx.to_device(m, 16) if m is not None else None for m in masks] if masks is not None else None
        out = ctx.alloc(max(total, 1) * dt.itemsize + 64)
        om = ctx.alloc(nbytes(total) + 8)
        ctx.dev_memset(om, 0xA5, nbytes(total) + 8)  # a validity word no kernel writes would show
        has = ctx.consolidate_column(dt.itemsize, d_chunks, [c.size for c in chunks], out, d_masks, offs, om)
        return out.download(dt, total), (om.download(np.uint8, nbytes(total)) if has else None)
    out = np.zeros(total, dtype=dt)
    om = np.full(nbytes(total) + 8, 0xA5, dtype=np.uint8)
    has = ctx.consolidate_column(dt.itemsize, chunks, [c.size for c in chunks], out, masks, offs, om)
    return out, (om[:nbytes(total)] if has else None)


def test_ref_integer_and_float(ctx):
    for col in KAT["integer_and_float"]["columns"]:
        chunks = [np.array(c, dtype=NP[col["type"]]) for c in col["chunks"]]
        for device in (True, False):
            out, mask = run(ctx, chunks, device=device)
            assert mask is None
            np.testing.assert_array_equal(out, np.array(col["expect"], dtype=NP[col["type"]]))


def test_ref_nullable(ctx, oracle):
    c = KAT["nullable"]
    chunks = [np.array(x, dtype=np.int32) for x in c["chunks"]]
    masks = [np.ascontiguousarray(oracle.pack_bits(v)[:8]) for v in c["validity"]]
    out, mask = run(ctx, chunks, masks, [0, 0])
    valid = oracle.unpack_bits(mask, 5)
    assert [int(v) if ok else None for v, ok in zip(out, valid)] == c["expect_get"]


def test_empty_supertable_is_rejected(ctx):
    with pytest.raises(ffi.MinarrowHipError) as e:
        ctx.consolidate_column(8, [], [], None)
    assert e.value.status == ffi.MA_ERR_INVALID_ARGUMENT and "empty SuperTable" in e.value.message


@pytest.mark.parametrize("dt", [np.int64, np.float64, np.int32, np.float32, np.uint16, np.uint8])
def test_random_chunks(ctx, oracle, dt):
    rng = np.random.default_rng(np.dtype(dt).itemsize)
    for lens in ([5], [3, 0, 2], [64, 64, 64], [1, 63, 65, 1000, 4096, 7], [100_003, 1, 50_000], [10_000] * 100):
        chunks = [rng.integers(0, 200, size=n).astype(dt) for n in lens]
        # no masks
        out, mask = run(ctx, chunks)
        assert mask is None
        np.testing.assert_array_equal(out, np.concatenate(chunks))
        # some chunks masked, with window offsets
        masks, offs = [], []
        for i, n in enumerate(lens):
            if i % 3 == 1:
                masks.append(None)
                offs.append(0)
            else:
                off = [0, 5, 64, 77][i % 4]
                masks.append(rng.integers(0, 256, size=(off + n) // 8 + 16, dtype=np.uint8))
                offs.append(off)
        want, want_mask = oracle.consolidate_column(chunks, masks, offs)
        for device in (True, False):
            out, mask = run(ctx, chunks, masks, offs, device=device)
            np.testing.assert_array_equal(out, want)
            if want_mask is None:
                assert mask is None
            else:
                np.testing.assert_array_equal(mask, want_mask[:nbytes(sum(lens))])


@pytest.mark.parametrize("k,dt,variant", [p for p in [(700, np.int32, 0), (700, np.int32, 256), (700, np.float64, 256), (9000, np.int32, 0),
                                                     (9000, np.int64, 0), (9000, np.int64, 1024), (9000, np.int32, 128)]
                                          if live_variants([p[2]])])  # 1024 (one segment) is a tuning form
def test_many_small_chunks(ctx, oracle, k, dt, variant):
    """700 chunks (more than the validity kernel stages in LDS), ragged lengths incl. empty ones, masks at odd bit
    offsets on two thirds of them; values and validity vs the oracle; and the Boolean column twin.
    variant 256 = the chunk-per-workgroup kernel on pinned-host descriptors (what 9000 chunks of a 4- or 8-byte column take by
    themselves, in three segments: 4096 + 8192... chunks; 1024 = as one segment), 128 = the tile-search kernel."""
    rng = np.random.default_rng(700)
    lens = [int(x) for x in rng.choice([0, 1, 7, 63, 64, 65, 130, 500, 3000] if k < 5000 else [0, 1, 7, 63, 64, 65, 130, 500], size=k)]
    chunks = [rng.integers(0, 1 << 30, size=n).astype(dt) for n in lens]
    masks, offs = [], []
    for i, n in enumerate(lens):
        if i % 3 == 0:
            masks.append(None)
            offs.append(0)
        else:
            off = int(rng.integers(0, 100))
            masks.append(rng.integers(0, 256, size=(off + n) // 8 + 16, dtype=np.uint8))
            offs.append(off)
    want, want_mask = oracle.consolidate_column(chunks, masks, offs)
    ctx.set_variant(variant)
    try:
        out, mask = run(ctx, chunks, masks, offs)
        dense, no_mask = run(ctx, chunks, None, None)
    finally:
        ctx.set_variant(0)
    np.testing.assert_array_equal(out, want)
    np.testing.assert_array_equal(mask, want_mask[:nbytes(sum(lens))])
    np.testing.assert_array_equal(dense, want)
    assert no_mask is None
    if k > 5000 or variant:
        return
    bchunks = [(rng.integers(0, 256, size=(n + 200) // 8 + 16, dtype=np.uint8), int(rng.integers(0, 100)), n) for n in lens]
    bmasks = [(m, o) if m is not None else None for m, o in zip(masks, offs)]
    want, want_mask = oracle.consolidate_boolean_column(bchunks, bmasks)
    got, got_mask = run_bool(ctx, bchunks, bmasks)
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(got_mask, want_mask)


@pytest.mark.parametrize("dt", [np.int8, np.int16, np.int32, np.float64])
@pytest.mark.parametrize("shape", ["aligned", "aligned_tail", "few_joins", "tail_in_join", "tiny_runs"])
@pytest.mark.parametrize("variant", live_variants([256, 256 + 1024]))  # the chunk form forced; in segments | as one segment (a tuning form)
def test_chunk_form_validity_words(ctx, oracle, dt, shape, variant):
    """The chunk-per-workgroup consolidate writes a chunk's validity words from the chunk's own workgroup and leaves only
    the words a chunk STARTS inside to a join pass (launched only when there is such a word). Shapes: every chunk a
    multiple of 64 rows (RechunkStrategy::Auto's 8192 among them: no join pass at all); the same with a ragged last chunk
    (the partial last word belongs to the chunk it begins in); a few ragged chunks among aligned ones; a last word that is
    itself a join; and runs of chunks shorter than a word, empty ones included (several starts inside one word: one owner).
    Bitmaps at odd bit offsets on two thirds of the chunks; the output bitmap is poisoned beforehand."""
    rng = np.random.default_rng(len(shape) * 7 + variant)
    k = 9000 if shape in ("aligned_tail", "few_joins") else 5000  # 9000: three segments unless variant 1024
    lens = [int(x) for x in rng.choice([0, 64, 128, 640, 8192], size=k, p=[0.1, 0.3, 0.3, 0.25, 0.05])]
    if shape == "aligned_tail":
        lens[-1] = 8192 + 37
    elif shape == "few_joins":
        for i in rng.choice(k, size=40, replace=False):
            lens[int(i)] = int(rng.choice([1, 63, 65, 100, 8191]))
    elif shape == "tail_in_join":
        lens[-3:] = [64 + 5, 0, 11]
    elif shape == "tiny_runs":
        for i in range(100, 400):
            lens[i] = int(rng.choice([0, 0, 1, 2, 5, 17, 63]))
        lens[-1] = 3
    chunks = [rng.integers(0, 100, size=n).astype(dt) for n in lens]
    masks, offs = [], []
    for i, n in enumerate(lens):
        if i % 3 == 0:
            masks.append(None)
            offs.append(0)
        else:
            off = int(rng.integers(0, 100))
            masks.append(rng.integers(0, 256, size=(off + n) // 8 + 16, dtype=np.uint8))
            offs.append(off)
    want, want_mask = oracle.consolidate_column(chunks, masks, offs)
    ctx.set_variant(variant)
    try:
        out, mask = run(ctx, chunks, masks, offs)
    finally:
        ctx.set_variant(0)
    np.testing.assert_array_equal(out, want)
    np.testing.assert_array_equal(mask, want_mask[:nbytes(sum(lens))])


def test_config5_shape_consolidate_then_reduce(ctx):
    """8 chunks (one per GPU in config 5; here on one device, scaled to 8 x 2^24 rows), schema {i64 v = i + chunk,
    f64 v * 0.1} as in benches/consolidate.rs:37-58. The per-column reduce of the logically consolidated table
    (sum of per-chunk sums, no data movement) equals the reduce of the physically consolidated column."""
    k, n = 8, 1 << 24
    chunks_i = [ctx.alloc(n * 8) for _ in range(k)]
    masks = [ctx.alloc(n // 8 + 64) for _ in range(k)]
    for c in range(k):
        ctx.synth_iota("i64", chunks_i[c], n, c)
        ctx.synth_validity(masks[c], n, seed=0xABC + c, null_every=10)
    out = ctx.alloc(k * n * 8)
    out_mask = ctx.alloc(k * n // 8 + 64)
    assert ctx.consolidate_column(8, chunks_i, [n] * k, out, masks, [0] * k, out_mask)
    parts = [ctx.sum("i64", chunks_i[c], n, mask=masks[c]) for c in range(k)]
    whole = ctx.sum("i64", out, k * n, mask=out_mask)
    assert whole == (sum(p[0] for p in parts), sum(p[1] for p in parts))
    assert ctx.popcount_mask(out_mask, 0, k * n) == whole[1]
    # dense check of the values: sum over chunks of sum(i + c)
    assert ctx.sum("i64", out, k * n)[0] == sum(n * (n - 1) // 2 + c * n for c in range(k))
    first = out.download(np.int64, 4, (3 * n) * 8)
    np.testing.assert_array_equal(first, np.arange(3, 7))


@pytest.mark.big
def test_config5_full_size_eight_billion_rows_on_one_gpu(ctx):
    """BASELINE configs[4] at its stated size on ONE GPU: a SuperTable of 8 x 10^9-row batches with 10 % nulls, BOTH its
    columns — i64 v = i + batch and f64 v = (i + batch) as f64, benches/consolidate.rs:37-58's pattern (the bench scales the
    float by 0.1; integer-valued here so that every sum has an exact closed form) — consolidated and reduced, one column after
    the other through the same 64 GB in + 64 GB out (+ 1 GB of validity each way: fits the 288 GB). Size-independent
    properties per column: the consolidated column reduces to the sum of the per-batch reduces (checksum of checksums, count
    included), its valid count is the popcount of the joined bitmap, its dense sum is the closed form, windows across every
    join equal the source rows, and the per-batch reduce as ONE ma_sum_chunks call agrees. f64: every sum within 1 ULP of
    the exact value — which the i64 column, sharing the bitmap and the values, supplies for the gated sums."""
    k, n = 8, 1_000_000_000
    M64 = (1 << 64) - 1
    src = ctx.alloc(k * n * 8)
    mstride = ((n + 511) // 512) * 64
    masks_buf = ctx.alloc(k * mstride + 64)
    chunks = [src.ptr + c * n * 8 for c in range(k)]
    masks = [masks_buf.ptr + c * mstride for c in range(k)]
    out = ctx.alloc(k * n * 8)
    out_mask = ctx.alloc(k * n // 8 + 64)
    dense_exact = sum(n * (n - 1) // 2 + c * n for c in range(k))  # 4.0e18 < 2^63
    for c in range(k):
        ctx.synth_validity(masks[c], n, seed=0xABC + c, null_every=10)

    def within_one_ulp(got, exact_int):
        return abs(got - exact_int) <= math.ulp(float(exact_int))

    # ---- the i64 column ------------------------------------------------------------------------------------------------
    for c in range(k):
        ctx.synth_iota("i64", chunks[c], n, c)  # benches/consolidate.rs:37-58: v = i + batch
    assert ctx.consolidate_column(8, chunks, [n] * k, out, masks, [0] * k, out_mask)
    parts = [ctx.sum("i64", chunks[c], n, mask=masks[c]) for c in range(k)]
    whole = ctx.sum("i64", out, k * n, mask=out_mask)
    assert (whole[0] & M64, whole[1]) == (sum(p[0] for p in parts) & M64, sum(p[1] for p in parts))
    assert 0.09 < 1 - whole[1] / (k * n) < 0.11
    assert ctx.popcount_mask(out_mask, 0, k * n) == whole[1]
    assert ctx.sum("i64", out, k * n)[0] & M64 == dense_exact & M64
    for c in range(1, k):  # every join: the last rows of batch c - 1, the first of batch c
        got = out.download(np.int64, 8, (c * n - 4) * 8)
        np.testing.assert_array_equal(got, np.concatenate([np.arange(n - 4, n) + (c - 1), np.arange(0, 4) + c]))
    total = ctx.sum_chunks("l", chunks, [n] * k, masks, [0] * k)
    assert (total[1] & M64, total[2]) == (whole[0] & M64, whole[1])
    gated_exact, gated_count = whole[0], whole[1]  # < 2^63: no wrap, the exact gated sum of the shared values
    assert 0 < gated_exact < dense_exact
    # ---- the f64 column of the same table: same values, same bitmaps, the same two buffers -----------------------------------
    ctx.dev_memset(out, 0xA5, k * n * 8)
    ctx.dev_memset(out_mask, 0x5A, k * n // 8 + 64)
    for c in range(k):
        ctx.synth_iota("f64", chunks[c], n, c)
    assert ctx.consolidate_column(8, chunks, [n] * k, out, masks, [0] * k, out_mask)
    fparts = [ctx.sum_dd("f64", chunks[c], n, mask=masks[c]) for c in range(k)]
    for c, (hi, lo, cnt) in enumerate(fparts):  # each batch's gated sum against its i64 twin (exact)
        assert cnt == parts[c][1] and within_one_ulp(hi + lo, parts[c][0]), c
    fwhole = ctx.sum("f64", out, k * n, mask=out_mask)
    assert fwhole[1] == gated_count and within_one_ulp(fwhole[0], gated_exact)
    from minarrow_amd.parallel import fold_dd

    assert within_one_ulp(fold_dd([(p[0], p[1]) for p in fparts]), gated_exact)  # the logical consolidate: sum of per-batch sums
    assert ctx.popcount_mask(out_mask, 0, k * n) == gated_count
    dense = ctx.sum("f64", out, k * n)
    assert dense[1] == k * n and within_one_ulp(dense[0], dense_exact)
    for c in range(1, k):
        got = out.download(np.float64, 8, (c * n - 4) * 8)
        np.testing.assert_array_equal(got, np.concatenate([np.arange(n - 4, n) + (c - 1), np.arange(0, 4) + c]).astype(np.float64))
    ftotal = ctx.sum_chunks("g", chunks, [n] * k, masks, [0] * k)
    assert ftotal[2] == gated_count and within_one_ulp(ftotal[0], gated_exact)
    for b in (src, masks_buf, out, out_mask):
        b.free()
    ctx.lib.ma_dev_pool_trim(ctx.handle, 0)  # 130 GB back to the driver before the next test


# ---- bit-packed columns: ma_consolidate_boolean_column (BooleanArray data bits + validity) ----------------------

def run_bool(ctx, chunks, masks=None, device=True):
    total = sum(c[2] for c in chunks)
    if device:
        d_chunks = [(ctx.to_device(c[0], 16), c[1], c[2]) for c in chunks]
        d_masks = [(ctx.to_device(m[0], 16), m[1]) if m is not None else None for m in masks] if masks is not None else None
        out, om = ctx.alloc(nbytes(total) + 8), ctx.alloc(nbytes(total) + 8)
        ctx.dev_memset(out, 0xA5, nbytes(total) + 8)  # a word no kernel writes would show
        ctx.dev_memset(om, 0xA5, nbytes(total) + 8)
        has = ctx.consolidate_boolean_column(d_chunks, out, d_masks, om)
        return out.download(np.uint8, nbytes(total)), (om.download(np.uint8, nbytes(total)) if has else None)
    out, om = np.zeros(nbytes(total) + 8, dtype=np.uint8), np.zeros(nbytes(total) + 8, dtype=np.uint8)
    has = ctx.consolidate_boolean_column(chunks, out, masks, om)
    return out[:nbytes(total)], (om[:nbytes(total)] if has else None)


@pytest.mark.parametrize("shape", ["aligned", "aligned_tail", "few_joins", "tail_in_join", "tiny_runs", "ragged"])
@pytest.mark.parametrize("variant", [0, 128])
def test_boolean_many_short_chunks(ctx, oracle, shape, variant):
    """A Boolean column rechunked into thousands of short chunks: from 1024 chunks a wave writes the output words wholly
    inside its chunk and a join pass the words a chunk starts inside (variant 128 = the searching kernel). Source bits at
    odd bit offsets, validity on half of the chunks, poisoned outputs; the same shapes as the numeric chunk form's test."""
    rng = np.random.default_rng(len(shape) * 3 + variant)
    k = 3000
    lens = [int(x) for x in rng.choice([0, 64, 128, 640, 8192], size=k, p=[0.1, 0.3, 0.3, 0.25, 0.05])]
    if shape == "aligned_tail":
        lens[-1] = 8192 + 37
    elif shape == "few_joins":
        for i in rng.choice(k, size=40, replace=False):
            lens[int(i)] = int(rng.choice([1, 63, 65, 100, 8191]))
    elif shape == "tail_in_join":
        lens[-3:] = [64 + 5, 0, 11]
    elif shape == "tiny_runs":
        for i in range(100, 400):
            lens[i] = int(rng.choice([0, 0, 1, 2, 5, 17, 63]))
        lens[-1] = 3
    elif shape == "ragged":
        lens = [int(x) for x in rng.integers(0, 3000, size=k)]
    chunks, masks = [], []
    for i, n in enumerate(lens):
        off = int(rng.integers(0, 130))
        chunks.append((rng.integers(0, 256, size=(off + n) // 8 + 24, dtype=np.uint8), off, n))
        if i % 2 == 0:
            moff = int(rng.integers(0, 130))
            masks.append((rng.integers(0, 256, size=(moff + n) // 8 + 24, dtype=np.uint8), moff))
        else:
            masks.append(None)
    want, want_mask = oracle.consolidate_boolean_column(chunks, masks)
    ctx.set_variant(variant)
    try:
        got, got_mask = run_bool(ctx, chunks, masks)
    finally:
        ctx.set_variant(0)
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(got_mask, want_mask)


def _bits(oracle, bools):
    return oracle.pack_bits(np.array(bools, dtype=bool)) if len(bools) else np.zeros(8, dtype=np.uint8)


def test_ref_boolean_vectors(ctx, oracle):
    b = KAT["boolean"]
    for c in b["extend_from_slice"]["cases"]:
        src = np.zeros(8, dtype=np.uint8)
        src[: len(c["src_bytes"])] = c["src_bytes"]
        out, mask = run_bool(ctx, [(_bits(oracle, c["start"]), 0, len(c["start"])), (src, 0, c["len"])])
        assert mask is None and oracle.unpack_bits(out, len(c["expect"])).tolist() == c["expect"]
    c = b["concat"]
    for device in (True, False):
        out, mask = run_bool(ctx, [(_bits(oracle, x), 0, len(x)) for x in c["chunks"]], device=device)
        assert mask is None and oracle.unpack_bits(out, 5).tolist() == c["expect"]
    c = b["concat_with_nulls"]
    out, mask = run_bool(ctx, [(_bits(oracle, x), 0, len(x)) for x in c["chunks"]], [(_bits(oracle, v), 0) for v in c["validity"]])
    data, valid = oracle.unpack_bits(out, 5), oracle.unpack_bits(mask, 5)
    assert [bool(d) if ok else None for d, ok in zip(data, valid)] == c["expect_get"]
    c = b["append_mask_onto_maskless"]
    out, mask = run_bool(ctx, [(_bits(oracle, x), 0, len(x)) for x in c["chunks"]],
                         [(_bits(oracle, v), 0) if v is not None else None for v in c["validity"]])
    assert oracle.unpack_bits(out, 5).tolist() == c["expect_data"]
    assert oracle.unpack_bits(mask, 5).tolist() == c["expect_validity"]


@pytest.mark.parametrize("device", [True, False])
def test_boolean_ragged_joins_match_oracle(ctx, oracle, device):
    rng = np.random.default_rng(99)
    for trial in range(12):
        k = int(rng.integers(1, 9))
        chunks, masks = [], []
        for i in range(k):
            n_bits = int(rng.choice([0, 1, 63, 64, 65, rng.integers(1, 5000), rng.integers(1, 200_000)]))
            src = rng.integers(0, 256, size=((n_bits + 200) // 64 + 2) * 8, dtype=np.uint8)
            off = int(rng.choice([0, 8, 64, rng.integers(0, 130)]))
            chunks.append((src, off, n_bits))
            if rng.random() < 0.5:
                masks.append((rng.integers(0, 256, size=src.size, dtype=np.uint8), int(rng.integers(0, 130))))
            else:
                masks.append(None)
        if sum(c[2] for c in chunks) == 0:
            continue
        want, want_mask = oracle.consolidate_boolean_column(chunks, masks)
        out, mask = run_bool(ctx, chunks, masks, device=device)
        np.testing.assert_array_equal(out, want)
        assert (mask is None) == (want_mask is None)
        if mask is not None:
            np.testing.assert_array_equal(mask, want_mask)


def test_boolean_large_column(ctx, oracle):
    """8 chunks x 125 M bits joined at odd bit positions (a 10^9-row BooleanArray column), bit-exact vs the oracle."""
    n = 125_000_003
    src = ctx.alloc(nbytes(n + 128) + 8)
    ctx.synth_validity(src, n + 128, seed=5, first_index=0, null_every=3)
    chunks = [(src, 7 * i + 1, n - 11 * i) for i in range(8)]
    total = sum(c[2] for c in chunks)
    out = ctx.alloc(nbytes(total) + 8)
    assert ctx.consolidate_boolean_column(chunks, out) is False
    host_src = src.download(np.uint8, nbytes(n + 128) + 8)
    want, _ = oracle.consolidate_boolean_column([(host_src, c[1], c[2]) for c in chunks])
    got = out.download(np.uint8, nbytes(total))
    assert np.array_equal(got, want)


def test_boolean_errors(ctx):
    with pytest.raises(ffi.MinarrowHipError) as e:
        ctx.consolidate_boolean_column([], None)
    assert e.value.status == ffi.MA_ERR_INVALID_ARGUMENT and "empty SuperTable" in e.value.message
    bits = np.zeros(16, dtype=np.uint8)
    with pytest.raises(ffi.MinarrowHipError) as e:
        ctx.consolidate_boolean_column([(bits, 0, 10)], np.zeros(16, dtype=np.uint8), [(bits, 0)], None)
    assert e.value.status == ffi.MA_ERR_INVALID_ARGUMENT


# ---- whole-table consolidation into one arena (src/structs/arena.rs:1187-1340) ---------------------------------------

def _arena_call(ctx, cols, batch_rows, pageable=False):
    """cols: [(chunks, masks or None, mask offsets or None)] of numpy arrays -> (arena bytes, data offs, mask offs, used)."""
    from minarrow_amd.host import arena_layout

    n_rows = sum(batch_rows)
    elem = [c[0][0].dtype.itemsize for c in cols]
    nulls = [c[1] is not None and any(m is not None for m in c[1]) for c in cols]
    _, _, capacity, _ = arena_layout(elem, nulls, n_rows)
    cells = [[ctx.to_device(ch, 64) if ch.size else None for ch in c[0]] for c in cols]
    masks = [[(ctx.to_device(m, 16) if m is not None else None) for m in c[1]] if c[1] is not None else [None] * len(batch_rows)
             for c in cols]
    offs = [c[2] if c[2] is not None else [0] * len(batch_rows) for c in cols]
    if pageable:
        arena = np.full(max(capacity, 64) + 64, 0xEE, dtype=np.uint8)
        base = (-arena.ctypes.data) % 64
        view = arena[base:base + max(capacity, 64)]
        d_off, m_off, used = ctx.consolidate_table_arena(elem, batch_rows, cells, view, capacity, masks, offs)
        return view[:capacity].copy(), d_off, m_off, used
    arena = ctx.alloc(max(capacity, 64))
    ffi.check(ctx.lib.ma_dev_memset(ctx.handle, arena.ptr, 0, max(capacity, 64)))
    d_off, m_off, used = ctx.consolidate_table_arena(elem, batch_rows, cells, arena, capacity, masks, offs)
    return arena.download(np.uint8, capacity), d_off, m_off, used


def test_arena_layout_is_host_arithmetic():
    from minarrow_amd.host import arena_layout

    a = KAT["arena"]
    assert arena_layout([8, 8], [False, True], 5) == ([0, 64], [None, 128], 192, a["full_table"]["expect_used"])
    d, m, cap, used = arena_layout([8] * 10, [True] * 10, 100)
    assert d == a["many_small"]["expect_data_offsets"] and m == a["many_small"]["expect_mask_offsets"]
    assert cap == 10 * (832 + 64) and used == 8896 + 13


def test_arena_reference_vectors(ctx, oracle):
    a = KAT["arena"]["full_table"]
    ids, prices = np.array(a["ids"], dtype=np.int64), np.array(a["prices"], dtype=np.float64)
    cols = [([ids], None, None), ([prices], [oracle.pad_bits(oracle.pack_bits(a["price_validity"]), 5)], [0])]
    arena, d_off, m_off, used = _arena_call(ctx, cols, [5])
    want, wd, wm, wu = oracle.consolidate_table_arena(cols)
    assert (d_off, m_off, used) == (wd, wm, wu) == ([0, 64], [None, 128], 129)
    np.testing.assert_array_equal(arena, want)
    m = KAT["arena"]["many_small"]
    cols = [([np.arange(m["rows"], dtype=np.int64) + i * m["rows"]],
             [oracle.pad_bits(oracle.pack_bits(np.ones(m["rows"], bool)), m["rows"])], [0]) for i in range(m["columns"])]
    arena, d_off, m_off, used = _arena_call(ctx, cols, [m["rows"]])
    assert [int(arena[o:o + 8].view(np.int64)[0]) for o in d_off] == m["expect_first_values"]
    np.testing.assert_array_equal(arena, oracle.consolidate_table_arena(cols)[0])


@pytest.mark.parametrize("pageable", [False, True])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_arena_random_tables_vs_oracle(ctx, oracle, seed, pageable):
    rng = np.random.default_rng(100 + seed)
    n_batches = int(rng.integers(1, 9))
    batch_rows = [int(rng.choice([0, 1, 63, 64, 65, 1000, 40_000, rng.integers(0, 9000)])) for _ in range(n_batches)]
    if sum(batch_rows) == 0:
        batch_rows[0] = 77
    types = [np.int64, np.float64, np.int32, np.float32, np.uint8, np.int16, np.uint64, np.int8]
    cols = []
    for dt in rng.permutation(types)[: int(rng.integers(1, len(types) + 1))]:
        chunks = [(rng.integers(0, 120, size=r)).astype(dt) for r in batch_rows]
        kind = rng.integers(0, 3)  # 0: no masks at all, 1: every batch masked, 2: some
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


@pytest.mark.parametrize("pageable", [False, True])
@pytest.mark.parametrize("aligned", [True, False])
def test_arena_many_short_batches_take_the_chunk_form(ctx, oracle, aligned, pageable):
    """A SuperTable of 1500 short batches (RechunkStrategy::Auto sizes; here 64 ... 640 rows, or ragged ones incl. empty
    batches): every column goes through the chunk-per-workgroup form — in-place descriptors, validity words by the chunk's
    own workgroup, join pass when batches start inside a word. All eight element types, masks on all / some / no batches."""
    rng = np.random.default_rng(17 + aligned)
    n_batches = 1500
    batch_rows = [int(x) for x in (rng.choice([64, 128, 640], size=n_batches) if aligned else rng.choice([0, 1, 63, 65, 200, 777], size=n_batches))]
    types = [np.int64, np.float64, np.int32, np.float32, np.uint8, np.int16, np.uint64, np.int8]
    cols = []
    for k, dt in enumerate(types):
        chunks = [(rng.integers(0, 120, size=r)).astype(dt) for r in batch_rows]
        kind = k % 3  # 0: no masks at all, 1: every batch masked, 2: some
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


def test_arena_bench_shape_and_errors(ctx, oracle):
    """benches/consolidate.rs:34-35: 100 tables x 10 000 rows; the numeric half of its 20-column table."""
    n_batches, rows = 100, 10_000
    cols = []
    for col in range(10):
        dt = np.int64 if col % 2 == 0 else np.float64
        scale = 1 if col % 2 == 0 else 0.1
        cols.append(([((np.arange(rows) + b * rows + col) * scale).astype(dt) for b in range(n_batches)], None, None))
    arena, d_off, m_off, used = _arena_call(ctx, cols, [rows] * n_batches)
    for c, off in enumerate(d_off):
        got = arena[off:off + n_batches * rows * 8].view(cols[c][0][0].dtype)
        np.testing.assert_array_equal(got, np.concatenate(cols[c][0]))
    assert m_off == [None] * 10
    with pytest.raises(ffi.MinarrowHipError) as e:  # Arena overflow (arena.rs:210-216)
        ctx.consolidate_table_arena([8], [4], [[ctx.to_device(np.arange(4, dtype=np.int64), 64)]], ctx.alloc(64), 16)
    assert e.value.status == ffi.MA_ERR_INVALID_ARGUMENT and "Arena overflow" in str(e.value)
    with pytest.raises(ffi.MinarrowHipError):  # "consolidate called on empty table set"
        ctx.consolidate_table_arena([8], [], [[]], ctx.alloc(64), 64)
