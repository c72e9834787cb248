"""Pins the oracle's consolidate restatement (src/traits/consolidate.rs:80-207) to the reference's SuperTable tests
(tests/golden/consolidate_kat.json from src/structs/chunked/super_table.rs:1305-1397). CPU only."""
import json
from pathlib import Path

import numpy as np

KAT = json.loads((Path(__file__).resolve().parent / "golden" / "consolidate_kat.json").read_text())
NP = {"i32": np.int32, "f64": np.float64}


def test_integer_and_float_kat(oracle):
    for col in KAT["integer_and_float"]["columns"]:
        chunks = [np.array(c, dtype=NP[col["type"]]) for c in col["chunks"]]
        out, mask = oracle.consolidate_column(chunks)
        assert mask is None
        np.testing.assert_array_equal(out, np.array(col["expect"], dtype=NP[col["type"]]))


def test_nullable_kat(oracle):
    c = KAT["nullable"]
    chunks = [np.array(x, dtype=np.int32) for x in c["chunks"]]
    masks = [oracle.pack_bits(v) for v in c["validity"]]
    out, mask = oracle.consolidate_column(chunks, masks, [0, 0])
    valid = oracle.unpack_bits(mask, 5)
    got = [int(v) if ok else None for v, ok in zip(out, valid)]
    assert got == c["expect_get"]


def test_maskless_chunk_is_all_valid(oracle):
    a, b = np.arange(70, dtype=np.int64), np.arange(5, dtype=np.int64)
    m = oracle.pack_bits(np.arange(70) % 3 == 0)
    out, mask = oracle.consolidate_column([a, b], [m, None], [0, 0])
    valid = oracle.unpack_bits(mask, 75)
    np.testing.assert_array_equal(valid[:70], np.arange(70) % 3 == 0)
    assert valid[70:].all()
    np.testing.assert_array_equal(out, np.concatenate([a, b]))


# ---- bit-packed columns (BooleanArray data / Bitmask::extend_from_bitmask_range, src/structs/bitmask.rs:520-592) ----

def _bits(oracle, bools):
    return oracle.pack_bits(np.array(bools, dtype=bool)) if len(bools) else np.zeros(8, dtype=np.uint8)


def test_boolean_extend_from_slice_kat(oracle):
    for c in KAT["boolean"]["extend_from_slice"]["cases"]:
        start = (_bits(oracle, c["start"]), 0, len(c["start"]))
        src = np.zeros(8, dtype=np.uint8)
        src[: len(c["src_bytes"])] = c["src_bytes"]
        out, mask = oracle.consolidate_boolean_column([start, (src, 0, c["len"])])
        assert mask is None
        assert oracle.unpack_bits(out, len(c["expect"])).tolist() == c["expect"]


def test_boolean_concat_kats(oracle):
    b = KAT["boolean"]
    c = b["concat"]
    out, mask = oracle.consolidate_boolean_column([(_bits(oracle, x), 0, len(x)) for x in c["chunks"]])
    assert mask is None and oracle.unpack_bits(out, 5).tolist() == c["expect"]
    c = b["concat_with_nulls"]
    out, mask = oracle.consolidate_boolean_column([(_bits(oracle, x), 0, len(x)) for x in c["chunks"]],
                                                  [(_bits(oracle, v), 0) for v in c["validity"]])
    data, valid = oracle.unpack_bits(out, 5), oracle.unpack_bits(mask, 5)
    assert [bool(d) if ok else None for d, ok in zip(data, valid)] == c["expect_get"]
    assert int((~valid).sum()) == c["null_count"]
    c = b["append_mask_onto_maskless"]
    out, mask = oracle.consolidate_boolean_column([(_bits(oracle, x), 0, len(x)) for x in c["chunks"]],
                                                  [(_bits(oracle, v), 0) if v is not None else None for v in c["validity"]])
    assert oracle.unpack_bits(out, 5).tolist() == c["expect_data"]
    assert oracle.unpack_bits(mask, 5).tolist() == c["expect_validity"]


def test_boolean_ranges_match_bit_concatenation(oracle):
    """All three paths of the reference (byte-aligned source and destination, shifted source, bit-level append)
    against plain concatenation of unpacked bits."""
    rng = np.random.default_rng(12)
    for trial in range(40):
        k = int(rng.integers(1, 7))
        chunks, masks, want, want_valid = [], [], [], []
        any_mask = False
        for i in range(k):
            n_bits = int(rng.integers(1, 400))
            src = rng.integers(0, 256, size=(n_bits + 7) // 8 + 9, dtype=np.uint8)
            off = int(rng.choice([0, 8, 64, rng.integers(0, 70)]))
            ln = int(rng.choice([0, 8, 64, rng.integers(0, n_bits + 1)]))
            ln = min(ln, src.size * 8 - off - 8)
            chunks.append((src, off, ln))
            want.append(np.unpackbits(src, bitorder="little")[off:off + ln])
            if rng.random() < 0.5:
                m = rng.integers(0, 256, size=src.size, dtype=np.uint8)
                moff = int(rng.integers(0, 9))
                masks.append((m, moff))
                want_valid.append(np.unpackbits(m, bitorder="little")[moff:moff + ln])
                any_mask = True
            else:
                masks.append(None)
                want_valid.append(np.ones(ln, dtype=np.uint8))
        out, mask = oracle.consolidate_boolean_column(chunks, masks)
        total = sum(c[2] for c in chunks)
        np.testing.assert_array_equal(np.unpackbits(out, bitorder="little")[:total], np.concatenate(want))
        assert not np.unpackbits(out, bitorder="little")[total:].any()  # mask_trailing_bits
        if any_mask:
            np.testing.assert_array_equal(np.unpackbits(mask, bitorder="little")[:total], np.concatenate(want_valid))
        else:
            assert mask is None


# ---- arena consolidation (src/structs/arena.rs) ----------------------------------------------------------------

def test_arena_cursor_rule_kats(oracle):
    a = KAT["arena"]
    offs, used = oracle.arena_regions([(r["elem"], r["count"]) for r in a["alignment"]["regions"]])
    assert offs == a["alignment"]["expect_offsets"]
    assert oracle.arena_regions([(1, 3)])[1] == a["used_after_three_bytes"]["expect_used"]
    offs, _ = oracle.arena_regions([(r["elem"], r["count"]) for r in a["multiple_types"]["regions"]])
    assert offs == a["multiple_types"]["expect_offsets"]
    assert oracle.arena_capacity_for_regions(a["capacity_for_regions"]["entries"]) == a["capacity_for_regions"]["expect"]


def test_arena_full_table_and_many_small(oracle):
    a = KAT["arena"]["full_table"]
    ids = np.array(a["ids"], dtype=np.int64)
    prices = np.array(a["prices"], dtype=np.float64)
    arena, d_off, m_off, used = oracle.consolidate_table_arena(
        [([ids], None, None), ([prices], [oracle.pack_bits(a["price_validity"])], [0])])
    assert (d_off, m_off, used) == ([a["expect_offsets"]["ids"], a["expect_offsets"]["prices"]],
                                    [None, a["expect_offsets"]["price_mask"]], a["expect_used"])
    np.testing.assert_array_equal(arena[0:40].view(np.int64), ids)
    np.testing.assert_array_equal(arena[64:104].view(np.float64), prices)
    assert oracle.unpack_bits(arena[128:136], 5).tolist() == a["price_validity"]
    m = KAT["arena"]["many_small"]
    cols = [([np.arange(m["rows"], dtype=np.int64) + i * m["rows"]], [oracle.pack_bits(np.ones(m["rows"], bool))], [0])
            for i in range(m["columns"])]
    arena, d_off, m_off, used = oracle.consolidate_table_arena(cols)
    assert d_off == m["expect_data_offsets"] and m_off == m["expect_mask_offsets"]
    assert [int(arena[o:o + 8].view(np.int64)[0]) for o in d_off] == m["expect_first_values"]
    assert all(oracle.unpack_bits(arena[o:o + 16], m["rows"]).all() for o in m_off)
    assert arena.size % 64 == 0 and used <= arena.size


def test_arena_equals_per_column_concat(oracle):
    """Both consolidate paths of the reference give the same columns (super_table.rs:657-743): the arena only changes
    where they live."""
    rng = np.random.default_rng(4)
    rows = [int(rng.integers(0, 300)) for _ in range(5)]
    cols = []
    for dt in (np.int64, np.float32, np.uint8, np.int16, np.float64):
        chunks = [rng.integers(0, 100, size=r).astype(dt) for r in rows]
        masks = [None if rng.random() < 0.4 else rng.integers(0, 256, size=r // 8 + 10, dtype=np.uint8) for r in rows]
        offs = [int(rng.integers(0, 9)) for _ in rows]
        cols.append((chunks, masks if dt != np.uint8 else None, offs if dt != np.uint8 else None))
    arena, d_off, m_off, used = oracle.consolidate_table_arena(cols)
    n = sum(rows)
    for (chunks, masks, offs), do, mo in zip(cols, d_off, m_off):
        want, want_bits = oracle.consolidate_column(chunks, masks, offs)
        np.testing.assert_array_equal(arena[do:do + want.nbytes].view(want.dtype), want)
        assert do % 64 == 0
        if want_bits is None:
            assert mo is None
        else:
            assert mo % 64 == 0
            np.testing.assert_array_equal(arena[mo:mo + (n + 7) // 8], want_bits[:(n + 7) // 8])


def test_product_arena_layout_matches_the_restatement(oracle):
    """ma_arena_layout is host arithmetic (no device call): the product's layout against the oracle's cursor rule."""
    from minarrow_amd.host import arena_layout

    rng = np.random.default_rng(8)
    for _ in range(50):
        n_cols = int(rng.integers(1, 12))
        elem = [int(rng.choice([1, 2, 4, 8])) for _ in range(n_cols)]
        nulls = [bool(rng.integers(0, 2)) for _ in range(n_cols)]
        n_rows = int(rng.choice([0, 1, 7, 8, 9, 63, 64, 65, 1000, 10_000, 123_457]))
        regions = []
        for e, h in zip(elem, nulls):
            regions.append((e, n_rows))
            if h:
                regions.append((1, (n_rows + 7) // 8))
        offs, used = oracle.arena_regions(regions)
        d, m, cap, got_used = arena_layout(elem, nulls, n_rows)
        it = iter(offs)
        want_d, want_m = [], []
        for h in nulls:
            want_d.append(next(it))
            want_m.append(next(it) if h else None)
        assert (d, m, got_used) == (want_d, want_m, used)
        assert cap == oracle.arena_capacity_for_regions([(n, e) for e, n in regions])
