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
