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
