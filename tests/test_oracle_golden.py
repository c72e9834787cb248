"""Pins the CPU oracle (oracle/minarrow_oracle.c) to every known-answer vector the reference's own tests hold
for the hot path (tests/golden/*.json, transcribed from src/kernels/arithmetic/mod.rs, src/kernels/bitmask/
{mod,std,simd}.rs, src/kernels/broadcast/*.rs, src/kernels/routing/binary_map.rs). CPU only.

Each vector is run through BOTH restated bodies: 64-byte aligned inputs take the SIMD restatement
(src/kernels/arithmetic/simd.rs) for every lane table build.rs can emit, inputs shifted off the boundary take
the scalar restatement (src/kernels/arithmetic/std.rs) — the same switch as dispatch.rs:86.
"""
import json
import math
from pathlib import Path

import numpy as np
import pytest

GOLD = Path(__file__).resolve().parent / "golden"
ARITH = json.loads((GOLD / "arithmetic_kat.json").read_text())
BITS = json.loads((GOLD / "bitmask_kat.json").read_text())
ROUTE = json.loads((GOLD / "routing_kat.json").read_text())
SUMS = json.loads((GOLD / "sums_kat.json").read_text())

NP = {"i8": np.int8, "u8": np.uint8, "i16": np.int16, "u16": np.uint16, "i32": np.int32, "u32": np.uint32,
      "i64": np.int64, "u64": np.uint64, "f32": np.float32, "f64": np.float64}
LANE_TABLES = ("avx512", "avx2", "sse2")


def lanes_for(o, tag, table):
    return o.LANES[table][np.dtype(NP[tag]).itemsize]


def placements(o, arr):
    """(aligned copy -> SIMD body, misaligned copy -> scalar body)"""
    return [(o.aligned_copy(arr), True), (o.aligned_copy(arr, offset_bytes=arr.dtype.itemsize), False)]


@pytest.mark.parametrize("tag", ARITH["int_dense"]["types"])
@pytest.mark.parametrize("table", LANE_TABLES)
def test_int_dense_kat(oracle, tag, table):
    g = ARITH["int_dense"]
    lhs0, rhs0 = np.array(g["lhs"], dtype=NP[tag]), np.array(g["rhs"], dtype=NP[tag])
    lanes = lanes_for(oracle, tag, table)
    for (lhs, simd_l), (rhs, _) in zip(placements(oracle, lhs0), placements(oracle, rhs0)):
        for op, expect in g["expect"].items():
            st, out, out_mask, used_simd = oracle.apply_int(lhs, rhs, op, lanes=lanes)
            assert st == 0 and out_mask is None and used_simd == simd_l
            np.testing.assert_array_equal(out, np.array(expect, dtype=NP[tag]))
        # Power: 1 wrapping_mul a, b times
        expect = []
        for a, b in zip(g["lhs"], g["rhs"]):
            acc = 1
            for _ in range(b):
                acc = (acc * a) % (1 << (8 * np.dtype(NP[tag]).itemsize))
            expect.append(acc)
        st, out, _, _ = oracle.apply_int(lhs, rhs, "power", lanes=lanes)
        assert st == 0
        np.testing.assert_array_equal(out.astype(object) % (1 << (8 * out.itemsize)), np.array(expect, dtype=object))
        # Division by zero must panic (status PANIC_DIV_ZERO restates the panic)
        zero = oracle.aligned_copy(np.array(g["divide_by_zero_rhs"], dtype=NP[tag]), offset_bytes=0 if simd_l else lhs.itemsize)
        for op in g["divide_by_zero_ops_must_panic"]:
            st, _, _, _ = oracle.apply_int(lhs, zero, op, lanes=lanes)
            assert st & oracle.PANIC_DIV_ZERO


@pytest.mark.parametrize("tag", ARITH["int_masked"]["types"])
@pytest.mark.parametrize("table", LANE_TABLES)
def test_int_masked_kat(oracle, tag, table):
    lanes = lanes_for(oracle, tag, table)
    for case in ARITH["int_masked"]["cases"]:
        lhs0, rhs0 = np.array(case["lhs"], dtype=NP[tag]), np.array(case["rhs"], dtype=NP[tag])
        mask = oracle.pack_bits(case["mask"])
        for (lhs, simd_l), (rhs, _) in zip(placements(oracle, lhs0), placements(oracle, rhs0)):
            st, out, out_mask, used = oracle.apply_int(lhs, rhs, case["op"], mask=mask, lanes=lanes)
            assert st == 0 and used == simd_l
            np.testing.assert_array_equal(out, np.array(case["expect"], dtype=NP[tag]))
            np.testing.assert_array_equal(oracle.unpack_bits(out_mask, len(case["mask"])), case["expect_mask"])
            # the individual bodies agree with the dispatcher
            for kind in ("masked_std", "masked_simd"):
                st2, out2, om2 = oracle.int_body(kind, lhs, rhs, case["op"], mask=mask, lanes=lanes)
                assert st2 == 0
                np.testing.assert_array_equal(out2, out)
                np.testing.assert_array_equal(oracle.unpack_bits(om2, lhs.size), case["expect_mask"])


def test_int_empty_kat(oracle):
    g = ARITH["int_empty"]
    for tag in ("i32", "u32", "i64", "u64"):
        e = np.array([], dtype=NP[tag])
        st, out, _, _ = oracle.apply_int(e, e, g["op"])
        assert st == 0 and out.size == 0
    assert oracle.apply_int(np.zeros(3, np.int64), np.zeros(2, np.int64), "add")[0] == oracle.LENGTH_MISMATCH


@pytest.mark.parametrize("tag", ARITH["float"]["types"])
@pytest.mark.parametrize("table", LANE_TABLES)
def test_float_kat(oracle, tag, table):
    g = ARITH["float"]
    lanes = lanes_for(oracle, tag, table)
    eps = g["eps"][tag]
    lhs0, rhs0 = np.array(g["lhs"], dtype=NP[tag]), np.array(g["rhs"], dtype=NP[tag])
    for (lhs, _), (rhs, _) in zip(placements(oracle, lhs0), placements(oracle, rhs0)):
        for op, expect in g["expect_exact"].items():
            st, out, _, _ = oracle.apply_float(lhs, rhs, op, lanes=lanes)
            assert st == 0
            np.testing.assert_array_equal(out, np.array(expect, dtype=NP[tag]))  # reference uses assert_eq!
        st, out, _, _ = oracle.apply_float(lhs, rhs, "remainder", lanes=lanes)
        assert np.all(np.abs(out - np.array(g["expect_within_eps"]["remainder"], dtype=NP[tag])) < eps)
        st, out, _, _ = oracle.apply_float(lhs, rhs, "power", lanes=lanes)
        expect = np.array([math.exp(b * math.log(a)) for a, b in zip(g["lhs"], g["rhs"])])
        assert np.all(np.abs(out.astype(np.float64) - expect) < max(eps, 1e-5 if tag == "f32" else eps) * np.maximum(1, expect))
        zero = oracle.aligned_copy(np.array(g["divide_by_zero_rhs"], dtype=NP[tag]))
        assert np.all(np.isinf(oracle.apply_float(lhs, zero, "divide", lanes=lanes)[1]))
        assert np.all(np.isnan(oracle.apply_float(lhs, zero, "remainder", lanes=lanes)[1]))
        m = g["masked"]
        st, out, out_mask, _ = oracle.apply_float(lhs, rhs, m["op"], mask=oracle.pack_bits(m["mask"]), lanes=lanes)
        np.testing.assert_array_equal(out, np.array(m["expect"], dtype=NP[tag]))
        np.testing.assert_array_equal(oracle.unpack_bits(out_mask, 4), m["mask"])
    e = np.array([], dtype=NP[tag])
    assert oracle.apply_float(e, e, "add")[1].size == 0


@pytest.mark.parametrize("tag", ARITH["fma"]["types"])
def test_fma_kat(oracle, tag):
    g = ARITH["fma"]
    a, b, c = (oracle.aligned_copy(np.array(g[k], dtype=NP[tag])) for k in ("lhs", "rhs", "acc"))
    for unfused in (False, True):
        st, out, om = oracle.apply_fma(a, b, c, force_unfused=unfused)
        assert st == 0 and om is None
        np.testing.assert_array_equal(out, np.array(g["expect"], dtype=NP[tag]))
        st, out, om = oracle.apply_fma(a, b, c, mask=oracle.pack_bits(g["masked"]["mask"]), force_unfused=unfused)
        np.testing.assert_array_equal(out, np.array(g["masked"]["expect"], dtype=NP[tag]))
        np.testing.assert_array_equal(oracle.unpack_bits(om, 3), g["masked"]["expect_mask"])
    e = np.array([], dtype=NP[tag])
    assert oracle.apply_fma(e, e, e)[1].size == 0
    assert oracle.apply_fma(a, b, c[:2].copy())[0] == oracle.LENGTH_MISMATCH
    # fused vs unfused differ exactly where the product needs more than one rounding
    x = oracle.aligned_copy(np.array([1 + 2.0 ** -30], dtype=np.float64))
    y = oracle.aligned_copy(np.array([1 - 2.0 ** -30], dtype=np.float64))
    z = oracle.aligned_copy(np.array([-1.0], dtype=np.float64))
    assert oracle.apply_fma(x, y, z)[1][0] == -(2.0 ** -60) and oracle.apply_fma(x, y, z, force_unfused=True)[1][0] == 0.0


def test_merge_masks_kat(oracle):
    g = ARITH["merge_masks"]
    out = oracle.merge_bitmasks(oracle.pack_bits(g["a"]), oracle.pack_bits(g["b"]), 4)
    np.testing.assert_array_equal(oracle.unpack_bits(out, 4), g["expect"])
    assert oracle.merge_bitmasks(None, None, 4) is None
    np.testing.assert_array_equal(oracle.unpack_bits(oracle.merge_bitmasks(oracle.pack_bits(g["a"]), None, 4), 4), g["a"])


def test_datetime_kat(oracle):
    """apply_datetime_i64 = merge_bitmasks_to_new (AND) + the integer kernels (dispatch.rs:309-372)."""
    g = ARITH["datetime_i64"]
    for case in g["cases"]:
        lhs = oracle.aligned_copy(np.array(case["lhs"], dtype=np.int64))
        rhs = oracle.aligned_copy(np.array(case["rhs"], dtype=np.int64))
        st, out, _ = oracle.int_body("dense_simd", lhs, rhs, case["op"], lanes=8)
        assert st == 0
        np.testing.assert_array_equal(out, case["expect"])
    m = g["masked"]
    merged = oracle.merge_bitmasks(oracle.pack_bits(m["lhs_mask"]), None, 4)
    lhs = oracle.aligned_copy(np.array(m["lhs"], dtype=np.int64))
    rhs = oracle.aligned_copy(np.array(m["rhs"], dtype=np.int64))
    st, out, om = oracle.int_body("masked_simd", lhs, rhs, m["op"], mask=merged, lanes=8)
    np.testing.assert_array_equal(out, m["expect"])
    np.testing.assert_array_equal(oracle.unpack_bits(om, 4), m["expect_mask"])


def test_int_power_short_vs_long_kat(oracle):
    g = ARITH["int_power_short_vs_long"]
    for n in g["lengths"]:
        lhs = oracle.aligned_copy(np.full(n, g["base"], dtype=np.uint32))
        rhs = oracle.aligned_copy(np.full(n, g["exp"], dtype=np.uint32))
        st, out, _ = oracle.int_body("dense_simd", lhs, rhs, "power", lanes=g["lanes"])
        assert st == 0 and np.all(out == g["expect_each"])


# ---- bitmask kernels -----------------------------------------------------------------------------------

@pytest.mark.parametrize("lanes", [8, 16, 32, 64])
def test_bitmask_simd_suite_kat(oracle, lanes):
    g = BITS["simd_suite"]
    a, b = g["and_or_xor"]["a"], g["and_or_xor"]["b"]
    pa, pb = oracle.pack_bits(a), oracle.pack_bits(b)
    for op, fn in (("and", np.logical_and), ("or", np.logical_or), ("xor", np.logical_xor)):
        np.testing.assert_array_equal(oracle.unpack_bits(oracle.bitmask_binop(op, pa, 0, pb, 0, 8), 8), fn(a, b))
    np.testing.assert_array_equal(oracle.unpack_bits(oracle.bitmask_not(oracle.pack_bits(g["not"]["a"]), 0, 4), 4), g["not"]["expect"])
    for c in g["in_mask"]:
        n = len(c["lhs"])
        out = oracle.bitmask_in(oracle.pack_bits(c["lhs"]), 0, oracle.pack_bits(c["rhs"]), 0, n)
        np.testing.assert_array_equal(oracle.unpack_bits(out, n), c["expect"])
    c = g["not_in_mask"]
    out = oracle.bitmask_not_in(oracle.pack_bits(c["lhs"]), 0, oracle.pack_bits(c["rhs"]), 0, 4)
    np.testing.assert_array_equal(oracle.unpack_bits(out, 4), c["expect"])
    c = g["eq_ne"]
    panics, eq = oracle.bitmask_eq(oracle.pack_bits(c["a"]), 0, oracle.pack_bits(c["b"]), 0, 4)
    assert not panics
    np.testing.assert_array_equal(oracle.unpack_bits(eq, 4), c["expect_eq"])
    _, ne = oracle.bitmask_eq(oracle.pack_bits(c["a"]), 0, oracle.pack_bits(c["b"]), 0, 4, negate=True)
    np.testing.assert_array_equal(oracle.unpack_bits(ne, 4), c["expect_ne"])
    c = g["all_eq"]
    pa = oracle.pack_bits(c["a"])
    assert oracle.bitmask_all_eq(pa, 0, pa.copy(), 0, 8) == 1
    flipped = list(c["a"])
    flipped[0] = not flipped[0]
    assert oracle.bitmask_all_eq(pa, 0, oracle.pack_bits(flipped), 0, 8) == 0
    c = g["all_ne"]
    assert oracle.bitmask_all_ne(oracle.pack_bits(c["a"]), 0, oracle.pack_bits(c["b"]), 0, 3) == 1
    assert oracle.bitmask_all_ne(oracle.pack_bits(c["a"]), 0, oracle.pack_bits(c["a"]), 0, 3) == 0
    assert oracle.bitmask_popcount(oracle.pack_bits(g["popcount"]["a"]), 0, 8) == g["popcount"]["expect"]
    n = 64 * lanes
    ones = oracle.pack_bits(np.ones(n, dtype=bool))
    assert oracle.all_true(ones, n, lanes) and not oracle.all_false(ones, n, lanes)
    ones[0] &= ~np.uint8(1 << 3)
    assert not oracle.all_true(ones, n, lanes)
    assert oracle.all_false(oracle.pack_bits(np.zeros(n, dtype=bool)), n, lanes)


def test_bitmask_std_suite_kat(oracle):
    g = BITS["std_suite"]
    for op in ("and", "or", "xor"):
        c = g[op]
        n = len(c["a"])
        out = oracle.bitmask_binop(op, oracle.pack_bits(c["a"]), 0, oracle.pack_bits(c["b"]), 0, n)
        np.testing.assert_array_equal(oracle.unpack_bits(out, n), c["expect"])
    np.testing.assert_array_equal(oracle.unpack_bits(oracle.bitmask_not(oracle.pack_bits(g["not"]["a"]), 0, 4), 4), g["not"]["expect"])
    for c in g["in_mask"]:
        out = oracle.bitmask_in(oracle.pack_bits(c["lhs"]), 0, oracle.pack_bits(c["rhs"]), 0, 3)
        np.testing.assert_array_equal(oracle.unpack_bits(out, 3), c["expect"])
    c = g["not_in_mask"]
    np.testing.assert_array_equal(
        oracle.unpack_bits(oracle.bitmask_not_in(oracle.pack_bits(c["lhs"]), 0, oracle.pack_bits(c["rhs"]), 0, 2), 2), c["expect"])
    np.testing.assert_array_equal(
        oracle.unpack_bits(oracle.bitmask_eq(oracle.pack_bits(g["eq"]["a"]), 0, oracle.pack_bits(g["eq"]["b"]), 0, 3)[1], 3), g["eq"]["expect"])
    np.testing.assert_array_equal(
        oracle.unpack_bits(oracle.bitmask_eq(oracle.pack_bits(g["ne"]["a"]), 0, oracle.pack_bits(g["ne"]["b"]), 0, 3, negate=True)[1], 3),
        g["ne"]["expect"])
    for c in g["all_eq"]:
        assert bool(oracle.bitmask_all_eq(oracle.pack_bits(c["a"]), 0, oracle.pack_bits(c["b"]), 0, len(c["a"]))) == c["expect"]
    for c in g["all_ne"]:
        assert bool(oracle.bitmask_all_ne(oracle.pack_bits(c["a"]), 0, oracle.pack_bits(c["b"]), 0, len(c["a"]))) == c["expect"]
    assert oracle.bitmask_popcount(oracle.pack_bits(g["popcount"]["a"]), 0, 6) == g["popcount"]["expect"]
    for c in g["all_true"]:
        assert oracle.all_true(oracle.pack_bits(c["a"]), len(c["a"]), None) == c["expect"]
        assert oracle.all_true(oracle.pack_bits(c["a"]), len(c["a"]), 8) == c["expect"]
    for c in g["all_false"]:
        assert oracle.all_false(oracle.pack_bits(c["a"]), len(c["a"]), None) == c["expect"]
        assert oracle.all_false(oracle.pack_bits(c["a"]), len(c["a"]), 8) == c["expect"]


def test_bitmask_struct_helpers_kat(oracle):
    g = BITS["mod_helpers"]
    for n, words in g["words_for"]:
        assert (n + 63) // 64 == words
    bits = oracle.pack_bits(np.zeros(128, dtype=bool))
    for i in g["words_view"]["set_bits"]:
        bits[i >> 3] |= np.uint8(1 << (i & 7))
    words = bits[:16].view(np.uint64)
    assert [f"{int(w):016X}" for w in words] == g["words_view"]["expect_words_hex"]
    raw = np.array([int(h, 16) for h in g["words_write"]["words_hex"]], dtype=np.uint64).view(np.uint8)
    for idx, val in g["words_write"]["expect_bytes"].items():
        assert raw[int(idx)] == val
    c = g["clear_trailing_bits"]
    buf = np.zeros(16, dtype=np.uint8)
    oracle.klib().mo_bitmask_new_set_all(buf.ctypes.data, c["len"], 1)
    assert buf[1] == c["expect_byte1"] and buf[0] == 0xFF and oracle.count_ones(buf, c["len"]) == c["len"]


# ---- routing / broadcast -------------------------------------------------------------------------------

def test_routing_kat(oracle):
    """A length-1 side is materialised to the other side's length and the ordinary kernel runs
    (src/kernels/routing/broadcast.rs:87-112); Array x Scalar goes through the same path
    (src/kernels/broadcast/array.rs:139-184)."""
    def run(op, lhs, rhs, dt):
        lhs, rhs = np.array(lhs, dtype=dt), np.array(rhs, dtype=dt)
        if lhs.size == 1 and rhs.size != 1:
            lhs = np.full(rhs.size, lhs[0], dtype=dt)
        if rhs.size == 1 and lhs.size != 1:
            rhs = np.full(lhs.size, rhs[0], dtype=dt)
        fn = oracle.apply_float if np.dtype(dt).kind == "f" else oracle.apply_int
        st, out, _, _ = fn(oracle.aligned_copy(lhs), oracle.aligned_copy(rhs), op)
        assert st == 0
        return out

    for c in ROUTE["array_array"]["cases"] + ROUTE["super_array"]["cases"]:
        np.testing.assert_array_equal(run(c["op"], c["lhs"], c["rhs"], np.int32), c["expect"])
    c = ROUTE["array_scalar"]
    np.testing.assert_array_equal(run(c["op"], c["lhs"], [c["scalar"]], np.int32), c["expect"])
    c = ROUTE["super_array"]["chunked_add"]
    for l, r, e in zip(c["lhs_chunks"], c["rhs_chunks"], c["expect_chunks"]):
        np.testing.assert_array_equal(run("add", l, r, np.int32), e)
    for c in ROUTE["binary_map_f64"]["cases"]:
        lhs = c.get("lhs", c.get("lhs_i32"))  # the i32 side is cast to f64 first (routing/arithmetic.rs:244-256)
        rhs = c.get("rhs", [c.get("scalar")])
        np.testing.assert_array_equal(run(c["op"], lhs, rhs, np.float64), c["expect"])


def test_sum_closed_forms_kat(oracle):
    for c in SUMS["iota"]:
        if c["n"] <= 1_000_000:
            a = np.arange(c["n"], dtype=np.int64)
            assert oracle.chunked_sum(a) == c["sum"] and oracle.simd_sum(a.astype(np.float64), 4) == float(c["sum"])
        assert c["n"] * (c["n"] - 1) // 2 == c["sum"]
        assert float(c["sum"]) == c["sum"]  # representable


# ---- Bitmask struct tests (src/structs/bitmask.rs:919-1108) -----------------------------------------------------
def _struct_kat():
    import json
    from pathlib import Path

    return json.loads((Path(__file__).resolve().parent / "golden" / "bitmask_struct_kat.json").read_text())


def test_bitmask_struct_vectors(oracle):
    k = _struct_kat()
    for c in k["count_and_all"]["cases"]:
        bits = oracle.pad_bits(oracle.pack_bits(c["bits"]), len(c["bits"]))
        n = len(c["bits"])
        assert oracle.count_ones(bits, n) == c["count_ones"] == oracle.bitmask_popcount(bits, 0, n)
        assert oracle.all_true(bits, n, lanes=None) == c["all_set"] and oracle.all_false(bits, n, lanes=None) == c["all_unset"]
    c = k["invert_union_intersect"]
    a, b = (oracle.pad_bits(oracle.pack_bits(c[x]), 8) for x in ("a", "b"))
    assert oracle.unpack_bits(oracle.bitmask_binop("or", a, 0, b, 0, 8), 8).tolist() == c["union"]
    assert oracle.unpack_bits(oracle.bitmask_union(a, b, 8), 8).tolist() == c["union"]
    assert oracle.unpack_bits(oracle.bitmask_binop("and", a, 0, b, 0, 8), 8).tolist() == c["intersect"]
    assert oracle.unpack_bits(oracle.bitmask_not(a, 0, 8), 8).tolist() == c["invert_a"]
    c = k["union_opt"]
    a, b = (oracle.pad_bits(oracle.pack_bits(c[x]), 4) for x in ("a", "b"))
    assert oracle.unpack_bits(oracle.bitmask_union(a, b, 4), 4).tolist() == c["expect"]
    c = k["concatenate"]
    out, mask = oracle.consolidate_boolean_column([(oracle.pad_bits(oracle.pack_bits(c["m1"]), 5), 0, 5),
                                                   (oracle.pad_bits(oracle.pack_bits(c["m2"]), 4), 0, 4)])
    assert mask is None and oracle.unpack_bits(out, 9).tolist() == c["expect"]
    c = k["slice_clone"]
    out, _ = oracle.consolidate_boolean_column([(oracle.pad_bits(oracle.pack_bits(c["bits"]), 10), c["offset"], c["len"])])
    assert oracle.unpack_bits(out, c["len"]).tolist() == c["expect"]
