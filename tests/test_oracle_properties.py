"""CPU property tests of the oracle itself, in the places the reference's own tests never go (every
reference mask test has <= 8 elements — no 64-bit word boundary, no tail with a mask, no window offset):

  * the SIMD-structured restatement (src/kernels/arithmetic/simd.rs) and the scalar restatement (std.rs) agree on
    random inputs for every lane table build.rs can emit, across word boundaries and ragged tails;
  * simd_mask / write_simd_mask_bits (src/utils.rs:221-283) round-trip at every offset;
  * the bitmask restatements agree with numpy bit logic where the reference's window granularity allows it."""
import zlib

import numpy as np
import pytest

INT_TYPES = [np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.int64, np.uint64]
OPS = ["add", "subtract", "multiply", "divide", "remainder", "power", "floordiv"]


def rand_ints(rng, dt, n, small=False):
    info = np.iinfo(dt)
    if small:
        return rng.integers(0 if info.min == 0 else -5, 12, size=n).astype(dt)
    a = rng.integers(info.min, info.max, size=n, dtype=dt, endpoint=True)
    if n:
        a[rng.integers(0, n, size=max(1, n // 8))] = 0
    return a


@pytest.mark.parametrize("dt", INT_TYPES)
@pytest.mark.parametrize("op", OPS)
def test_int_simd_and_std_restatements_agree(oracle, dt, op):
    rng = np.random.default_rng(zlib.crc32(f"{np.dtype(dt).name}-{op}".encode()))
    lanes_set = {oracle.LANES[t][np.dtype(dt).itemsize] for t in oracle.LANES}
    for n in (0, 1, 7, 63, 64, 65, 130, 1000):
        lhs = oracle.aligned_copy(rand_ints(rng, dt, n))
        rhs = oracle.aligned_copy(rand_ints(rng, dt, n, small=(op == "power")))
        mask = oracle.pack_bits(rng.random(n) < 0.7)
        st_std, out_std, om_std = oracle.int_body("masked_std", lhs, rhs, op, mask=mask)
        for lanes in lanes_set:
            st, out, om = oracle.int_body("masked_simd", lhs, rhs, op, mask=mask, lanes=lanes)
            np.testing.assert_array_equal(out, out_std)
            np.testing.assert_array_equal(oracle.unpack_bits(om, n), oracle.unpack_bits(om_std, n))
            assert st & ~oracle.PANIC_OVERFLOW == 0
        # all-valid mask takes the dense branch of the SIMD body (simd.rs:151-266): same results
        ones = oracle.pack_bits(np.ones(n, dtype=bool))
        _, out_a, om_a = oracle.int_body("masked_std", lhs, rhs, op, mask=ones)
        for lanes in lanes_set:
            _, out_b, om_b = oracle.int_body("masked_simd", lhs, rhs, op, mask=ones, lanes=lanes)
            np.testing.assert_array_equal(out_a, out_b)
            np.testing.assert_array_equal(oracle.unpack_bits(om_a, n), oracle.unpack_bits(om_b, n))
        # dense bodies agree where nothing panics
        safe_rhs = rhs.copy()
        if op in ("divide", "remainder", "floordiv"):
            safe_rhs[safe_rhs == 0] = 1
        s1, o1, _ = oracle.int_body("dense_std", lhs, safe_rhs, op)
        for lanes in lanes_set:
            s2, o2, _ = oracle.int_body("dense_simd", lhs, safe_rhs, op, lanes=lanes)
            np.testing.assert_array_equal(o1, o2)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_float_simd_and_std_restatements_agree(oracle, dt):
    rng = np.random.default_rng(5)
    for n in (0, 3, 64, 65, 1000):
        lhs = oracle.aligned_copy((rng.standard_normal(n) * 100).astype(dt))
        rhs = oracle.aligned_copy((rng.standard_normal(n) * 3).astype(dt))
        mask = oracle.pack_bits(rng.random(n) < 0.5)
        for op in ("add", "subtract", "multiply", "divide", "remainder", "floordiv"):
            _, a, ma = oracle.float_body("masked_std", lhs, rhs, op, mask=mask)
            for lanes in (2, 4, 8, 16):
                _, b, mb = oracle.float_body("masked_simd", lhs, rhs, op, mask=mask, lanes=lanes)
                np.testing.assert_array_equal(a.view(np.uint8), b.view(np.uint8))
                np.testing.assert_array_equal(oracle.unpack_bits(ma, n), oracle.unpack_bits(mb, n))


def test_simd_mask_and_write_round_trip(oracle):
    """src/utils.rs:221-283 at every offset of a 200-bit mask, lane counts 2..64."""
    rng = np.random.default_rng(1)
    n = 200
    valid = rng.random(n) < 0.5
    bits = oracle.pack_bits(valid)
    for lanes in (2, 4, 8, 16, 32, 64):
        out = oracle.pack_bits(np.zeros(n + 64, dtype=bool))
        for off in range(0, n - lanes + 1):
            m = oracle.simd_mask_bits(bits, n, off, n, lanes)
            assert m == sum(int(valid[off + l]) << l for l in range(lanes)), (lanes, off)
        for off in range(0, n - lanes + 1, lanes):
            oracle.write_mask_bits(out, off, oracle.simd_mask_bits(bits, n, off, n, lanes), lanes)
        covered = (n // lanes) * lanes
        np.testing.assert_array_equal(oracle.unpack_bits(out, covered), valid[:covered])
        # lanes beyond `len` are cleared
        assert oracle.simd_mask_bits(bits, n, n - 3, n, lanes) == sum(int(valid[n - 3 + l]) << l for l in range(min(3, lanes)))


def test_bitmask_restatements_vs_numpy(oracle):
    rng = np.random.default_rng(2)
    for n in (1, 63, 64, 65, 200, 1000):
        a, b = rng.random(n + 64) < 0.5, rng.random(n + 64) < 0.5
        pa, pb = oracle.pack_bits(a), oracle.pack_bits(b)
        for off in (0, 8, 64):  # byte-aligned windows are exact bit logic
            for op, fn in (("and", np.logical_and), ("or", np.logical_or), ("xor", np.logical_xor)):
                got = oracle.unpack_bits(oracle.bitmask_binop(op, pa, off, pb, off, n), n)
                np.testing.assert_array_equal(got, fn(a[off:off + n], b[off:off + n]))
            np.testing.assert_array_equal(oracle.unpack_bits(oracle.bitmask_not(pa, off, n), n), ~a[off:off + n])
        assert oracle.bitmask_popcount(pa, 0, n) == int(a[:n].sum()) == oracle.count_ones(pa, n)
        assert oracle.bitmask_popcount(pa, 64, n) == int(a[64:64 + n].sum())
        assert oracle.all_true(pa, n, None) == bool(a[:n].all())
        assert oracle.all_false(pa, n, None) == (not a[:n].any())
        np.testing.assert_array_equal(oracle.unpack_bits(oracle.merge_bitmasks(pa, pb, n), n), a[:n] & b[:n])
        np.testing.assert_array_equal(oracle.unpack_bits(oracle.bitmask_union(pa, pb, n), n), a[:n] | b[:n])
        # the documented SIMD quirk: a fully valid mask whose length is not a multiple of 64 reports "not all true"
        # when n_words % LANES == 0 (src/kernels/bitmask/simd.rs:668-674) - callers only lose the dense fast path
    ones = oracle.pack_bits(np.ones(2 * 64 - 5, dtype=bool))
    assert oracle.all_true(ones, 2 * 64 - 5, None) is True
    assert oracle.all_true(ones, 2 * 64 - 5, 2) is False
    assert oracle.all_true(ones, 2 * 64 - 5, 4) is True
