"""CPU tests of the oracle's sum restatements (benches/benchmark_parallel_simd.rs:44-98,
benches/hotloop_benchmark_simd.rs:56-174, benches/hotloop_benchmark_std.rs:49-57).

The reference never asserts a sum; the pins are the closed forms of its bench inputs."""
import math

import numpy as np
import pytest


@pytest.mark.parametrize("n", [0, 1, 3, 4, 5, 1000, 4097, 1_000_000])
def test_iota_closed_forms(oracle, n):
    expect = n * (n - 1) // 2
    a = np.arange(n, dtype=np.int64)
    f = a.astype(np.float64)
    for lanes in (1, 2, 4, 8, 16):
        assert oracle.simd_sum(a, lanes) == expect
        assert oracle.simd_sum(f, lanes) == float(expect)      # all partials < 2^53: exact in any order
        assert oracle.simd_sum_unrolled4(f, lanes) == float(expect)
    assert oracle.sum_scalar(a) == expect and oracle.sum_scalar(f) == float(expect)
    assert oracle.chunked_sum(a, 1 << 10, 4) == expect and oracle.chunked_sum(f, 1 << 10, 4) == float(expect)
    for threads in (1, 2, 5):
        assert oracle.par_sum(a, 1 << 12, 4, threads) == expect
        assert oracle.par_sum(f, 1 << 12, 4, threads) == float(expect)


def test_integer_sum_wraps_like_release_rust(oracle):
    a = np.array([np.iinfo(np.int64).max, 1, 5], dtype=np.int64)
    assert oracle.sum_scalar(a) == np.iinfo(np.int64).min + 5
    assert oracle.simd_sum(a, 4) == np.iinfo(np.int64).min + 5


def test_float_sum_order_is_the_reference_order(oracle):
    """4-lane accumulate, then ordered horizontal add from -0.0, then the scalar tail."""
    a = np.array([1e16, 1.0, -1e16, 1.0, 1.0, 1.0, 1.0, 1.0, 3.0], dtype=np.float64)
    lanes = [a[0] + a[4], a[1] + a[5], a[2] + a[6], a[3] + a[7]]
    expect = (((-0.0 + lanes[0]) + lanes[1]) + lanes[2]) + lanes[3] + a[8]
    assert oracle.simd_sum(a, 4) == expect
    assert oracle.simd_sum(np.array([-0.0, -0.0, -0.0, -0.0]), 4) == 0.0  # acc starts at +0.0 per lane
    assert math.copysign(1.0, oracle.simd_sum(np.array([], dtype=np.float64), 4)) == 1.0  # -0.0 + (+0.0 lanes)
    # parallel == sequential chunk order (deterministic restatement of Rayon's unspecified tree)
    rng = np.random.default_rng(0)
    b = rng.standard_normal(100_000) * 1e10
    assert oracle.par_sum(b, 1 << 10, 4, 4) == oracle.chunked_sum(b, 1 << 10, 4)


@pytest.mark.parametrize("off", [0, 1, 7, 9, 64])
def test_masked_sum_definition(oracle, off):
    rng = np.random.default_rng(off)
    n = 1000
    a = rng.integers(-(1 << 62), 1 << 62, size=n, dtype=np.int64)
    bits = rng.integers(0, 256, size=(n + off) // 8 + 2, dtype=np.uint8)
    valid = np.unpackbits(bits, bitorder="little")[off:off + n].astype(bool)
    s, c = oracle.masked_sum(a, bits, off)
    assert c == int(valid.sum())
    assert s == int(a[valid].sum())
    f = rng.standard_normal(n)
    s, c = oracle.masked_sum(f, bits, off)
    assert c == int(valid.sum()) and abs(s - math.fsum(f[valid])) < 1e-9
