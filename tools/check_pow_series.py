#!/usr/bin/env python3
"""Reference values for the hand-written ln / exp series behind float Power (minarrow_amd/csrc/ma_binary.hpp:
pow_f64_ln, pow_f32_ln, pow_f32_exp) — the reference computes `(rhs * lhs.ln()).exp()` (src/kernels/arithmetic/std.rs:153,
simd.rs:570,585) through the host libm; round 2 replaced the device libm calls with series and claimed accuracies that
nothing in the tree reproduced. This script does: pure Python (`decimal` at 80 significant digits = 265 bits, whose ln / exp
are correctly rounded at that precision), deterministic, and writes tests/golden/pow_series_kat.npz:

  ln64_x, ln64_hi, ln64_lo      f64 inputs x > 0 (log-uniform over the whole range incl. subnormals, a dense cloud around 1,
                                powers of two, the ends of the exponent range, the mantissa cut at 1/sqrt2 | sqrt2) and ln x as
                                hi = RN64(ln x), lo = RN32(ln x - hi): the exact value to ~2^-77, so that a result's error
                                is known to a thousandth of an ULP
  ln32_x, ln32_want             f32 inputs, RN32(ln x)
  exp32_y, exp32_want           f32 inputs in [-104, 89], RN32(exp y)
  pow32_a, pow32_b, pow32_want  f32 pairs and the exact three-rounding formulation RN32(exp(RN32(b * RN32(ln a)))) — what a
                                correctly rounded logf / expf would make of std.rs:153
  pow64_a, pow64_b, pow64_want  the same with f64 roundings

tests/test_gpu_pow_series.py evaluates the device functions on these inputs (ma_test_pow_series) and holds them to the
claims. Re-generate with `python tools/check_pow_series.py` (about two minutes); `--check` verifies the committed file
against a re-computation of a sample.
"""
import argparse
import sys
from decimal import Decimal, getcontext
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
OUT = ROOT / "tests" / "golden" / "pow_series_kat.npz"
getcontext().prec = 80


def rn64(d: Decimal) -> float:
    return float(d)  # decimal string -> strtod: correctly rounded (the 80-digit value is far inside any double rounding case)


def rn32(d: Decimal) -> np.float32:
    """Round-to-nearest-even of the exact value into f32 (no double rounding through f64)."""
    c = np.float32(float(d))
    if not np.isfinite(c):
        return c
    best, best_err = c, abs(d - Decimal(float(c)))
    for nb in (np.nextafter(c, np.float32(np.inf)), np.nextafter(c, np.float32(-np.inf))):
        if np.isfinite(nb):
            err = abs(d - Decimal(float(nb)))
            if err < best_err:
                best, best_err = nb, err
    return best


def ln64_inputs(rng, n):
    parts = []
    k = n // 2
    # log-uniform over [2^-1074, 2^1024): exponent uniform, mantissa uniform
    e = rng.integers(-1074, 1024, size=k)
    m = 1.0 + rng.random(k)
    parts.append(np.ldexp(m, e))
    # a dense cloud around 1, where ln cancels: 1 +- 2^-j (1 + u)
    j = rng.integers(1, 53, size=n // 4)
    parts.append(1.0 + np.ldexp(1.0 + rng.random(n // 4), -j) * rng.choice([-1.0, 1.0], size=n // 4))
    # [1/2, 2) uniformly: the reduced argument range, both sides of the 1/sqrt2 cut
    parts.append(0.5 + 1.5 * rng.random(n // 8))
    # powers of two and their neighbours, the ends of the exponent range
    ee = np.arange(-1074, 1024, dtype=np.int64)
    p2 = np.ldexp(1.0, ee)
    parts.append(p2)
    parts.append(np.nextafter(p2, np.inf))
    parts.append(np.nextafter(p2[1:], 0.0))
    edge = np.array([np.finfo(np.float64).tiny, np.finfo(np.float64).max, 5e-324, 2.2250738585072009e-308,
                     0.70710678118654746, 0.70710678118654757, 1.4142135623730949, 1.4142135623730951,
                     np.e, 1.0, 10.0, 0.1, 3.0], dtype=np.float64)
    parts.append(edge)
    x = np.concatenate(parts).astype(np.float64)
    x = x[(x > 0) & np.isfinite(x)]
    return x


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100_000, help="base sample count per table")
    ap.add_argument("--check", action="store_true", help="re-compute a sample of the committed file and compare")
    args = ap.parse_args()
    rng = np.random.default_rng(20261004)

    if args.check:
        z = np.load(OUT)
        idx = rng.integers(0, z["ln64_x"].size, size=500)
        for i in idx:
            d = Decimal(float(z["ln64_x"][i])).ln()
            hi = rn64(d)
            assert hi == z["ln64_hi"][i] and np.float32(rn64(d - Decimal(hi))) == z["ln64_lo"][i], i
        idx = rng.integers(0, z["pow32_a"].size, size=500)
        for i in idx:
            l = rn32(Decimal(float(z["pow32_a"][i])).ln())
            y = np.float32(l) * np.float32(z["pow32_b"][i])
            assert rn32(Decimal(float(y)).exp()) == z["pow32_want"][i], i
        print(f"{OUT}: sample re-computed, identical")
        return 0

    out = {}
    # ---- f64 ln
    x = ln64_inputs(rng, args.n * 115 // 100)  # + ~6 300 structured points: > 10^5 in all
    hi = np.empty_like(x)
    lo = np.empty_like(x)
    for i, v in enumerate(x):
        d = Decimal(float(v)).ln()
        hi[i] = rn64(d)
        lo[i] = rn64(d - Decimal(float(hi[i])))
        if i % 20000 == 0:
            print(f"ln64 {i}/{x.size}", file=sys.stderr, flush=True)
    out.update(ln64_x=x, ln64_hi=hi, ln64_lo=lo.astype(np.float32))  # the residual only needs a few digits
    # ---- f32 ln: log-uniform over the normal + subnormal f32 range, a cloud around 1
    n = args.n * 6 // 10
    e = rng.integers(-149, 128, size=n // 2)
    a32 = np.ldexp(1.0 + rng.random(n // 2), e).astype(np.float32)
    j = rng.integers(1, 24, size=n // 2)
    near = (1.0 + np.ldexp(1.0 + rng.random(n // 2), -j) * rng.choice([-1.0, 1.0], size=n // 2)).astype(np.float32)
    a32 = np.concatenate([a32, near])
    a32 = a32[(a32 > 0) & np.isfinite(a32)]
    out["ln32_x"] = a32
    out["ln32_want"] = np.array([rn32(Decimal(float(v)).ln()) for v in a32], dtype=np.float32)
    print("ln32 done", file=sys.stderr, flush=True)
    # ---- f32 exp over the whole finite-result range (expf underflows below -103.97, overflows above 88.72)
    y32 = np.concatenate([rng.uniform(-104.0, 89.0, size=n // 2), rng.standard_normal(n // 2) * 3.0,
                          np.ldexp(rng.standard_normal(n // 8), -rng.integers(0, 40, size=n // 8))]).astype(np.float32)
    out["exp32_y"] = y32
    out["exp32_want"] = np.array([rn32(Decimal(float(v)).exp()) for v in y32], dtype=np.float32)
    print("exp32 done", file=sys.stderr, flush=True)
    # ---- the three-rounding formulation, f32 and f64
    for tag, dt, rn in (("32", np.float32, rn32), ("64", np.float64, lambda d: np.float64(rn64(d)))):
        n = args.n * 6 // 10 if tag == "32" else args.n // 5
        a = np.exp(rng.uniform(-8, 8, size=n)).astype(dt)
        b = (rng.standard_normal(n) * 3).astype(dt)
        want = np.empty(n, dtype=dt)
        for i in range(n):
            l = dt(rn(Decimal(float(a[i])).ln()))
            y = dt(l * b[i])  # one IEEE multiplication in the working type
            want[i] = rn(Decimal(float(y)).exp())
            if i % 20000 == 0:
                print(f"pow{tag} {i}/{n}", file=sys.stderr, flush=True)
        out[f"pow{tag}_a"], out[f"pow{tag}_b"], out[f"pow{tag}_want"] = a, b, want
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT}: " + ", ".join(f"{k}[{v.size}]" for k, v in out.items()))
    return 0


if __name__ == "__main__":
    sys.exit(main())
