"""GPU tests of the Arrow C Data Interface entry points (ma_sum_arrow, ma_mean_arrow, ma_apply_arrow).

Inputs come from PyArrow through `_export_to_c` — the protocol the reference's Python bridge uses
(pyo3/src/ffi/to_rust.rs:282) — including slices, whose non-zero `offset` the reference's own exporter never
produces (src/ffi/arrow_c_ffi.rs:1773) but other producers do. Expected values: the golden vectors of
tests/arrow_c_integration.rs, numpy / pyarrow.compute as an independent cross-check, and the CPU oracle."""
import json
import math
from pathlib import Path

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pytest

from minarrow_amd import ffi
from minarrow_amd.arrow_c import Exported

pytestmark = pytest.mark.gpu

KAT = json.loads((Path(__file__).resolve().parent / "golden" / "arrow_c_kat.json").read_text())
PA_TYPE = {"i": pa.int32(), "l": pa.int64(), "I": pa.uint32(), "L": pa.uint64(), "f": pa.float32(), "g": pa.float64()}
OPS = {"add": 0, "subtract": 1, "multiply": 2, "divide": 3}


def test_golden_arrays_sum(ctx):
    for case in KAT["cases"]:
        arr = pa.array(case["values"], type=PA_TYPE[case["format"]])
        vals = [v for v in case["values"] if v is not None]
        with Exported(arr) as ex:
            f, i, c = ctx.sum_arrow(ex.array_ptr, ex.schema_ptr)
            assert c == len(vals)
            if case["format"] in "ilIL":
                assert i == sum(vals) and f == float(sum(vals))
            else:
                exact = math.fsum(np.array(vals, dtype=arr.type.to_pandas_dtype()).astype(np.float64).tolist())
                assert abs(f - exact) <= math.ulp(exact)
            m, c2 = ctx.mean_arrow(ex.array_ptr, ex.schema_ptr)
            assert c2 == c and abs(m - f / c) <= 2 * math.ulp(f / c)


@pytest.mark.parametrize("fmt", list(PA_TYPE))
def test_random_arrays_with_nulls_and_slices(ctx, fmt):
    rng = np.random.default_rng(ord(fmt))
    n = 70_003
    np_dt = PA_TYPE[fmt].to_pandas_dtype()
    if fmt in "fg":
        vals = (rng.standard_normal(n) * 1e4).astype(np_dt)
    else:
        info = np.iinfo(np_dt)
        vals = rng.integers(info.min // 4, info.max // 4, size=n, dtype=np_dt)
    nulls = rng.random(n) < 0.1
    arr = pa.array(vals, type=PA_TYPE[fmt], mask=nulls)
    for sl in (arr, arr.slice(1, 50_000), arr.slice(67, 1000), arr.slice(64 * 5, 4096), pa.array(vals, type=PA_TYPE[fmt])):
        with Exported(sl) as ex:
            assert ex.array.offset == sl.offset
            f, i, c = ctx.sum_arrow(ex.array_ptr, ex.schema_ptr)
        keep = np.asarray(sl.is_valid())
        sel = np.asarray(sl.fill_null(0))[keep]
        assert c == int(keep.sum()) == len(sl) - sl.null_count
        if fmt in "fg":
            exact = math.fsum(sel.astype(np.float64).tolist())
            assert abs(f - exact) <= math.ulp(exact)
        else:
            want = int(sel.astype(object).sum())
            assert (i - want) % (1 << 64) == 0
            assert (pc.sum(sl).as_py() - want) % (1 << 64) == 0  # independent cross-check (pyarrow wraps too)


def test_unsupported_formats_are_rejected(ctx):
    for arr in (pa.array(["a", "b"]), pa.array([True, False]), pa.array(np.array([1, 2], dtype=np.float16)),
                pa.array([1, 2], type=pa.date32())):
        with Exported(arr) as ex:
            with pytest.raises(ffi.MinarrowHipError) as e:
                ctx.sum_arrow(ex.array_ptr, ex.schema_ptr)
            assert e.value.status == ffi.MA_ERR_UNSUPPORTED


def run_apply(ctx, op, lhs, rhs):
    n = max(len(lhs), len(rhs))
    np_dt = lhs.type.to_pandas_dtype()
    out = np.zeros(n, dtype=np_dt)
    validity = np.zeros(((n + 63) // 64) * 8 + 8, dtype=np.uint8)
    with Exported(lhs) as a, Exported(rhs) as b:
        has = ctx.apply_arrow(OPS[op], (a.array_ptr, a.schema_ptr), (b.array_ptr, b.schema_ptr), out, validity)
    valid = np.unpackbits(validity, bitorder="little")[:n].astype(bool) if has else np.ones(n, dtype=bool)
    return out, valid, has


@pytest.mark.parametrize("fmt", ["l", "i", "g", "f"])
def test_apply_arrow(ctx, fmt):
    rng = np.random.default_rng(5)
    n = 20_011
    np_dt = PA_TYPE[fmt].to_pandas_dtype()
    a_vals = rng.integers(1, 1000, size=n).astype(np_dt)
    b_vals = rng.integers(1, 1000, size=n).astype(np_dt)
    a_null, b_null = rng.random(n) < 0.1, rng.random(n) < 0.2
    cases = [
        (pa.array(a_vals, type=PA_TYPE[fmt]), pa.array(b_vals, type=PA_TYPE[fmt])),
        (pa.array(a_vals, type=PA_TYPE[fmt], mask=a_null), pa.array(b_vals, type=PA_TYPE[fmt])),
        (pa.array(a_vals, type=PA_TYPE[fmt]), pa.array(b_vals, type=PA_TYPE[fmt], mask=b_null)),
        (pa.array(a_vals, type=PA_TYPE[fmt], mask=a_null), pa.array(b_vals, type=PA_TYPE[fmt], mask=b_null)),
        (pa.array(a_vals, type=PA_TYPE[fmt], mask=a_null).slice(3, 9000), pa.array(b_vals, type=PA_TYPE[fmt], mask=b_null).slice(70, 9000)),
    ]
    for lhs, rhs in cases:
        for op, fn in (("add", pc.add), ("subtract", pc.subtract), ("multiply", pc.multiply)):
            out, valid, has = run_apply(ctx, op, lhs, rhs)
            want = fn(lhs, rhs)  # Arrow semantics: null if either side is null (= AND of validities)
            want_valid = np.asarray(want.is_valid())
            assert has == (lhs.null_count + rhs.null_count > 0)
            np.testing.assert_array_equal(valid, want_valid)
            np.testing.assert_array_equal(out[valid], np.asarray(want.fill_null(0))[want_valid])
            assert np.all(out[~valid] == 0)  # null slots hold 0 (src/kernels/arithmetic/simd.rs:315)
    # length-1 side is broadcast (src/kernels/routing/broadcast.rs:87-112), either side
    lhs = pa.array(a_vals, type=PA_TYPE[fmt])
    one = pa.array([7], type=PA_TYPE[fmt])
    out, valid, has = run_apply(ctx, "multiply", lhs, one)
    np.testing.assert_array_equal(out, a_vals * np_dt(7))
    out, valid, has = run_apply(ctx, "subtract", one, lhs)
    np.testing.assert_array_equal(out, np_dt(7) - a_vals)
    assert not has


def test_apply_arrow_errors(ctx):
    a = pa.array([1, 2, 3], type=pa.int64())
    with pytest.raises(ffi.MinarrowHipError) as e:
        run_apply(ctx, "add", a, pa.array([1, 2], type=pa.int64()))
    assert e.value.status == ffi.MA_ERR_LENGTH_MISMATCH and "cannot broadcast arrays of length 3 and 2" in e.value.message
    with pytest.raises(ffi.MinarrowHipError) as e:
        run_apply(ctx, "add", a, pa.array([1.0, 2.0, 3.0]))
    assert e.value.status == ffi.MA_ERR_UNSUPPORTED
    with pytest.raises(ffi.MinarrowHipError) as e:
        run_apply(ctx, "divide", a, pa.array([1, 0, 3], type=pa.int64()))
    assert e.value.status == ffi.MA_ERR_DIVIDE_BY_ZERO


@pytest.mark.parametrize("pa_type,code", [(pa.int8(), "c"), (pa.uint8(), "C"), (pa.int16(), "s"), (pa.uint16(), "S")])
def test_narrow_integer_arrays_reduce_but_do_not_route(ctx, pa_type, code):
    """The reference's extended_numeric_types as Arrow arrays ('c' 'C' 's' 'S'): ma_sum_arrow / ma_mean_arrow reduce them
    (nulls, slices with a non-zero offset), against pyarrow.compute; the arithmetic type matrix
    (arithmetic_dispatch, src/kernels/routing/arithmetic.rs:280-339) has no arm for them: UnsupportedType."""
    rng = np.random.default_rng(ord(code))
    n = 50_017
    np_dt = pa_type.to_pandas_dtype()
    info = np.iinfo(np_dt)
    vals = rng.integers(info.min, info.max, size=n, endpoint=True).astype(np_dt)
    arr = pa.array(vals, type=pa_type, mask=rng.random(n) < 0.1)
    for sl in (arr, arr.slice(3, 40_000), arr.slice(64 * 7 + 5, 4099), pa.array(vals, type=pa_type)):
        with Exported(sl) as ex:
            assert ex.schema.format == code.encode() if isinstance(ex.schema.format, bytes) else True
            f, i, c = ctx.sum_arrow(ex.array_ptr, ex.schema_ptr)
            m, c2 = ctx.mean_arrow(ex.array_ptr, ex.schema_ptr)
        want = pc.sum(sl.cast(pa.int64())).as_py() or 0
        assert c == c2 == len(sl) - sl.null_count and i == want and f == float(want)
        assert m == float(want) / c
    with Exported(arr) as l, Exported(arr) as r:
        out = ctx.alloc(n * 8)
        with pytest.raises(ffi.MinarrowHipError) as e:
            ctx.apply_arrow(0, (l.array_ptr, l.schema_ptr), (r.array_ptr, r.schema_ptr), out, None)
        assert e.value.status == ffi.MA_ERR_UNSUPPORTED
