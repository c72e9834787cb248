"""GPU parity for the fused i32 <-> f32/f64 promotion (arithmetic_dispatch's promote_to_float64!/32! arms,
src/kernels/routing/arithmetic.rs:244-269, 342-373): bit-identical to casting both sides and calling apply_float_*
— which is what the reference does and what the oracle computes here. Also through ma_apply_arrow, the
resolve_binary_arithmetic equivalent (type matrix + length-1 broadcast)."""
import json
from pathlib import Path

import numpy as np
import pyarrow as pa
import pytest

from minarrow_amd import ffi
from minarrow_amd.arrow_c import Exported

pytestmark = pytest.mark.gpu

OPS = {"add": 0, "subtract": 1, "multiply": 2, "divide": 3, "remainder": 4, "floordiv": 6}
ROUTE = json.loads((Path(__file__).resolve().parent / "golden" / "routing_kat.json").read_text())


def float_equal(got, want):
    nan = np.isnan(want)
    np.testing.assert_array_equal(np.isnan(got), nan)
    ui = np.uint32 if got.dtype == np.float32 else np.uint64
    np.testing.assert_array_equal(got.view(ui)[~nan], want.view(ui)[~nan])


@pytest.mark.parametrize("fl", [np.float64, np.float32])
@pytest.mark.parametrize("int_side", ["lhs", "rhs"])
def test_promote_matches_cast_then_apply_float(ctx, oracle, fl, int_side):
    rng = np.random.default_rng(3)
    ftag = "f64" if fl == np.float64 else "f32"
    for n in (1, 7, 64, 1000, 4097, 70_003):
        ints = rng.integers(-(1 << 31), (1 << 31) - 1, size=n, dtype=np.int32)
        ints[rng.integers(0, n, size=max(1, n // 9))] = 0
        flts = (rng.standard_normal(n) * 1e3).astype(fl)
        flts[rng.integers(0, n, size=max(1, n // 11))] = 0
        bits = rng.integers(0, 256, size=n // 8 + 16, dtype=np.uint8)
        for op in OPS:
            lhs, rhs = (ints, flts) if int_side == "lhs" else (flts, ints)
            st, want, _, _ = oracle.apply_float(oracle.aligned_copy(lhs.astype(fl)), oracle.aligned_copy(rhs.astype(fl)), op)
            ltag, rtag = ("i32", ftag) if int_side == "lhs" else (ftag, "i32")
            for shift in (0, 1):  # shift 1: operands lose their common vector phase -> element-wise path
                dl = ctx.to_device(np.concatenate([np.zeros(shift, lhs.dtype), lhs]), 64)
                dr = ctx.to_device(rhs, 64)
                out = ctx.alloc(n * np.dtype(fl).itemsize + 64)
                ctx.apply_promote(ltag, rtag, dl.offset(shift * lhs.itemsize), dr, OPS[op], out, n, n)
                float_equal(out.download(fl, n), want)
            # masked
            window = oracle.pad_bits(np.packbits(np.unpackbits(bits, bitorder="little")[5:5 + n], bitorder="little"), n)
            st, wantm, want_mask = oracle.float_body("masked_std", lhs.astype(fl), rhs.astype(fl), op, mask=window)
            out = ctx.alloc(n * np.dtype(fl).itemsize + 64)
            om = ctx.alloc(((n + 63) // 64) * 8 + 8)
            ctx.apply_promote(ltag, rtag, ctx.to_device(lhs, 64), ctx.to_device(rhs, 64), OPS[op], out, n, n,
                              mask=ctx.to_device(bits, 16), mask_bit_offset=5, out_mask=om)
            float_equal(out.download(fl, n), wantm)
            np.testing.assert_array_equal(om.download(np.uint8, ((n + 63) // 64) * 8), want_mask[:((n + 63) // 64) * 8])
    with pytest.raises(ffi.MinarrowHipError) as e:
        ctx.apply_promote("i32", ftag, np.zeros(3, np.int32), np.zeros(2, fl), 0, np.zeros(3, fl), 3, 2)
    assert e.value.status == ffi.MA_ERR_LENGTH_MISMATCH


def run_arrow(ctx, op, lhs, rhs, out_dtype):
    n = max(len(lhs), len(rhs))
    out = np.zeros(n, dtype=out_dtype)
    validity = np.zeros(((n + 63) // 64) * 8 + 8, dtype=np.uint8)
    with Exported(lhs) as a, Exported(rhs) as b:
        has = ctx.apply_arrow(OPS[op], (a.array_ptr, a.schema_ptr), (b.array_ptr, b.schema_ptr), out, validity)
    return out, has


def test_arrow_routing_type_matrix(ctx):
    """test_binary_map_type_cast — src/kernels/routing/binary_map.rs:120-135: [1,2,3] (i32) + [10,20,30] (f64)."""
    c = [x for x in ROUTE["binary_map_f64"]["cases"] if "lhs_i32" in x][0]
    out, has = run_arrow(ctx, c["op"], pa.array(c["lhs_i32"], type=pa.int32()), pa.array(c["rhs"], type=pa.float64()), np.float64)
    np.testing.assert_array_equal(out, c["expect"])
    assert not has
    i = pa.array([1, 2, 3, 4], type=pa.int32())
    f64 = pa.array([0.5, 2.0, -3.0, 4.0], type=pa.float64())
    f32 = pa.array([0.5, 2.0, -3.0, 4.0], type=pa.float32())
    np.testing.assert_array_equal(run_arrow(ctx, "multiply", f64, i, np.float64)[0], [0.5, 4.0, -9.0, 16.0])
    np.testing.assert_array_equal(run_arrow(ctx, "subtract", i, f32, np.float32)[0], np.array([0.5, 0.0, 6.0, 0.0], np.float32))
    np.testing.assert_array_equal(run_arrow(ctx, "divide", f32, i, np.float32)[0], np.array([0.5, 1.0, -1.0, 1.0], np.float32))
    # promotion + length-1 broadcast on either side
    np.testing.assert_array_equal(run_arrow(ctx, "add", i, pa.array([0.25], type=pa.float64()), np.float64)[0], [1.25, 2.25, 3.25, 4.25])
    np.testing.assert_array_equal(run_arrow(ctx, "multiply", pa.array([3], type=pa.int32()), f64, np.float64)[0], [1.5, 6.0, -9.0, 12.0])
    # everything else stays UnsupportedType (routing/arithmetic.rs:403-405)
    for other in (pa.array([1, 2, 3, 4], type=pa.int64()), pa.array([1, 2, 3, 4], type=pa.uint32())):
        with pytest.raises(ffi.MinarrowHipError) as e:
            run_arrow(ctx, "add", other, f64, np.float64)
        assert e.value.status == ffi.MA_ERR_UNSUPPORTED
