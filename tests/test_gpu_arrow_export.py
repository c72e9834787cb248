"""GPU tests of the producer side of the Arrow C Data boundary: ma_apply_arrow_export and
ma_apply_arrow_batch_export hand back library-owned ArrowArray / ArrowSchema pairs (the contract of
create_arrow_export, src/ffi/arrow_c_ffi.rs:1742-1821) that PyArrow imports with `_import_from_c` — the consumer
the reference's own Python bridge talks to (pyo3/src/ffi/to_py.rs). Table semantics follow
broadcast_table_with_operator (src/kernels/broadcast/table.rs:31-63)."""
import ctypes as C

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pytest

from minarrow_amd import ffi
from minarrow_amd.arrow_c import Exported

pytestmark = pytest.mark.gpu

PA_TYPE = {"i": pa.int32(), "l": pa.int64(), "I": pa.uint32(), "L": pa.uint64(), "f": pa.float32(), "g": pa.float64()}
OPS = {"add": 0, "subtract": 1, "multiply": 2, "divide": 3}


def export_apply(ctx, op, lhs, rhs, name=None):
    with Exported(lhs) as a, Exported(rhs) as b:
        return ctx.apply_arrow_export(OPS[op], (a.array_ptr, a.schema_ptr), (b.array_ptr, b.schema_ptr), name)


@pytest.mark.parametrize("fmt", list(PA_TYPE))
def test_exported_array_is_importable_and_equal(ctx, fmt):
    rng = np.random.default_rng(ord(fmt) + 1)
    n = 33_333
    np_dt = PA_TYPE[fmt].to_pandas_dtype()
    a_vals = rng.integers(1, 1000, size=n).astype(np_dt)
    b_vals = rng.integers(1, 1000, size=n).astype(np_dt)
    a_null, b_null = rng.random(n) < 0.1, rng.random(n) < 0.2
    dense = (pa.array(a_vals, type=PA_TYPE[fmt]), pa.array(b_vals, type=PA_TYPE[fmt]))
    nulls = (pa.array(a_vals, type=PA_TYPE[fmt], mask=a_null), pa.array(b_vals, type=PA_TYPE[fmt], mask=b_null))
    sliced = (nulls[0].slice(5, 20_000), nulls[1].slice(77, 20_000))
    for lhs, rhs in (dense, nulls, sliced):
        for op, fn in (("add", pc.add), ("subtract", pc.subtract), ("multiply", pc.multiply)):
            if fmt in "IL" and op == "subtract":
                continue  # pyarrow's unchecked unsigned subtract wraps too, but keep the cross-check simple
            owned = export_apply(ctx, op, lhs, rhs, name="result")
            # the struct itself, before a consumer takes it (create_arrow_export's field values)
            assert owned.array.length == len(lhs) and owned.array.offset == 0 and owned.array.n_buffers == 2
            assert owned.array.n_children == 0 and not owned.array.dictionary
            assert owned.schema.format == fmt.encode() and owned.schema.name == b"result"
            want = fn(lhs, rhs)
            has_validity = bool(owned.array.buffers[0])
            assert owned.array.null_count == (-1 if has_validity else 0)  # create_arrow_export, arrow_c_ffi.rs:1750
            assert has_validity == (lhs.null_count + rhs.null_count > 0)
            assert owned.schema.flags == (2 if has_validity else 0)
            for k in range(2):
                if owned.array.buffers[k]:
                    assert owned.array.buffers[k] % 64 == 0  # check_alignment, arrow_c_ffi.rs:1722-1738
            got = owned.to_pyarrow()  # ownership moves to PyArrow
            assert owned.released
            assert got.type == PA_TYPE[fmt] and got.null_count == want.null_count
            assert got.equals(want)
            # null slots hold 0 in the values buffer (simd.rs:315)
            raw = np.frombuffer(got.buffers()[1], dtype=np_dt, count=len(got))
            assert np.all(raw[~np.asarray(got.is_valid())] == 0)
            del got, raw  # PyArrow calls release -> ma_free_pinned


def test_release_callback_frees_and_marks_released(ctx):
    a = pa.array(np.arange(1000, dtype=np.int64))
    owned = export_apply(ctx, "add", a, a)
    assert owned.array.release and owned.schema.release and owned.array.private_data
    vals = np.ctypeslib.as_array(C.cast(owned.array.buffers[1], C.POINTER(C.c_int64)), shape=(1000,)).copy()
    np.testing.assert_array_equal(vals, 2 * np.arange(1000))
    owned.close()
    assert owned.released and not owned.array.private_data
    owned.close()  # a second close is a no-op (release == NULL)


def test_export_name_defaults_to_left_field_and_broadcast(ctx):
    lhs = pa.array(np.arange(10, dtype=np.float64))
    one = pa.array([2.5])
    with Exported(lhs) as a, Exported(one) as b:
        owned = ctx.apply_arrow_export(OPS["multiply"], (a.array_ptr, a.schema_ptr), (b.array_ptr, b.schema_ptr))
        assert owned.schema.name == (a.schema.name or b"")
    got = owned.to_pyarrow()
    assert got.equals(pa.array(np.arange(10) * 2.5))
    # Int32 (op) Float64 -> Float64 (routing/arithmetic.rs:342-373)
    got = export_apply(ctx, "add", pa.array([1, 2, 3], type=pa.int32()), pa.array([0.5, 0.25, 0.125])).to_pyarrow()
    assert got.type == pa.float64() and got.to_pylist() == [1.5, 2.25, 3.125]
    # empty arrays
    got = export_apply(ctx, "add", pa.array([], type=pa.int64()), pa.array([], type=pa.int64())).to_pyarrow()
    assert len(got) == 0 and got.type == pa.int64()


def test_export_errors_leave_nothing_allocated(ctx):
    a = pa.array([1, 2, 3], type=pa.int64())
    for rhs, status in ((pa.array([1, 2], type=pa.int64()), ffi.MA_ERR_LENGTH_MISMATCH),
                        (pa.array([1.0, 2.0, 3.0]), ffi.MA_ERR_UNSUPPORTED),
                        (pa.array(["x", "y", "z"]), ffi.MA_ERR_UNSUPPORTED)):
        with pytest.raises(ffi.MinarrowHipError) as e:
            export_apply(ctx, "add", a, rhs)
        assert e.value.status == status
    with pytest.raises(ffi.MinarrowHipError) as e:
        export_apply(ctx, "divide", a, pa.array([1, 0, 3], type=pa.int64()))
    assert e.value.status == ffi.MA_ERR_DIVIDE_BY_ZERO


def make_batch(rng, n, with_nulls):
    cols = {
        "ints": pa.array(rng.integers(-1000, 1000, size=n), type=pa.int64(), mask=(rng.random(n) < 0.1) if with_nulls else None),
        "small": pa.array(rng.integers(-1000, 1000, size=n).astype(np.int32), type=pa.int32()),
        "floats": pa.array(rng.standard_normal(n), type=pa.float64(), mask=(rng.random(n) < 0.3) if with_nulls else None),
        "f32": pa.array(rng.standard_normal(n).astype(np.float32), type=pa.float32()),
    }
    return pa.RecordBatch.from_pydict(cols)


@pytest.mark.parametrize("with_nulls", [False, True])
def test_table_op_table_over_record_batches(ctx, with_nulls):
    rng = np.random.default_rng(17)
    n = 50_001
    lhs, rhs = make_batch(rng, n, with_nulls), make_batch(rng, n, with_nulls)
    rhs = rhs.rename_columns(["r0", "r1", "r2", "r3"])
    for op, fn in (("add", pc.add), ("multiply", pc.multiply), ("subtract", pc.subtract)):
        with Exported(lhs) as a, Exported(rhs) as b:
            owned = ctx.apply_arrow_batch_export(OPS[op], (a.array_ptr, a.schema_ptr), (b.array_ptr, b.schema_ptr))
        assert owned.schema.format == b"+s" and owned.array.n_children == 4 and owned.array.length == n
        got = owned.to_pyarrow(record_batch=True)
        assert owned.released
        assert got.schema.names == lhs.schema.names  # left table's field names (table.rs:55-57)
        for c in range(4):
            want = fn(lhs.column(c), rhs.column(c))
            assert got.column(c).type == want.type
            assert got.column(c).equals(want), (op, c)


def test_table_slices_and_mixed_promotions(ctx):
    rng = np.random.default_rng(3)
    n = 10_000
    lhs = make_batch(rng, n, True).slice(13, 7000)  # children carry offset 13
    rhs = pa.RecordBatch.from_pydict({
        "a": pa.array(rng.integers(1, 9, size=n), type=pa.int64()),
        "b": pa.array(rng.standard_normal(n), type=pa.float64()),  # Int32 (op) Float64 -> Float64
        "c": pa.array(rng.standard_normal(n), type=pa.float64()),
        "d": pa.array(rng.integers(1, 9, size=n).astype(np.int32), type=pa.int32()),  # Float32 (op) Int32 -> Float32
    }).slice(200, 7000)
    with Exported(lhs) as a, Exported(rhs) as b:
        owned = ctx.apply_arrow_batch_export(OPS["add"], (a.array_ptr, a.schema_ptr), (b.array_ptr, b.schema_ptr))
    got = owned.to_pyarrow(record_batch=True)
    assert got.num_rows == 7000
    assert [str(t) for t in got.schema.types] == ["int64", "double", "double", "float"]
    assert got.column(0).equals(pc.add(lhs.column(0), rhs.column(0)))
    assert got.column(1).equals(pc.add(pc.cast(lhs.column(1), pa.float64()), rhs.column(1)))
    assert got.column(2).equals(pc.add(lhs.column(2), rhs.column(2)))
    assert got.column(3).equals(pc.add(lhs.column(3), pc.cast(rhs.column(3), pa.float32())))


def test_table_errors(ctx):
    rng = np.random.default_rng(4)
    lhs = make_batch(rng, 100, False)
    fewer = lhs.select([0, 1])
    with Exported(lhs) as a, Exported(fewer) as b:
        with pytest.raises(ffi.MinarrowHipError) as e:
            ctx.apply_arrow_batch_export(0, (a.array_ptr, a.schema_ptr), (b.array_ptr, b.schema_ptr))
    assert e.value.status == ffi.MA_ERR_LENGTH_MISMATCH and "Table column count mismatch: 4 vs 2" in e.value.message
    # a column pair outside the type matrix fails the whole call and frees the columns already produced
    bad = pa.RecordBatch.from_pydict({"ints": lhs.column(0), "small": lhs.column(1), "floats": lhs.column(0), "f32": lhs.column(3)})
    with Exported(lhs) as a, Exported(bad) as b:
        with pytest.raises(ffi.MinarrowHipError) as e:
            ctx.apply_arrow_batch_export(0, (a.array_ptr, a.schema_ptr), (b.array_ptr, b.schema_ptr))
    assert e.value.status == ffi.MA_ERR_UNSUPPORTED
    # not a struct array
    arr = pa.array([1, 2, 3], type=pa.int64())
    with Exported(arr) as a, Exported(arr) as b:
        with pytest.raises(ffi.MinarrowHipError) as e:
            ctx.apply_arrow_batch_export(0, (a.array_ptr, a.schema_ptr), (b.array_ptr, b.schema_ptr))
    assert e.value.status == ffi.MA_ERR_UNSUPPORTED
    # row-count mismatch inside a column pair
    with Exported(lhs) as a, Exported(lhs.slice(0, 50)) as b:
        with pytest.raises(ffi.MinarrowHipError) as e:
            ctx.apply_arrow_batch_export(0, (a.array_ptr, a.schema_ptr), (b.array_ptr, b.schema_ptr))
    assert e.value.status == ffi.MA_ERR_LENGTH_MISMATCH


def test_wide_table_shares_one_pinned_slab_and_columns_outlive_the_batch(ctx):
    """300 columns in one call: one pinned allocation, every kernel enqueued back to back, one synchronise. A column
    kept alive after the batch is gone still reads its values (each array holds its own reference on the slab)."""
    import gc
    import time

    rng = np.random.default_rng(30)
    n, k = 1000, 300
    L = pa.RecordBatch.from_pydict({f"c{i}": pa.array(rng.integers(0, 100, size=n), type=pa.int64()) for i in range(k)})
    R = pa.RecordBatch.from_pydict({f"c{i}": pa.array(rng.integers(0, 100, size=n), type=pa.int64(),
                                                       mask=(rng.random(n) < 0.1) if i % 2 else None) for i in range(k)})
    with Exported(L) as a, Exported(R) as b:
        t0 = time.perf_counter()
        owned = ctx.apply_arrow_batch_export(OPS["add"], (a.array_ptr, a.schema_ptr), (b.array_ptr, b.schema_ptr))
        dt = time.perf_counter() - t0
    assert dt < 0.5  # was ~0.1 s of hipHostMalloc alone with two pinned allocations per column
    # all value buffers come from one allocation: consecutive, 64-byte aligned offsets
    ptrs = [owned.array.children[i].contents.buffers[1] for i in range(k)]
    assert all(p % 64 == 0 for p in ptrs) and max(ptrs) - min(ptrs) < k * (n * 8 + 4096)
    got = owned.to_pyarrow(record_batch=True)
    keep = got.column(7)
    want7 = pc.add(L.column(7), R.column(7))
    for i in (0, 1, 150, 299):
        assert got.column(i).equals(pc.add(L.column(i), R.column(i)))
    del got, owned
    gc.collect()
    assert keep.equals(want7) and keep.null_count == want7.null_count


def test_large_host_resident_arrow_arrays_cross_in_tiles(ctx):
    """PyArrow arrays live in pageable host memory: at 2^24 rows (128 MiB per f64 operand) the values cross PCIe through
    the staging ring, the result lands in the export's pinned slab — same bits and null handling as the small cases."""
    rng = np.random.default_rng(12)
    n = (1 << 24) + 12_345
    a_vals, b_vals = rng.standard_normal(n), rng.standard_normal(n)
    a_null = rng.random(n) < 0.1
    lhs = pa.array(a_vals, mask=a_null).slice(3, n - 7)   # non-zero offset: values and validity windows both shift
    rhs = pa.array(b_vals).slice(3, n - 7)
    owned = export_apply(ctx, "multiply", lhs, rhs)
    got = owned.to_pyarrow()
    want = pc.multiply(lhs, rhs)
    assert got.null_count == want.null_count and got.equals(want)
    raw = np.frombuffer(got.buffers()[1], dtype=np.float64, count=len(got))
    assert np.all(raw[~np.asarray(got.is_valid())] == 0)
    # sums of the same arrays through the Arrow entry point (tiled reduction of a pageable column)
    with Exported(lhs) as a:
        s, _, c = ctx.sum_arrow(a.array_ptr, a.schema_ptr)
    import math
    exact = math.fsum(a_vals[3:3 + n - 7][~a_null[3:3 + n - 7]].tolist())
    assert c == int((~a_null[3:3 + n - 7]).sum()) and abs(s - exact) <= math.ulp(exact)
