"""GPU tests of ma_apply_arrow_stream_export: SuperTable (op) SuperTable as an ArrowArrayStream operator
(broadcast_super_table_with_operator, src/kernels/broadcast/super_table.rs:37-72, over the record-batch streams of
src/ffi/arrow_c_ffi.rs:2104-2260). Producers and the consumer are PyArrow RecordBatchReaders."""
import ctypes as C

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pytest

from minarrow_amd import ffi
from minarrow_amd.arrow_c import ArrowArrayStream, ExportedStream

pytestmark = pytest.mark.gpu


def make_batches(rng, sizes, with_nulls, names=("a", "b", "c")):
    out = []
    for n in sizes:
        cols = {
            names[0]: pa.array(rng.integers(-1000, 1000, size=n), type=pa.int64(), mask=(rng.random(n) < 0.1) if with_nulls else None),
            names[1]: pa.array(rng.standard_normal(n), type=pa.float64(), mask=(rng.random(n) < 0.2) if with_nulls else None),
            names[2]: pa.array(rng.integers(-50, 50, size=n).astype(np.int32), type=pa.int32()),
        }
        out.append(pa.RecordBatch.from_pydict(cols))
    return out


def run_operator(ctx, op, lhs_batches, rhs_batches):
    lhs = pa.RecordBatchReader.from_batches(lhs_batches[0].schema, lhs_batches)
    rhs = pa.RecordBatchReader.from_batches(rhs_batches[0].schema, rhs_batches)
    out = ArrowArrayStream()
    l, r = ExportedStream(lhs), ExportedStream(rhs)
    ctx.apply_arrow_stream_export(op, l.ptr, r.ptr, C.addressof(out))
    assert not l.stream.release and not r.stream.release  # moved into the operator
    return pa.RecordBatchReader._import_from_c(C.addressof(out))


@pytest.mark.parametrize("with_nulls", [False, True])
def test_stream_operator_matches_per_batch_results(ctx, with_nulls):
    rng = np.random.default_rng(8)
    sizes = [1000, 1, 70_003, 64, 5000]
    L, R = make_batches(rng, sizes, with_nulls), make_batches(rng, sizes, with_nulls, names=("x", "y", "z"))
    for op, fn in ((0, pc.add), (2, pc.multiply), (1, pc.subtract)):
        reader = run_operator(ctx, op, L, R)
        assert reader.schema.names == ["a", "b", "c"]  # left field names (table.rs:55-57)
        assert [str(t) for t in reader.schema.types] == ["int64", "double", "int32"]
        got = list(reader)
        assert len(got) == len(sizes)
        for g, l, r in zip(got, L, R):
            for c in range(3):
                assert g.column(c).equals(fn(l.column(c), r.column(c)))
        del got, reader  # releases the result batches and the operator (which releases both inputs)


def test_stream_operator_promotes_and_reports_schema(ctx):
    rng = np.random.default_rng(9)
    L = [pa.RecordBatch.from_pydict({"v": pa.array(rng.integers(0, 9, size=n).astype(np.int32), type=pa.int32())}) for n in (10, 20)]
    R = [pa.RecordBatch.from_pydict({"w": pa.array(rng.standard_normal(n), type=pa.float64())}) for n in (10, 20)]
    reader = run_operator(ctx, 0, L, R)
    assert str(reader.schema.types[0]) == "double" and reader.schema.names == ["v"]  # Int32 (op) Float64 -> Float64
    for g, l, r in zip(reader, L, R):
        assert g.column(0).equals(pc.add(pc.cast(l.column(0), pa.float64()), r.column(0)))


def test_stream_operator_errors_go_through_the_stream_protocol(ctx):
    rng = np.random.default_rng(10)
    L = make_batches(rng, [100, 100, 100], False)
    # chunk count mismatch: rhs ends first (super_table.rs:46-55)
    reader = run_operator(ctx, 0, L, make_batches(rng, [100, 100], False))
    assert reader.read_next_batch().num_rows == 100 and reader.read_next_batch().num_rows == 100
    with pytest.raises(Exception) as e:
        reader.read_next_batch()
    assert "SuperTable chunk count mismatch" in str(e.value)
    # row count mismatch inside a batch
    reader = run_operator(ctx, 0, L, make_batches(rng, [100, 99, 100], False))
    reader.read_next_batch()
    with pytest.raises(Exception) as e:
        reader.read_next_batch()
    assert "batch 1" in str(e.value)
    # column count mismatch is reported by get_schema
    two = [b.select([0, 1]) for b in make_batches(rng, [100], False)]
    out = ArrowArrayStream()
    l = ExportedStream(pa.RecordBatchReader.from_batches(L[0].schema, L[:1]))
    r = ExportedStream(pa.RecordBatchReader.from_batches(two[0].schema, two))
    ctx.apply_arrow_stream_export(0, l.ptr, r.ptr, C.addressof(out))
    with pytest.raises(Exception) as e:
        pa.RecordBatchReader._import_from_c(C.addressof(out))
    assert "column count mismatch" in str(e.value)
    if out.release:
        C.CFUNCTYPE(None, C.c_void_p)(out.release)(C.addressof(out))
    # argument validation
    with pytest.raises(ffi.MinarrowHipError) as e:
        ctx.apply_arrow_stream_export(0, l.ptr, r.ptr, C.addressof(out))  # inputs already moved / released
    assert e.value.status == ffi.MA_ERR_INVALID_ARGUMENT
