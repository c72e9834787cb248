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


@pytest.mark.parametrize("with_nulls", [False, True])
def test_small_batches_are_gathered_and_come_back_batch_by_batch(ctx, with_nulls):
    """Batch pairs under 1 MiB per column are gathered into pinned tiles, a tile is one call of the batch operator, and the
    results come back as slices (Arrow `offset`) of the tile's result — same batch boundaries, names, types and values as
    the batch-by-batch form. 1500 ragged pairs (empty ones, sliced ones, every third column-pair without nulls) around two
    large pairs that take the direct path; more rows than one tile holds."""
    rng = np.random.default_rng(21)
    sizes = [int(x) for x in rng.choice([0, 1, 63, 64, 65, 1000, 4097, 8192], size=1500)]
    sizes[200], sizes[900] = 400_000, 300_000  # 3.2 / 2.4 MB per 8-byte column: the direct path
    sizes += [8192] * 150                       # > 2^20 rows in all: several tiles
    L, R = make_batches(rng, [n + 7 for n in sizes], with_nulls), make_batches(rng, [n + 7 for n in sizes], with_nulls, names=("x", "y", "z"))
    L = [b.slice(7, n) if k % 4 == 0 else b.slice(0, n) for k, (b, n) in enumerate(zip(L, sizes))]
    R = [b.slice(3, n) if k % 3 == 0 else b.slice(0, n) for k, (b, n) in enumerate(zip(R, sizes))]
    for op, fn in ((0, pc.add), (2, pc.multiply)):
        reader = run_operator(ctx, op, L, R)
        assert reader.schema.names == ["a", "b", "c"]
        got = list(reader)
        assert [g.num_rows for g in got] == sizes
        for g, l, r in zip(got, L, R):
            for c in range(3):
                assert g.column(c).equals(fn(l.column(c), r.column(c)))
        del got, reader


def test_gathered_batches_keep_error_positions(ctx):
    """An error inside a gathered tile is reported at ITS batch, after every batch in front of it: a dense integer division
    by zero in pair 5 of 12 small pairs (the tile call fails, the held batches are replayed one by one); a row-count
    mismatch in pair 3; one stream ending first after 4 pairs."""
    rng = np.random.default_rng(22)
    mk = lambda vals: pa.RecordBatch.from_pydict({"v": pa.array(vals, type=pa.int64())})  # noqa: E731
    L = [mk(rng.integers(1, 100, size=500)) for _ in range(12)]
    R = [mk(rng.integers(1, 100, size=500)) for _ in range(12)]
    bad = rng.integers(1, 100, size=500)
    bad[77] = 0
    R[5] = mk(bad)
    reader = run_operator(ctx, 3, L, R)  # Divide
    for k in range(5):
        b = reader.read_next_batch()
        assert b.num_rows == 500 and b.column(0).equals(pc.divide(L[k].column(0), R[k].column(0)))
    with pytest.raises(Exception) as e:
        reader.read_next_batch()
    assert "batch 5" in str(e.value)
    del reader
    R2 = list(R)
    R2[5] = R[4]
    R2[3] = mk(rng.integers(1, 100, size=499))
    reader = run_operator(ctx, 0, L, R2)
    for k in range(3):
        assert reader.read_next_batch().column(0).equals(pc.add(L[k].column(0), R2[k].column(0)))
    with pytest.raises(Exception) as e:
        reader.read_next_batch()
    assert "batch 3" in str(e.value)
    del reader
    reader = run_operator(ctx, 0, L[:6], R2[:3] + [R[4]])
    for k in range(4):
        assert reader.read_next_batch().num_rows == 500
    with pytest.raises(Exception) as e:
        reader.read_next_batch()
    assert "SuperTable chunk count mismatch: 5 vs 4" in str(e.value)


def test_gathered_slices_own_their_buffers_and_carry_validity_only_where_a_side_had_it(ctx):
    """Two properties of the slices a gathered tile is handed out as, checked on the raw C structs:
    (1) a batch pair neither side of which had validity comes back WITHOUT a validity buffer (null_count 0) even when a
        later pair of the same tile made the tile's column nullable — what the batch-by-batch form and the reference's mask
        union (src/kernels/broadcast/super_array.rs:224) give;
    (2) a child moved out of its parent stays valid after the parent is released (Arrow C Data Interface: moving children):
        its `buffers` array is its own."""
    from minarrow_amd.arrow_c import ArrowArray

    n = 1000
    dense = lambda v: pa.RecordBatch.from_pydict({"v": pa.array(np.arange(n, dtype=np.int64) + v)})  # noqa: E731
    nulls = pa.RecordBatch.from_pydict({"v": pa.array(np.arange(n, dtype=np.int64), mask=np.arange(n) % 5 == 0)})
    L, R = [dense(1), nulls, dense(7)], [dense(10), dense(20), dense(30)]
    lhs = pa.RecordBatchReader.from_batches(L[0].schema, L)
    rhs = pa.RecordBatchReader.from_batches(R[0].schema, R)
    out = ArrowArrayStream()
    l, r = ExportedStream(lhs), ExportedStream(rhs)
    ctx.apply_arrow_stream_export(0, l.ptr, r.ptr, C.addressof(out))
    get_next = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)(out.get_next)
    release = C.CFUNCTYPE(None, C.c_void_p)
    got = []
    for k in range(3):
        a = ArrowArray()
        assert get_next(C.addressof(out), C.addressof(a)) == 0 and a.release and a.length == n and a.n_children == 1
        got.append(a)
    kids = [g.children[0].contents for g in got]
    assert not kids[0].buffers[0] and kids[0].null_count == 0   # no side had validity: no bitmap, although the tile has one
    assert kids[1].buffers[0] and kids[1].null_count == -1      # the pair with nulls carries the tile's bitmap
    assert not kids[2].buffers[0] and kids[2].null_count == 0
    # move batch 1's child out, release the parent, and only then read through the child
    moved = ArrowArray()
    C.memmove(C.addressof(moved), C.addressof(kids[1]), C.sizeof(ArrowArray))
    kids[1].release = None  # "moved": the parent must not release it again
    release(got[1].release)(C.addressof(got[1]))
    assert moved.n_buffers == 2 and moved.buffers[0] and moved.buffers[1]
    vals = np.ctypeslib.as_array(C.cast(moved.buffers[1], C.POINTER(C.c_int64)), shape=(moved.offset + n,))[moved.offset:]
    np.testing.assert_array_equal(vals[1:5], (np.arange(n) + np.arange(n) + 20)[1:5])
    bits = np.ctypeslib.as_array(C.cast(moved.buffers[0], C.POINTER(C.c_uint8)), shape=((moved.offset + n + 7) // 8,))
    valid = np.unpackbits(bits, bitorder="little")[moved.offset:moved.offset + n].astype(bool)
    np.testing.assert_array_equal(valid, np.arange(n) % 5 != 0)
    release(moved.release)(C.addressof(moved))
    for g in (got[0], got[2]):
        release(g.release)(C.addressof(g))
    end = ArrowArray()
    assert get_next(C.addressof(out), C.addressof(end)) == 0 and not end.release  # end of stream
    release(out.release)(C.addressof(out))


@pytest.mark.parametrize("op", [3, 4, 6])  # Divide, Remainder, FloorDiv
def test_a_dense_zero_divisor_gathered_next_to_a_nullable_batch_still_raises(ctx, op):
    """Integer Div / Rem / FloorDiv: the dense kernel raises on a zero divisor (the reference panics,
    src/kernels/arithmetic/std.rs:53-77), the Bitmask-gated one nulls the row out (std.rs:95-138). A tile column is gated
    as soon as one of its batches has validity — so a DENSE batch with a zero divisor must never share a tile with a nullable
    one, or its error would turn into a quiet, valid 0. Pairs 0-1 dense, 2 nullable (its zero divisor nulls the row), 3 dense,
    4 dense with a zero divisor: batches 0-3 come back as the batch-by-batch form gives them, batch 4 is the error."""
    rng = np.random.default_rng(40 + op)
    n = 600
    mk = lambda vals, mask=None: pa.RecordBatch.from_pydict({"v": pa.array(vals, type=pa.int64(), mask=mask)})  # noqa: E731
    L = [mk(rng.integers(-1000, 1000, size=n)) for _ in range(6)]
    R = [mk(rng.integers(1, 50, size=n)) for _ in range(6)]
    nullable = rng.integers(1, 50, size=n)
    nullable[10] = 0                      # a zero divisor in the nullable batch: that row becomes null, no error
    R[2] = mk(nullable, mask=np.arange(n) % 7 == 3)
    bad = rng.integers(1, 50, size=n)
    bad[123] = 0
    R[4] = mk(bad)                        # dense: the reference panics here
    reader = run_operator(ctx, op, L, R)
    single = lambda l, r: list(run_operator(ctx, op, [l], [r]))[0]  # noqa: E731 — the batch-by-batch form of one pair
    for k in range(4):
        b = reader.read_next_batch()
        assert b.column(0).equals(single(L[k], R[k]).column(0)), k
        if k != 2:
            assert b.column(0).null_count == 0 and b.column(0).buffers()[0] is None
        else:
            valid = np.array(b.column(0).is_valid())
            assert not valid[10] and valid.sum() == n - 1 - int((np.arange(n) % 7 == 3).sum()) + int(10 % 7 == 3)
    with pytest.raises(Exception) as e:
        reader.read_next_batch()
    assert "batch 4" in str(e.value) and "by zero in a dense integer kernel" in str(e.value)
    del reader
