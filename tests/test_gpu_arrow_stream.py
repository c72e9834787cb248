"""GPU tests of ma_sum_arrow_stream: a chunked table (the reference's SuperTable) arriving batch by batch over
the Arrow C Stream interface (src/ffi/arrow_c_ffi.rs:160-184, 2104-2260), here produced by PyArrow's
RecordBatchReader. The fold over batches must equal the sum over the consolidated column."""
import math

import numpy as np
import pyarrow as pa
import pytest

from minarrow_amd import ffi
from minarrow_amd.arrow_c import ExportedStream

pytestmark = pytest.mark.gpu


def make_table(rng, lens, with_nulls=True):
    batches, all_i, all_f, keep_i, keep_f = [], [], [], [], []
    for n in lens:
        vi = rng.integers(-(1 << 40), 1 << 40, size=n, dtype=np.int64)
        vf = rng.standard_normal(n) * 1e6
        mi = rng.random(n) < 0.1 if with_nulls else np.zeros(n, dtype=bool)
        mf = rng.random(n) < 0.2 if with_nulls else np.zeros(n, dtype=bool)
        batches.append(pa.record_batch([pa.array(vi, mask=mi), pa.array(vf, mask=mf), pa.array(vi.astype(np.int32))],
                                       names=["id", "val", "small"]))
        all_i.append(vi), all_f.append(vf), keep_i.append(~mi), keep_f.append(~mf)
    return batches, np.concatenate(all_i), np.concatenate(all_f), np.concatenate(keep_i), np.concatenate(keep_f)


@pytest.mark.parametrize("lens", [[5], [1000, 0, 70_001, 64, 3], [100_000] * 12])
def test_record_batch_stream(ctx, lens):
    rng = np.random.default_rng(len(lens))
    batches, vi, vf, ki, kf = make_table(rng, lens)
    schema = batches[0].schema
    with ExportedStream(pa.RecordBatchReader.from_batches(schema, batches)) as s:
        f, i, c, rows, nb = ctx.sum_arrow_stream(s.ptr, 0)
    assert (rows, nb, c) == (sum(lens), len(lens), int(ki.sum()))
    assert i == int(vi[ki].sum()) and f == float(i)
    with ExportedStream(pa.RecordBatchReader.from_batches(schema, batches)) as s:
        f, i, c, rows, nb = ctx.sum_arrow_stream(s.ptr, 1)
    exact = math.fsum(vf[kf].tolist())
    assert c == int(kf.sum()) and abs(f - exact) <= math.ulp(exact)
    with ExportedStream(pa.RecordBatchReader.from_batches(schema, batches)) as s:
        f, i, c, rows, nb = ctx.sum_arrow_stream(s.ptr, 2)  # i32 column without nulls
    assert c == sum(lens) and i == int(vi.astype(np.int32).astype(np.int64).sum())


def test_sliced_batches_keep_their_offsets(ctx):
    rng = np.random.default_rng(9)
    batches, vi, vf, ki, kf = make_table(rng, [5000, 5000])
    sliced = [b.slice(13, 4000) for b in batches]
    with ExportedStream(pa.RecordBatchReader.from_batches(batches[0].schema, sliced)) as s:
        f, i, c, rows, nb = ctx.sum_arrow_stream(s.ptr, 0)
    sel = np.concatenate([np.arange(13, 4013), np.arange(5013, 9013)])
    assert rows == 8000 and c == int(ki[sel].sum()) and i == int(vi[sel][ki[sel]].sum())


def test_stream_errors(ctx):
    batches = [pa.record_batch([pa.array(["a", "b"])], names=["s"])]
    with ExportedStream(pa.RecordBatchReader.from_batches(batches[0].schema, batches)) as s:
        with pytest.raises(ffi.MinarrowHipError) as e:
            ctx.sum_arrow_stream(s.ptr, 0)
        assert e.value.status == ffi.MA_ERR_UNSUPPORTED
    with ExportedStream(pa.RecordBatchReader.from_batches(batches[0].schema, batches)) as s:
        with pytest.raises(ffi.MinarrowHipError) as e:
            ctx.sum_arrow_stream(s.ptr, 3)
        assert e.value.status == ffi.MA_ERR_INVALID_ARGUMENT


def test_thousands_of_small_batches_are_gathered_into_tiles(ctx):
    """A SuperTable rechunked at RechunkStrategy::Auto travels as 8192-row record batches: batches under 4 MiB are gathered
    into pinned 8-MiB tiles (values copied, validity appended bit by bit, batches without a bitmap contributing valid bits
    once a batch of the tile has one) and a tile is one copy + one sum. 2500 batches of ragged lengths — some empty, some
    sliced to odd offsets, a third without nulls —, several tiles' worth of rows, with three large batches (direct path) in
    between; every total equals the sum over the consolidated column."""
    rng = np.random.default_rng(77)
    lens = [int(x) for x in rng.choice([0, 1, 63, 64, 65, 1000, 8191, 8192, 8193], size=2500)]
    for at in (100, 1200, 2499):
        lens[at] = 700_000  # 5.6 MB of i64: the direct path, flushing the tile gathered so far
    batches, all_i, all_f, keep_i, keep_f = [], [], [], [], []
    for k, n in enumerate(lens):
        pad = 13 if k % 5 == 0 else 0  # sliced batches keep their offsets
        vi = rng.integers(-(1 << 40), 1 << 40, size=n + pad, dtype=np.int64)
        vf = rng.standard_normal(n + pad) * 1e6
        mi = rng.random(n + pad) < 0.1 if k % 3 else np.zeros(n + pad, dtype=bool)
        mf = rng.random(n + pad) < 0.2 if k % 3 != 1 else np.zeros(n + pad, dtype=bool)
        b = pa.record_batch([pa.array(vi, mask=mi), pa.array(vf, mask=mf), pa.array(vi.astype(np.int32))], names=["id", "val", "small"])
        batches.append(b.slice(pad, n))
        all_i.append(vi[pad:]), all_f.append(vf[pad:]), keep_i.append(~mi[pad:]), keep_f.append(~mf[pad:])
    vi, vf, ki, kf = np.concatenate(all_i), np.concatenate(all_f), np.concatenate(keep_i), np.concatenate(keep_f)
    schema = batches[0].schema
    with ExportedStream(pa.RecordBatchReader.from_batches(schema, batches)) as s:
        f, i, c, rows, nb = ctx.sum_arrow_stream(s.ptr, 0)
    assert (rows, nb, c) == (sum(lens), len(lens), int(ki.sum()))
    assert i == int(vi[ki].sum())
    with ExportedStream(pa.RecordBatchReader.from_batches(schema, batches)) as s:
        f, i, c, rows, nb = ctx.sum_arrow_stream(s.ptr, 1)
    exact = math.fsum(vf[kf].tolist())
    assert c == int(kf.sum()) and abs(f - exact) <= math.ulp(exact)
    with ExportedStream(pa.RecordBatchReader.from_batches(schema, batches)) as s:
        f, i, c, rows, nb = ctx.sum_arrow_stream(s.ptr, 2)
    assert c == sum(lens) and i == int(vi.astype(np.int32).astype(np.int64).sum())
