#!/usr/bin/env python3
"""Throughput of ma_apply_arrow_stream_export (SuperTable (op) SuperTable as a stream operator) on PyArrow record batches:
bytes of input + output per second, for a batch size.   python tools/bench_stream_op.py <rows per batch> <batches>"""
import ctypes as C
import json
import sys
import time
from pathlib import Path

import numpy as np
import pyarrow as pa

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from minarrow_amd.arrow_c import ArrowArrayStream, ExportedStream  # noqa: E402
from minarrow_amd.host import Context  # noqa: E402

rows, nb = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8192, 5000)
ctx = Context(0)
rng = np.random.default_rng(0)
mk = lambda: pa.RecordBatch.from_pydict({"a": pa.array(rng.integers(-1000, 1000, size=rows), type=pa.int64()),  # noqa: E731
                                         "b": pa.array(rng.standard_normal(rows), type=pa.float64(), mask=rng.random(rows) < 0.1)})
L, R = [mk()] * nb, [mk()] * nb
best = 1e9
for _ in range(3):
    lhs = pa.RecordBatchReader.from_batches(L[0].schema, L)
    rhs = pa.RecordBatchReader.from_batches(R[0].schema, R)
    out = ArrowArrayStream()
    l, r = ExportedStream(lhs), ExportedStream(rhs)
    t0 = time.perf_counter()
    ctx.apply_arrow_stream_export(0, l.ptr, r.ptr, C.addressof(out))
    reader = pa.RecordBatchReader._import_from_c(C.addressof(out))
    n = 0
    for b in reader:
        n += b.num_rows
    best = min(best, time.perf_counter() - t0)
    del reader
print(json.dumps({"rows_per_batch": rows, "batches": nb, "ms": round(best * 1e3, 2), "us_per_batch": round(best * 1e6 / nb, 1),
                  "gbps_in_plus_out": round(n * 2 * 8 * 3 / best / 1e9, 2)}))
