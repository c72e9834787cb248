#!/usr/bin/env python3
"""Throughput of ma_sum_arrow_stream on a PyArrow RecordBatchReader (host-resident batches -> pinned slots -> GPU)."""
import json
import sys
import time
from pathlib import Path

import numpy as np
import pyarrow as pa

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from minarrow_amd.arrow_c import ExportedStream  # noqa: E402
from minarrow_amd.host import Context  # noqa: E402

rows_per_batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
n_batches = int(sys.argv[2]) if len(sys.argv) > 2 else 32
ctx = Context(0)
rng = np.random.default_rng(0)
vals = rng.standard_normal(rows_per_batch)
mask = rng.random(rows_per_batch) < 0.1
batch = pa.record_batch([pa.array(vals, mask=mask), pa.array(np.arange(rows_per_batch, dtype=np.int64))], names=["v", "id"])
batches = [batch] * n_batches
out = {}
for col, name in ((0, "f64_10pct_nulls"), (1, "i64_dense")):
    best = 1e9
    for _ in range(3):
        with ExportedStream(pa.RecordBatchReader.from_batches(batch.schema, batches)) as s:
            t0 = time.perf_counter()
            f, i, c, rows, nb = ctx.sum_arrow_stream(s.ptr, col)
            best = min(best, time.perf_counter() - t0)
    out[name] = {"ms": best * 1e3, "gbps": rows * 8 / best / 1e9, "grows_per_s": rows / best / 1e9, "rows": rows, "batches": nb}
print(json.dumps(out))
