#!/usr/bin/env python3
"""Threads sharing ONE context: do their synchronous calls overlap? (SURVEY §8(b) Threading; the context's lanes,
ma_common.hpp.) T threads x R synchronous sums of a small column each, against the same T x R calls from one thread.
MINARROW_HIP_LANES=1 restores the round-1 behaviour (every call queues behind the context's lock)."""
import json
import sys
import threading
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from minarrow_amd.host import Context  # noqa: E402


def run(ctx, bufs, n, reps, threads):
    want = [int(np.arange(n, dtype=np.int64).sum()) + i * n for i in range(len(bufs))]

    def work(i):
        for _ in range(reps):
            assert ctx.sum("i64", bufs[i], n) == (want[i], n)

    t0 = time.perf_counter()
    if threads == 1:
        for i in range(len(bufs)):
            work(i)
    else:
        ts = [threading.Thread(target=work, args=(i,)) for i in range(len(bufs))]
        [t.start() for t in ts]
        [t.join() for t in ts]
    return time.perf_counter() - t0


def main():
    ctx = Context(0)
    out = {}
    for n in (4096, 1 << 20, 1 << 24):
        T, reps = 4, (3000 if n <= 4096 else 1000 if n <= (1 << 20) else 300)
        bufs = []
        for i in range(T):
            b = ctx.alloc(n * 8)
            ctx.synth_iota("i64", b, n, i)
            bufs.append(b)
        run(ctx, bufs, n, 20, T)  # lanes exist from here on
        serial = run(ctx, bufs, n, reps, 1)
        conc = run(ctx, bufs, n, reps, T)
        out[str(n)] = {"threads": T, "calls": T * reps, "serial_s": serial, "concurrent_s": conc, "ratio": conc / serial,
                       "us_per_call_serial": serial / (T * reps) * 1e6, "us_per_call_concurrent": conc / (T * reps) * 1e6}
        for b in bufs:
            b.free()
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
