#!/usr/bin/env python3
"""Stress test of the single-launch cross-workgroup reduction (partials + ticket + last-workgroup fold): many short
launches at many grid sizes, async back to back, on several contexts from several host threads, every result checked.
A stale read of a partial (missing agent-scope release/acquire) shows up as a wrong sum."""
import sys
import threading
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from minarrow_amd.host import Context, PinnedBuffer  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
errors, counts = [], []


def worker(seed):
    ctx = Context(0)
    rng = np.random.default_rng(seed)
    n_max = 3_000_000
    a = rng.integers(-(1 << 40), 1 << 40, size=n_max, dtype=np.int64)
    f = rng.standard_normal(n_max)
    bits = rng.integers(0, 256, size=n_max // 8 + 64, dtype=np.uint8)
    d, df, m = ctx.to_device(a, 64), ctx.to_device(f, 64), ctx.to_device(bits, 16)
    csum = np.concatenate([[0], np.cumsum(a)])
    slots = PinnedBuffer(64 * 64)
    view = slots.view(np.int64, 512)
    ctx.set_async(True)
    t_end = time.time() + seconds
    done = 0
    while time.time() < t_end and not errors:
        batch = []
        for k in range(64):
            n = int(rng.integers(1, n_max))
            off = int(rng.integers(0, n_max - n + 1)) & ~1  # keep 16-byte alignment of the window
            ctx.set_grid(int(rng.choice([0, 1, 2, 3, 7, 64, 255, 256, 257, 1000, 2048, 4096])))
            ctx.sum_into("i64", d.offset(off * 8), n, out_sum=slots.ptr + k * 64, out_count=slots.ptr + k * 64 + 8)
            batch.append((k, n, off))
        ctx.synchronize()
        for k, n, off in batch:
            want = int(csum[off + n] - csum[off])
            if int(view[k * 8]) != want or int(view[k * 8 + 1]) != n:
                errors.append(f"seed {seed}: n={n} off={off} got {int(view[k * 8])} want {want}")
        done += len(batch)
    ctx.set_async(False)
    ctx.set_grid(0)
    # a few masked + f64 checks through the synchronous path
    for _ in range(50):
        n = int(rng.integers(1, n_max))
        got = ctx.sum("i64", d, n, mask=m, mask_bit_offset=int(rng.integers(0, 200)))
        done += 1
        if got[1] > n:
            errors.append("count > n")
    counts.append(done)
    ctx.close()


def shared_worker(ctx, seed):
    """Round 2: SYNCHRONOUS sums from several threads through ONE context — its lanes (own stream, partials and arrival
    tickets each) and the polled completion word, on dense and masked windows at random sizes and offsets."""
    rng = np.random.default_rng(1000 + seed)
    n_max = 2_000_000
    a = rng.integers(-(1 << 40), 1 << 40, size=n_max, dtype=np.int64)
    bits = rng.integers(0, 256, size=n_max // 8 + 64, dtype=np.uint8)
    valid = np.unpackbits(bits, bitorder="little")[:n_max].astype(bool)
    d, m = ctx.to_device(a, 64), ctx.to_device(bits, 16)
    out = ctx.alloc(20000 * 8 + 64)
    csum = np.concatenate([[0], np.cumsum(a)])
    msum = np.concatenate([[0], np.cumsum(np.where(valid, a, 0))])
    mcnt = np.concatenate([[0], np.cumsum(valid)])
    t_end = time.time() + seconds
    done = 0
    while time.time() < t_end and not errors:
        n = int(rng.integers(1, n_max))
        off = int(rng.integers(0, n_max - n + 1)) & ~1
        got = ctx.sum("i64", d.offset(off * 8), n)
        if got != (int(csum[off + n] - csum[off]), n):
            errors.append(f"shared ctx seed {seed}: dense n={n} off={off} got {got}")
        got = ctx.sum("i64", d.offset(off * 8), n, mask=m, mask_bit_offset=off)
        if got != (int(msum[off + n] - msum[off]), int(mcnt[off + n] - mcnt[off])):
            errors.append(f"shared ctx seed {seed}: masked n={n} off={off} got {got}")
        done += 2
        # bitmap scans (ticketed epilogue, device accumulator zeroed by the last workgroup) and a synchronous elementwise
        # call followed by a download (both end in the stream-stamped, polled wait)
        nb = int(rng.integers(1, n_max))
        lo = off & ~63  # popcount addresses its window by whole words (bitmask/simd.rs): keep the offset on one
        hi = lo + min(nb, n_max - lo)
        got = ctx.popcount_mask(m, lo, hi - lo)
        if got != int(mcnt[hi] - mcnt[lo]):
            errors.append(f"shared ctx seed {seed}: popcount bits [{lo},{hi}) got {got} want {int(mcnt[hi] - mcnt[lo])}")
        k = int(rng.integers(1, 20000))
        ctx.apply("i64", d.offset(off * 8), d.offset(off * 8), 0, out, min(k, n), min(k, n))
        res = out.download(np.int64, min(k, n))
        if not np.array_equal(res, a[off:off + min(k, n)] * 2):
            errors.append(f"shared ctx seed {seed}: a + a n={min(k, n)} off={off} mismatch")
        done += 2
    counts.append(done)


def heartbeat():  # a long silent run looks hung to the GPU pool's watchdog: say something every 30 s
    t0 = time.time()
    while time.time() - t0 < seconds + 30 and not stop_heartbeat.is_set():
        stop_heartbeat.wait(30)
        print(f"[{time.time() - t0:5.0f} s] running, {len(errors)} errors so far", flush=True)


stop_heartbeat = threading.Event()
hb = threading.Thread(target=heartbeat, daemon=True)
hb.start()
shared = Context(0)
threads = [threading.Thread(target=worker, args=(s,)) for s in range(4)]
threads += [threading.Thread(target=shared_worker, args=(shared, s)) for s in range(4)]
[t.start() for t in threads]
[t.join() for t in threads]
stop_heartbeat.set()
shared.close()
print(f"{sum(counts)} reductions checked on 4 private contexts (async, random grids) + 4 threads sharing one context "
      f"(synchronous sums, bitmap scans, elementwise + download: lanes + polled completion), {len(errors)} errors")
for e in errors[:10]:
    print(e)
sys.exit(1 if errors else 0)
