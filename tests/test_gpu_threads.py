"""Thread-safety of the boundary: the reference's kernels are pure and re-entrant and are called from Rayon workers
(src/kernels/arithmetic/mod.rs:29-31); the C ABI must tolerate many host threads — sharing one context (calls
serialise on its lock) or owning one context each (independent streams and scratch)."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def worker(ctx, seed, n, errors, reps=20):
    try:
        rng = np.random.default_rng(seed)
        a = rng.integers(-(1 << 40), 1 << 40, size=n, dtype=np.int64)
        b = rng.integers(1, 1 << 20, size=n, dtype=np.int64)
        da, db = ctx.to_device(a, 64), ctx.to_device(b, 64)
        out = ctx.alloc(n * 8 + 64)
        want_sum = int(a.sum())
        want_mul = a * b
        for _ in range(reps):
            assert ctx.sum("i64", da, n) == (want_sum, n)
            ctx.apply("i64", da, db, 2, out, n, n)
            np.testing.assert_array_equal(out.download(np.int64, n), want_mul)
    except Exception as e:  # noqa: BLE001
        errors.append(repr(e))


def test_many_threads_one_shared_context(ctx):
    errors = []
    threads = [threading.Thread(target=worker, args=(ctx, s, 50_000 + 1000 * s, errors)) for s in range(8)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors


def test_threads_sharing_one_context_overlap():
    """SURVEY §8(b) Threading: a shared context must not serialise its callers. Four threads each make synchronous
    sums of their own 2^20-row column through ONE context. Deterministic part: the context grows lanes (own stream +
    reduction scratch) when calls overlap, and a context created with MINARROW_HIP_LANES=1 never does. Timing part: the
    same four threads finish sooner on the context with lanes than on the one without (0.4-0.7x on MI355X,
    tools/bench_lanes.py; asserted loosely — the box's CPU quota and the GIL are in this figure too). Results are
    checked on every call."""
    import os
    import time

    from minarrow_amd.host import Context

    def make(lanes):
        saved = {k: os.environ.get(k) for k in ("MINARROW_HIP_POLL_US", "MINARROW_HIP_LANES")}
        # the blocking wait is what a serialising context queues its callers behind; the polled wait of synchronous
        # reductions would blur the comparison (both variables are read at context creation)
        os.environ["MINARROW_HIP_POLL_US"] = "0"
        os.environ["MINARROW_HIP_LANES"] = str(lanes)
        try:
            return Context(0)
        finally:
            for k, v in saved.items():
                if v is None:
                    del os.environ[k]
                else:
                    os.environ[k] = v

    n, T, reps = 1 << 20, 4, 300
    want = [n * (n - 1) // 2 + i * n for i in range(T)]
    errors = []
    timings = {}
    for lanes in (4, 1):
        ctx = make(lanes)
        bufs = []
        for i in range(T):
            b = ctx.alloc(n * 8)
            ctx.synth_iota("i64", b, n, i)
            bufs.append(b)

        def work(i, count):
            try:
                for _ in range(count):
                    assert ctx.sum("i64", bufs[i], n) == (want[i], n)
            except Exception as e:  # noqa: BLE001
                errors.append(repr(e))

        def concurrent(count):
            ts = [threading.Thread(target=work, args=(i, count)) for i in range(T)]
            t0 = time.perf_counter()
            [t.start() for t in ts]
            [t.join() for t in ts]
            return time.perf_counter() - t0

        concurrent(30)
        timings[lanes] = min(concurrent(reps) for _ in range(3))
        grown = ctx.lib.ma_ctx_lane_count(ctx.handle)
        if lanes == 1:
            assert grown == 1
        else:
            assert 2 <= grown <= 4, grown  # overlapping calls ran on lanes instead of queueing on the context's lock
        for b in bufs:
            b.free()
        ctx.close()
    assert not errors, errors
    assert timings[4] < 0.95 * timings[1], f"with lanes {timings[4]:.4f} s, without {timings[1]:.4f} s"


def test_one_context_per_thread():
    from minarrow_amd.host import Context

    errors = []
    ctxs = [Context(0) for _ in range(6)]
    threads = [threading.Thread(target=worker, args=(c, 100 + i, 200_003, errors)) for i, c in enumerate(ctxs)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    for c in ctxs:
        c.close()
    assert not errors, errors


def host_worker(ctx, seed, n, errors, reps=4):
    """Host-resident operands: every call runs the tiled staging pipeline (its own helper thread, the context's ring)."""
    try:
        rng = np.random.default_rng(seed)
        a = rng.integers(-(1 << 40), 1 << 40, size=n, dtype=np.int64)
        b = rng.integers(1, 1 << 20, size=n, dtype=np.int64)
        out = np.zeros(n, dtype=np.int64)
        want_sum, want_mul = int(a.sum()), a * b
        for _ in range(reps):
            assert ctx.sum("i64", a, n) == (want_sum, n)
            ctx.apply("i64", a, b, 2, out, n, n)
            np.testing.assert_array_equal(out, want_mul)
    except Exception as e:  # noqa: BLE001
        errors.append(repr(e))


def test_tiled_host_operands_from_many_threads(ctx):
    from minarrow_amd.host import Context

    errors = []
    own = [Context(0) for _ in range(3)]
    for c in own + [ctx]:
        c.set_staging_tile(256 << 10)
    try:
        threads = [threading.Thread(target=host_worker, args=(c, 7 + i, 150_011 + 4096 * i, errors)) for i, c in enumerate(own)]
        threads += [threading.Thread(target=host_worker, args=(ctx, 50 + i, 140_003, errors)) for i in range(3)]  # shared
        [t.start() for t in threads]
        [t.join() for t in threads]
    finally:
        ctx.set_staging_tile(32 << 20)
        for c in own:
            c.close()
    assert not errors, errors
