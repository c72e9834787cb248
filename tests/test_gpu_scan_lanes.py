"""GPU parity tests for ma_scan_lanes_*: back-to-back fused sums on one GPU as a pipeline — consecutive scans on two streams of
the context's device, each started when the one before it has begun to drain. The reference's shape is its hot loop of sums
(benches/hotloop_benchmark_avg_std.rs:48-62: ITERATIONS passes, an i64 and an f64 sum each; the pass itself: hotloop_benchmark_std.rs:109-127: one pass per call over an IntegerArray / FloatArray); every scan's results must be
the ones ma_sum_fused gives on one stream: integers and counts bit-exact, f64 within 1 ULP of the exactly rounded sum."""
import math

import numpy as np
import pytest

from minarrow_amd.host import Context, ScanLanes

MA_OP_ADD = 0  # ArithmeticOperator::Add (src/enums/operators.rs)

pytestmark = pytest.mark.gpu

M64 = (1 << 64) - 1


def _check(rec, k, want_sum, want_count, exact):
    w = rec.download(np.uint64, 8, 64 * k)
    assert int(w[0]) == want_sum & M64 and int(w[1]) == want_count == int(w[4]), k
    hi, lo = (float(x) for x in w[2:4].view(np.float64))
    assert abs((hi + lo) - exact) <= math.ulp(exact) if exact else hi + lo == 0.0, k


@pytest.mark.parametrize("is_async", [False, True])
@pytest.mark.parametrize("rows", [0, 1, 4097, 50_000, 3_000_017])
def test_every_scan_of_the_pipeline_gets_its_own_results(oracle, rows, is_async):
    """12 scans over 4 distinct column pairs, dense and Bitmask-gated in turn, each into its own record: enqueue-only in either
    mode of the context, results after ma_scan_lanes_synchronize."""
    with Context(0) as ctx:
        ctx.set_async(is_async)
        rng = np.random.default_rng(rows + 5)
        cols = []
        for _ in range(4):
            a = rng.integers(-(1 << 62), 1 << 62, size=rows, dtype=np.int64)
            f = rng.standard_normal(rows) * 1e6
            bits = rng.integers(0, 256, size=(rows + 77) // 8 + 16, dtype=np.uint8)
            cols.append((a, f, bits, ctx.to_device(a, 64), ctx.to_device(f, 64), ctx.to_device(bits, 16)))
        rec = ctx.alloc(64 * 12)
        with ScanLanes(ctx) as lanes:
            for k in range(12):
                a, f, bits, da, df, dm = cols[k % 4]
                mask = (dm, 13) if k % 3 == 2 else ()
                lanes.sum_fused([("l", da, rows, rec.ptr + 64 * k, *mask), ("g", df, rows, rec.ptr + 64 * k + 16, *mask)])
            lanes.synchronize()
            assert lanes.scans == 12
            for k in range(12):
                a, f, bits, *_ = cols[k % 4]
                if k % 3 == 2:
                    want_s, want_c = oracle.masked_sum(a, bits, 13)
                    valid = np.unpackbits(bits, bitorder="little")[13:13 + rows].astype(bool)
                    exact = math.fsum(f[valid].tolist())
                else:
                    want_s, want_c = oracle.sum_scalar(a), rows
                    exact = math.fsum(f.tolist())
                _check(rec, k, want_s, want_c, exact)


def test_work_the_host_enqueues_itself_is_ordered_with_the_scans(oracle):
    """A kernel on the context's stream that WRITES the column the next scan reads (on either lane) comes first; a kernel that
    overwrites it after ma_scan_lanes_join comes after the scans that read it."""
    rows = 2_000_003
    with Context(0) as ctx:
        ctx.set_async(True)
        rng = np.random.default_rng(9)
        a = rng.integers(-(1 << 40), 1 << 40, size=rows, dtype=np.int64)
        f = rng.standard_normal(rows)
        da, df, dout = ctx.to_device(a, 64), ctx.to_device(f, 64), ctx.alloc(rows * 8 + 64)
        rec = ctx.alloc(64 * 8)
        with ScanLanes(ctx) as lanes:
            for k in range(8):  # out = a + k, then the scan of `out` — alternating lanes, so half the scans run on the second stream
                lanes.join()  # the add below overwrites what the scan before it read
                ctx.apply_scalar("i64", "rhs", da, rows, k, MA_OP_ADD, dout)
                lanes.sum_fused([("l", dout, rows, rec.ptr + 64 * k), ("g", df, rows, rec.ptr + 64 * k + 16)])
            lanes.synchronize()
            base = oracle.sum_scalar(a)
            exact = math.fsum(f.tolist())
            for k in range(8):
                _check(rec, k, base + k * rows, rows, exact)


def test_a_refused_scan_leaves_the_pipeline_as_it_was(ctx):
    from minarrow_amd import ffi

    col = ctx.alloc(4096 * 8)
    ctx.synth_iota("i64", col, 4096, 0)
    rec = ctx.alloc(64 * 3)
    with ScanLanes(ctx) as lanes:
        lanes.sum_fused([("l", col, 4096, rec.ptr)])
        with pytest.raises(ffi.MinarrowHipError):  # a misaligned column: nothing is launched, nothing is stamped
            lanes.sum_fused([("l", col.ptr + 4, 4000, rec.ptr + 64)])
        assert lanes.scans == 1
        lanes.sum_fused([("l", col, 4096, rec.ptr + 128)])  # gated on the FIRST scan's early stamp: must not wait for good
        lanes.synchronize()
        for k in (0, 2):
            w = rec.download(np.uint64, 2, 64 * k)
            assert [int(x) for x in w] == [4096 * 4095 // 2, 4096]


TYPES = {"c": np.int8, "C": np.uint8, "s": np.int16, "S": np.uint16, "i": np.int32, "I": np.uint32, "l": np.int64, "L": np.uint64,
         "f": np.float32, "g": np.float64}


@pytest.mark.parametrize("fmt", list(TYPES))
@pytest.mark.parametrize("rows", [0, 777, 1_000_003])
def test_single_column_scans_of_every_type_through_the_pipeline(ctx, oracle, fmt, rows):
    """ma_scan_lanes_sum: the single-column kernels of ma_<t>_sum on the two lanes in turn — dense and Bitmask-gated scans of
    two distinct columns, every scan into its own {sum, lo, count} words; integers bit-exact (wrapping), floats within 1 ULP of
    the exactly rounded sum of the widened values."""
    dt = TYPES[fmt]
    rng = np.random.default_rng(rows * 31 + ord(fmt))
    cols = []
    for _ in range(2):
        if np.issubdtype(dt, np.integer):
            info = np.iinfo(dt)
            v = rng.integers(info.min, info.max, size=rows, dtype=dt, endpoint=True)
        else:
            v = (rng.standard_normal(rows) * 1e3).astype(dt)
        bits = rng.integers(0, 256, size=(rows + 77) // 8 + 16, dtype=np.uint8)
        cols.append((v, bits, ctx.to_device(v, 64), ctx.to_device(bits, 16)))
    rec = ctx.alloc(32 * 8)
    ctx.dev_memset(rec, 0xFF, 32 * 8)
    with ScanLanes(ctx) as lanes:
        for k in range(8):
            v, bits, dv, dm = cols[k & 1]
            gated = k >= 4
            lanes.sum(fmt, dv, rows, rec.ptr + 32 * k, out_count=rec.ptr + 32 * k + 16, out_lo=(rec.ptr + 32 * k + 8) if fmt in "fg" else None,
                      mask=dm if gated else None, mask_bit_offset=13 if gated else 0)
        lanes.synchronize()
        assert lanes.scans == 8
    w = rec.download(np.uint64, 32)
    for k in range(8):
        v, bits, *_ = cols[k & 1]
        valid = np.unpackbits(bits, bitorder="little")[13:13 + rows].astype(bool) if k >= 4 else np.ones(rows, dtype=bool)
        assert int(w[4 * k + 2]) == int(valid.sum()), (k, fmt)
        if fmt in "fg":
            hi, lo = (float(x) for x in w[4 * k:4 * k + 2].view(np.float64))
            exact = math.fsum(v[valid].astype(np.float64).tolist())
            assert abs((hi + lo) - exact) <= (math.ulp(exact) if exact else 0.0), (k, fmt)
        else:
            want = int(v[valid].astype(object).sum()) if rows else 0
            assert int(w[4 * k]) == want & M64, (k, fmt)


def test_a_pipelined_single_column_sum_refuses_host_resident_operands(ctx):
    from minarrow_amd import ffi

    host = np.arange(4096, dtype=np.int64)
    out = ctx.alloc(64)
    with ScanLanes(ctx) as lanes:
        with pytest.raises(ffi.MinarrowHipError):
            lanes.sum("l", host, 4096, out.ptr)  # a pageable column would have to be staged: the call only enqueues
        dev = ctx.to_device(host, 64)
        with pytest.raises(ffi.MinarrowHipError):
            lanes.sum("l", dev, 4096, np.zeros(1, dtype=np.int64))  # a pageable result
        lanes.sum("l", dev, 4096, out.ptr, out_count=out.ptr + 8)
        lanes.synchronize()
        assert [int(x) for x in out.download(np.uint64, 2)] == [4096 * 4095 // 2, 4096]


def test_two_hundred_gated_scans_while_more_streams_than_hardware_queues_are_busy():
    """The one-GPU case of the queue-aliasing hazard: the runtime maps a process's streams onto 4 hardware queues, so with 8 other
    contexts stepping through their own scans the pipeline's two lanes share queues with them — a gate (a wait across streams)
    then sits in a queue in front of, or behind, somebody else's work. Every gate waits for work enqueued BEFORE it, so the
    pipeline must still drain, under a deadline, with every pass's own results."""
    import threading

    rows = (1 << 22) + 77
    tri = rows * (rows - 1) // 2
    others = [Context(0) for _ in range(8)]
    stop = threading.Event()
    errors = []

    def busy(c, seed):
        try:
            c.set_async(True)
            col = c.alloc(rows * 8)
            out = c.alloc(64)
            c.synth_iota("i64", col, rows, seed)
            while not stop.is_set():
                for _ in range(20):
                    c.sum_into("i64", col, rows, out_sum=out.ptr, out_count=out.ptr + 8)
                c.synchronize()
                got = out.download(np.int64, 2)
                if (int(got[0]), int(got[1])) != (tri + seed * rows, rows):
                    errors.append((seed, got.tolist()))
                    return
        except Exception as e:  # noqa: BLE001
            errors.append((seed, repr(e)))

    threads = [threading.Thread(target=busy, args=(c, k)) for k, c in enumerate(others)]
    try:
        with Context(0) as ctx:
            ctx.set_async(True)
            ints, flts = ctx.alloc(rows * 8), ctx.alloc(rows * 8)
            ctx.synth_iota("i64", ints, rows, 0)
            ctx.synth_iota("f64", flts, rows, 0)
            rec = ctx.alloc(64 * 200)
            ctx.dev_memset(rec, 0, 64 * 200)
            ctx.synchronize()
            [t.start() for t in threads]
            with ScanLanes(ctx) as lanes:
                for k in range(200):
                    lanes.sum_fused([("l", ints, rows, rec.ptr + 64 * k), ("g", flts, rows, rec.ptr + 64 * k + 16)])
                    if k % 50 == 49:
                        lanes.synchronize_for(30_000)
                assert lanes.scans == 200 and lanes.is_broken == 0
            stop.set()
            [t.join(timeout=60) for t in threads]
            assert not any(t.is_alive() for t in threads) and not errors, errors[:3]
            w = rec.download(np.uint64, 8 * 200).reshape(200, 8)
            assert (w[:, 0] == tri).all() and (w[:, 1] == rows).all() and (w[:, 4] == rows).all()
            hi_lo = w[:, 2:4].copy().view(np.float64)
            assert (hi_lo.sum(axis=1) == float(tri)).all()  # < 2^53: exact
    finally:
        stop.set()
        for c in others:
            c.close()


def test_a_gate_nobody_opens_is_an_error_within_the_deadline_not_a_hung_host():
    """ma_scan_lanes_synchronize_for: a scan that never starts (the hold hook: a word nobody writes) keeps every later scan behind
    its early stamp. The wait returns after its deadline with an error that names the lane and the sequence, the gates have been
    released, both streams have run empty — the caller's context, whose stream is lane 0, is usable at once — and the pipeline
    takes no more scans; destroying it does not block; the context takes a new one."""
    import time

    from minarrow_amd import ffi

    rows = (1 << 22) + 5
    tri = rows * (rows - 1) // 2
    with Context(0) as ctx:
        ints, flts = ctx.alloc(rows * 8), ctx.alloc(rows * 8)
        ctx.synth_iota("i64", ints, rows, 0)
        ctx.synth_iota("f64", flts, rows, 0)
        rec = ctx.alloc(64 * 8)
        ctx.dev_memset(rec, 0, 64 * 8)
        for held_scan in (1, 2):  # the held scan on the pipeline's own stream, then on the caller's
            lanes = ScanLanes(ctx)
            cols = lambda k: [("l", ints, rows, rec.ptr + 64 * k), ("g", flts, rows, rec.ptr + 64 * k + 16)]  # noqa: E731
            lanes.sum_fused(cols(0))
            if held_scan == 2:
                lanes.sum_fused(cols(1))
            lanes.synchronize_for(20_000)
            lanes.test_hold_next_scan()
            for k in range(2, 6):
                lanes.sum_fused(cols(k))
            t0 = time.perf_counter()
            with pytest.raises(ffi.MinarrowHipError) as e:
                lanes.synchronize_for(300)
            waited = time.perf_counter() - t0
            text = str(e.value)
            assert e.value.status == ffi.MA_ERR_DEVICE and "did not finish within 300 ms" in text and "lane " in text, text
            assert "early stamp" in text and "of sequence" in text and "have run empty" in text, text
            assert 0.28 < waited < 5.0 and lanes.is_broken == 1, (waited, lanes.is_broken)
            with pytest.raises(ffi.MinarrowHipError) as e2:
                lanes.sum_fused(cols(7))
            assert e2.value.status == ffi.MA_ERR_DEVICE and "destroy it" in str(e2.value)
            with pytest.raises(ffi.MinarrowHipError):
                lanes.synchronize()
            t0 = time.perf_counter()
            lanes.close()
            assert time.perf_counter() - t0 < 2.0
            assert ctx.sum("i64", ints, rows) == (tri, rows)  # the context itself is fine
            with ScanLanes(ctx) as again:  # ... and takes a new pipeline
                again.sum_fused(cols(6))
                again.synchronize_for(20_000)
            _check(rec, 6, tri, rows, float(tri))


def test_the_fault_hooks_are_inert_in_a_process_that_did_not_ask_for_them():
    """MINARROW_HIP_TEST_HOOKS is read when the library is loaded: without it the hold hook refuses, and nothing is held."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    code = ("from minarrow_amd import ffi\nfrom minarrow_amd.host import Context, ScanLanes\n"
            "with Context(0) as c:\n    with ScanLanes(c) as l:\n        st = l.lib.ma_scan_lanes_test_hold_next_scan(l.handle)\n"
            "        print(st, l.lib.ma_test_hooks_enabled())\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=str(Path(__file__).resolve().parent.parent),
                       env={k: v for k, v in os.environ.items() if k != "MINARROW_HIP_TEST_HOOKS"})
    assert r.returncode == 0 and r.stdout.split() == ["3", "0"], r.stdout + r.stderr  # MA_ERR_UNSUPPORTED
