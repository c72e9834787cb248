"""GPU parity tests for ma_scan_lanes_*: back-to-back fused sums on one GPU as a pipeline — consecutive scans on two streams of
the context's device, each started when the one before it has begun to drain. The reference's shape is its hot loop of sums
(benches/hotloop_benchmark_std.rs:109-127: one pass per call over an IntegerArray / FloatArray); every scan's results must be
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
