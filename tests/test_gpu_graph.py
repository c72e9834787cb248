"""GPU tests of hipGraph capture (ma_ctx_capture_begin / _end, ma_graph_launch): a recorded sequence of ABI calls
replays bit-identically to the eager calls, against refilled buffers, and host-synchronising calls are refused
while recording. The workload shape is the reference's launch-bound hot loop — many small columns
(benches/hotloop_benchmark_avg_simd.rs:205-208: a 1000-row sum repeated 1000 times)."""
import numpy as np
import pytest

from minarrow_amd import ffi

pytestmark = pytest.mark.gpu

ADD, SUB, MUL, DIV = 0, 1, 2, 3


def test_recorded_pipeline_replays_on_new_data(ctx, oracle):
    n = 100_003
    rng = np.random.default_rng(1)
    a, b, o1, o2 = (ctx.alloc(n * 8) for _ in range(4))
    m, om = ctx.alloc(n // 8 + 64), ctx.alloc(n // 8 + 64)
    slot = ctx.alloc(64)
    ctx.synth_validity(m, n, seed=3, null_every=7)
    valid = np.unpackbits(m.download(np.uint8, n // 8 + 8), bitorder="little")[:n].astype(bool)

    ctx.capture_begin()
    ctx.apply("f64", a, b, ADD, o1, n, n)                       # o1 = a + b
    ctx.apply_scalar("f64", "rhs", o1, n, 2.5, MUL, o2, mask=m, out_mask=om)   # o2 = o1 * 2.5 where valid else 0
    ctx.sum_into("f64", o2, n, slot.ptr, slot.ptr + 8, mask=om)                # {sum, count} -> device slot
    g = ctx.capture_end()
    assert g.nodes >= 3
    # nothing ran while recording
    for trial in range(3):
        x, y = rng.standard_normal(n), rng.standard_normal(n)
        a.upload(x)
        b.upload(y)
        ctx.synchronize()
        g.launch()
        want = np.where(valid, (x + y) * 2.5, 0.0)
        np.testing.assert_array_equal(o2.download(np.float64, n), want)
        got_sum = slot.download(np.float64, 1)[0]
        got_cnt = slot.download(np.uint64, 1, 8)[0]
        assert got_cnt == valid.sum()
        exact = oracle.exact_sum(want[valid]) if hasattr(oracle, "exact_sum") else float(np.sum(want[valid], dtype=np.longdouble))
        assert abs(got_sum - exact) <= 4 * np.spacing(abs(exact)) + 1e-9
        # the eager path gives the same bits
        e1, e2 = ctx.alloc(n * 8), ctx.alloc(n * 8)
        eom = ctx.alloc(n // 8 + 64)
        ctx.apply("f64", a, b, ADD, e1, n, n)
        ctx.apply_scalar("f64", "rhs", e1, n, 2.5, MUL, e2, mask=m, out_mask=eom)
        s, c = ctx.sum("f64", e2, n, mask=eom)
        assert s == got_sum and c == got_cnt
    g.destroy()


def test_host_synchronising_calls_are_refused_while_recording(ctx):
    n = 4096
    a, o = ctx.alloc(n * 8), ctx.alloc(n * 8)
    bits = ctx.alloc(n // 8 + 64)
    ctx.synth_iota("i64", a, n, 1)
    host = np.arange(n, dtype=np.int64)
    ctx.capture_begin()
    try:
        for call in (lambda: ctx.apply("i64", host, host, ADD, o, n, n),          # pageable input would need staging
                     lambda: ctx.sum("i64", a, n),                                  # result into host memory
                     lambda: ctx.popcount_mask(bits, 0, n),                         # scan returning a value
                     lambda: ctx.synchronize(),
                     lambda: ctx.alloc(64),
                     lambda: ctx.consolidate_column(8, [a], [n], o)):
            with pytest.raises(ffi.MinarrowHipError) as e:
                call()
            assert e.value.status == ffi.MA_ERR_INVALID_ARGUMENT
        with pytest.raises(ffi.MinarrowHipError):
            ctx.capture_begin()  # no nesting
        ctx.apply_scalar("i64", "rhs", a, n, 3, MUL, o)  # the capture is still intact
    finally:
        g = ctx.capture_end()
    assert g.nodes == 1
    g.launch()
    np.testing.assert_array_equal(o.download(np.int64, n), np.arange(1, n + 1) * 3)
    with pytest.raises(ffi.MinarrowHipError):
        ctx.capture_end()  # nothing in progress


def test_recorded_dense_integer_division_reports_zero_divisors(ctx):
    n = 10_000
    a, b, o = ctx.alloc(n * 4), ctx.alloc(n * 4), ctx.alloc(n * 4)
    ctx.synth_iota("i32", a, n, 100)
    ctx.synth_iota("i32", b, n, 1)
    ctx.capture_begin()
    ctx.apply("i32", a, b, DIV, o, n, n)
    g = ctx.capture_end()
    g.launch()  # no zero: fine
    np.testing.assert_array_equal(o.download(np.int32, n), np.arange(100, 100 + n, dtype=np.int32) // np.arange(1, n + 1, dtype=np.int32))
    ctx.synth_iota("i32", b, n, 0)  # b[0] = 0
    with pytest.raises(ffi.MinarrowHipError) as e:
        g.launch()  # the reference panics here (std.rs:53-77)
    assert e.value.status == ffi.MA_ERR_DIVIDE_BY_ZERO
    ctx.synth_iota("i32", b, n, 2)
    g.launch()  # the latch was cleared
    ctx.set_async(True)
    ctx.synth_iota("i32", b, n, 0)
    g.launch()  # async: reported by the next synchronize
    with pytest.raises(ffi.MinarrowHipError) as e:
        ctx.synchronize()
    assert e.value.status == ffi.MA_ERR_DIVIDE_BY_ZERO
    ctx.set_async(False)


def test_thousand_small_sums_in_one_graph(ctx):
    """The hot-loop shape: 1000 independent 1000-row i64 sums; one graph launch produces all 1000 results."""
    k, n = 1000, 1000
    data = ctx.alloc(k * n * 8)
    out = ctx.alloc(k * 8)
    ctx.synth_iota("i64", data, k * n, 0)
    ctx.capture_begin()
    for i in range(k):
        ctx.sum_into("i64", data.ptr + i * n * 8, n, out.ptr + i * 8)
    g = ctx.capture_end()
    assert g.nodes == k
    g.launch()
    got = out.download(np.int64, k)
    base = np.arange(k, dtype=np.int64) * n
    np.testing.assert_array_equal(got, base * n + n * (n - 1) // 2)
