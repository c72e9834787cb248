"""GPU parity tests for ma_sum_fused: several long 8-byte columns (i64 / u64 / f64, dense or Bitmask-gated) summed in ONE
launch, through the C ABI. Per-column results must equal the single-column entry points' (ma_i64_sum, ma_f64_sum_dd), i.e.
the oracle's restatement of benches/benchmark_parallel_simd.rs:44-98: integers and counts bit-exact, f64 within 1 ULP of the
exactly rounded sum (math.fsum)."""
import math

import numpy as np
import pytest

from minarrow_amd.host import tuning_build

pytestmark = pytest.mark.gpu

M64 = (1 << 64) - 1


def ulps(a: float, b: float) -> float:
    return 0.0 if a == b else abs(a - b) / math.ulp(b)


def unpack(bits, off, n):
    return np.unpackbits(bits, bitorder="little")[off:off + n].astype(bool)


def _records(ctx, n_cols):
    return ctx.alloc(64 * n_cols)


def _read(rec, c):
    w = rec.download(np.uint64, 8, 64 * c)
    return w


@pytest.mark.parametrize("n_i,n_f", [(0, 0), (1, 3), (63, 64), (4096, 4097), (8191, 12345), (100_003, 70_001),
                                      ((1 << 20) + 37, (1 << 20) - 5), (3_000_017, 2_500_000), (5, 3_000_017)])
def test_i64_plus_f64_dense(ctx, oracle, n_i, n_f):
    """The headline step's shape: an i64 column and an f64 column of (possibly) different lengths, one launch."""
    rng = np.random.default_rng(n_i * 7 + n_f)
    a = rng.integers(-(1 << 62), 1 << 62, size=n_i, dtype=np.int64)
    f = rng.standard_normal(n_f) * 1e6
    da, df = ctx.to_device(a, pad_bytes=64), ctx.to_device(f, pad_bytes=64)
    rec = _records(ctx, 1)
    ctx.dev_memset(rec, 0xAB, 64)
    ctx.sum_fused([("l", da, n_i, rec.ptr), ("g", df, n_f, rec.ptr + 16)])
    w = _read(rec, 0)
    assert int(w[0]) == oracle.sum_scalar(a) & M64 and int(w[1]) == n_i
    hi, lo = w[2:4].view(np.float64)
    exact = math.fsum(f.tolist())
    assert int(w[4]) == n_f
    assert ulps(float(hi), exact) <= 1 if exact else hi == 0
    # the pair is the one ma_f64_sum_dd gives for the same launch-independent exact arithmetic: hi + lo rounds to hi
    assert float(hi) + float(lo) == float(hi)
    single_i = ctx.sum("i64", da, n_i)
    assert single_i == (oracle.sum_scalar(a), n_i)


@pytest.mark.parametrize("n", [1, 64, 200, 4097, 9000, 100_003, (1 << 20) + 37])
@pytest.mark.parametrize("bit_off", [0, 3, 64, 77, 130])
def test_masked_columns(ctx, oracle, n, bit_off):
    """Config 5's table step: i64 + f64 columns sharing one validity bitmap, plus a dense u64 column in the same launch."""
    rng = np.random.default_rng(n * 31 + bit_off)
    a = rng.integers(-(1 << 63), (1 << 63) - 1, size=n, dtype=np.int64)
    f = rng.standard_normal(n) * 1e8
    u = rng.integers(0, (1 << 64) - 1, size=n, dtype=np.uint64)
    bits = rng.integers(0, 256, size=(bit_off + n + 7) // 8 + 8, dtype=np.uint8)
    valid = unpack(bits, bit_off, n)
    f[~valid] = np.nan  # null slots may hold anything
    da, df, du = (ctx.to_device(x, pad_bytes=64) for x in (a, f, u))
    m = ctx.to_device(bits, pad_bytes=16)
    rec = _records(ctx, 2)
    ctx.sum_fused([("l", da, n, rec.ptr, m, bit_off), ("g", df, n, rec.ptr + 16, m, bit_off), ("L", du, n, rec.ptr + 64)])
    w0, w1 = _read(rec, 0), _read(rec, 1)
    want_sum, want_cnt = oracle.masked_sum(a, bits, bit_off)
    assert (int(w0[0]), int(w0[1])) == (want_sum & M64, want_cnt)
    exact = math.fsum(f[valid].tolist())
    hi = float(w0[2:3].view(np.float64)[0])
    assert int(w0[4]) == want_cnt and (ulps(hi, exact) <= 1 if exact else hi == 0)
    assert int(w1[0]) == int(u.sum(dtype=np.uint64)) and int(w1[1]) == n
    # equal to the single-column entry points, bit for bit (same accumulators, same kind of fold)
    assert ctx.sum("i64", da, n, mask=m, mask_bit_offset=bit_off) == (want_sum, want_cnt)
    one_hi, one_lo, one_cnt = ctx.sum_dd("f64", df, n, mask=m, mask_bit_offset=bit_off)
    assert one_cnt == want_cnt and ulps(one_hi, exact) <= 1 if exact else True


def test_null_count_zero_takes_the_dense_kernel(ctx):
    n = 300_000
    a = np.arange(1, n + 1, dtype=np.int64)
    none_set = np.zeros(n // 8 + 16, dtype=np.uint8)
    d, m = ctx.to_device(a), ctx.to_device(none_set)
    rec = _records(ctx, 1)
    ctx.sum_fused([("l", d, n, rec.ptr, m, 0, 0)])  # null_count = 0: the bitmap is never read (simd.rs:144 gate)
    w = _read(rec, 0)
    assert (int(w[0]), int(w[1])) == (int(a.sum()), n)
    ctx.sum_fused([("l", d, n, rec.ptr, m, 0, -1)])
    w = _read(rec, 0)
    assert (int(w[0]), int(w[1])) == (0, 0)


def test_four_columns_unaligned_windows_and_async(ctx, oracle):
    """Views that start mid-vector (data + 1 element), four columns, enqueue-only, repeated: every launch the same bits."""
    n = 1_234_567
    rng = np.random.default_rng(5)
    cols = [rng.integers(-(1 << 62), 1 << 62, size=n + 1, dtype=np.int64) for _ in range(2)]
    fl = [rng.standard_normal(n + 1) for _ in range(2)]
    dev = [ctx.to_device(x, pad_bytes=64) for x in cols + fl]
    rec = _records(ctx, 2)
    spec = [("l", dev[0].ptr + 8, n, rec.ptr), ("g", dev[2].ptr + 8, n, rec.ptr + 16),
            ("l", dev[1].ptr, n, rec.ptr + 64), ("g", dev[3].ptr, n, rec.ptr + 64 + 16)]
    ctx.set_async(True)
    seen = set()
    for _ in range(5):
        ctx.sum_fused(spec)
        ctx.synchronize()
        seen.add(rec.download(np.uint64, 16).tobytes())
    ctx.set_async(False)
    assert len(seen) == 1  # bit-reproducible for a fixed launch shape
    w0, w1 = _read(rec, 0), _read(rec, 1)
    assert int(w0[0]) == oracle.sum_scalar(cols[0][1:]) & M64 and int(w0[1]) == n
    assert int(w1[0]) == oracle.sum_scalar(cols[1][:n]) & M64 and int(w1[1]) == n
    assert ulps(float(w0[2:3].view(np.float64)[0]), math.fsum(fl[0][1:].tolist())) <= 1
    assert ulps(float(w1[2:3].view(np.float64)[0]), math.fsum(fl[1][:n].tolist())) <= 1


def test_rejections(ctx):
    from minarrow_amd import ffi

    n = 1000
    a = np.arange(n, dtype=np.int64)
    d = ctx.to_device(a)
    rec = _records(ctx, 1)
    with pytest.raises(ffi.MinarrowHipError) as e:
        ctx.sum_fused([("i", d, n, rec.ptr)])  # 4-byte format
    assert e.value.status == ffi.MA_ERR_UNSUPPORTED
    with pytest.raises(ffi.MinarrowHipError) as e:
        ctx.sum_fused([("l", a, n, rec.ptr)])  # pageable host column: not staged here
    assert e.value.status == ffi.MA_ERR_INVALID_ARGUMENT
    with pytest.raises(ffi.MinarrowHipError) as e:
        ctx.sum_fused([("l", d, n, rec.ptr)] * 5)  # more than MA_FUSED_MAX_COLUMNS
    assert e.value.status == ffi.MA_ERR_INVALID_ARGUMENT
    with pytest.raises(ffi.MinarrowHipError) as e:
        ctx.sum_fused([("l", d.ptr + 4, n - 1, rec.ptr)])  # not element-aligned
    assert e.value.status == ffi.MA_ERR_INVALID_ARGUMENT


@pytest.mark.big
def test_one_billion_rows_both_columns_one_launch(ctx):
    """BASELINE configs[1] at full size as ONE launch: sum(0..10^9) over an i64 and an f64 iota column. Closed forms."""
    n = 1_000_000_000
    ci, cf = ctx.alloc(n * 8), ctx.alloc(n * 8)
    ctx.synth_iota("i64", ci, n, 0)
    ctx.synth_iota("f64", cf, n, 0)
    rec = _records(ctx, 1)
    ctx.sum_fused([("l", ci, n, rec.ptr), ("g", cf, n, rec.ptr + 16)])
    w = _read(rec, 0)
    expect = n * (n - 1) // 2
    assert int(w[0]) == expect and int(w[1]) == n and int(w[4]) == n
    hi = float(w[2:3].view(np.float64)[0])
    assert abs(hi - float(expect)) <= math.ulp(float(expect))
    # the 8-way row-chunk partition of the multi-GPU step, each share as one fused launch: checksum of checksums
    from minarrow_amd.parallel import fold_dd, row_chunks

    tot_i, pairs = 0, []
    recs = _records(ctx, 8)
    for k, (lo, hi_) in enumerate(row_chunks(n, 8)):
        ctx.sum_fused([("l", ci.ptr + lo * 8, hi_ - lo, recs.ptr + 64 * k), ("g", cf.ptr + lo * 8, hi_ - lo, recs.ptr + 64 * k + 16)])
    for k in range(8):
        w = _read(recs, k)
        tot_i = (tot_i + int(w[0])) & M64
        pairs.append(tuple(float(x) for x in w[2:4].view(np.float64)))
    assert tot_i == expect & M64
    assert abs(fold_dd(pairs) - float(expect)) <= math.ulp(float(expect))
    ci.free()
    cf.free()


@pytest.mark.big
def test_one_billion_rows_gated_i64_and_f64_sharing_a_bitmap_one_launch(ctx):
    """The fused Bitmask-gated step at full size (BASELINE configs[3]'s column next to its f64 twin, configs[4]'s per-GPU step
    at 8x its share): a 10^9-row i64 and a 10^9-row f64 column (v[i] = i) sharing ONE validity bitmap with 10 % nulls, both in
    one launch, against the single-column kernels: the i64 sum and both counts bit for bit, the f64 sum within 1 ULP of the
    exact gated sum (the i64 result: same values, no wrap). Also at an unaligned bit offset."""
    n = 1_000_000_000
    ci, cf = ctx.alloc(n * 8), ctx.alloc(n * 8)
    mask = ctx.alloc(n // 8 + 192)
    ctx.synth_iota("i64", ci, n, 0)
    ctx.synth_iota("f64", cf, n, 0)
    ctx.synth_validity(mask, n + 77, seed=0xFEED, null_every=10)
    for off in (0, 77):
        rec = _records(ctx, 1)
        ctx.sum_fused([("l", ci, n, rec.ptr, mask, off), ("g", cf, n, rec.ptr + 16, mask, off)])
        w = _read(rec, 0)
        si, cnt = ctx.sum("i64", ci, n, mask=mask, mask_bit_offset=off)
        hi, lo, fcnt = ctx.sum_dd("f64", cf, n, mask=mask, mask_bit_offset=off)
        assert 0.09 < 1 - cnt / n < 0.11 and cnt == ctx.popcount_mask(mask, off, n)
        assert int(w[0]) == si & M64 and int(w[1]) == cnt == int(w[4]) == fcnt
        fh, fl = (float(x) for x in w[2:4].view(np.float64))
        assert abs((fh + fl) - si) <= math.ulp(float(si)) and abs((hi + lo) - si) <= math.ulp(float(si))
    for b in (ci, cf, mask):
        b.free()


def test_stamped_launch_and_exchange_waiting_on_the_stamp(ctx, oracle):
    """ma_sum_fused_stamped: the launch's final thread stores the caller's value to a signal word BEHIND its results, and
    ma_comm_sum_exchange_overlapped_on_stamp makes the exchange stream wait for that word instead of an event on the scan
    stream. Six steps alternate two record sets; every step's folded results must be that step's."""
    from minarrow_amd.host import Comm

    n = 1_000_003
    rng = np.random.default_rng(41)
    a = rng.integers(-(1 << 62), 1 << 62, size=n, dtype=np.int64)
    f = rng.standard_normal(n) * 1e5
    da, df = ctx.to_device(a, pad_bytes=64), ctx.to_device(f, pad_bytes=64)
    stamp = ctx.stamp_alloc()
    word = np.zeros(1, dtype=np.uint64)

    def read_stamp():
        assert ctx.lib.ma_dev_download(ctx.handle, word.ctypes.data, stamp, 8) == 0
        return int(word[0])

    assert read_stamp() == 0
    rec = _records(ctx, 1)
    call = ctx.prepare_sum_fused([("l", da, n, rec.ptr), ("g", df, n, rec.ptr + 16)], stamp=stamp)
    call(5)
    ctx.synchronize()
    assert read_stamp() == 5
    w = _read(rec, 0)
    assert int(w[0]) == oracle.sum_scalar(a) & M64 and int(w[1]) == n and int(w[4]) == n

    comm = Comm(ctx, Comm.unique_id(), 0, 1)
    stamps = [ctx.stamp_alloc(), ctx.stamp_alloc()]
    try:
        ctx.set_async(True)
        sets = [(ctx.alloc(64), ctx.alloc(64), ctx.alloc(32)) for _ in range(2)]
        wants, seq = [], [0, 0]
        for step in range(6):
            k = step % 2
            loc, gat, fin = sets[k]
            lo, hi = (step * 4099) % 100_000, n - (step * 777) % 50_000
            comm.slot_wait(k)
            seq[k] += 1
            launch = ctx.prepare_sum_fused([("l", da.offset(lo * 8), hi - lo, loc.ptr), ("g", df.offset(lo * 8), hi - lo, loc.ptr + 16)],
                                           stamp=stamps[k])
            launch(seq[k])
            comm.sum_exchange_overlapped_on_stamp(k, stamps[k], seq[k], loc, 1, 1, gat, fin)
            wants.append((k, oracle.sum_scalar(a[lo:hi]) & M64, hi - lo, math.fsum(f[lo:hi].tolist())))
        comm.synchronize()
        ctx.set_async(False)
        ctx.synchronize()
        for k, want_i, want_n, want_f in wants[-2:]:
            r = sets[k][2].download(np.uint64, 4)
            assert int(r[0]) == want_i and int(r[1]) == want_n and int(r[3]) == want_n
            assert ulps(float(r[2:3].view(np.float64)[0]), want_f) <= 1
    finally:
        comm.close()
        for s in stamps + [stamp]:
            ctx.stamp_free(s)


def test_two_contexts_alternate_gated_on_the_early_stamp(ctx, oracle):
    """ma_sum_fused_stamped_early + ma_ctx_wait_value: consecutive independent scans on two contexts of one device, each made to
    wait for the EARLY stamp of the one before (stored while that launch drains: two of its eight ticket shards have arrived) — its
    ramp runs under the previous scan's stragglers. Every scan's results must be its own (distinct columns in turn, dense and gated), the final stamps
    end at their sequences, and the early word never runs ahead of a launch that has not started."""
    from minarrow_amd.host import Context

    other = Context(0)
    lanes = [ctx, other]
    rng = np.random.default_rng(77)
    n = 3_000_017
    cols = []
    for k in range(4):
        a = rng.integers(-(1 << 62), 1 << 62, size=n, dtype=np.int64)
        f = rng.standard_normal(n) * 1e6
        bits = rng.integers(0, 256, size=(n + 77) // 8 + 16, dtype=np.uint8)
        cols.append((a, f, bits, ctx.to_device(a, 64), ctx.to_device(f, 64), ctx.to_device(bits, 16)))
    stamps = [c.stamp_alloc() for c in lanes]
    recs = [_records(ctx, 1) for _ in range(8)]
    for c in lanes:
        c.set_async(True)
    seq = [0, 0]
    try:
        for step in range(8):
            lane, (a, f, bits, da, df, dm) = step & 1, cols[step % 4]
            gated = step >= 4
            if step:
                lanes[lane].wait_value(stamps[lane ^ 1] + 8, seq[lane ^ 1])
            seq[lane] += 1
            mask = (dm, 13) if gated else ()
            lanes[lane].prepare_sum_fused([("l", da, n, recs[step].ptr, *mask), ("g", df, n, recs[step].ptr + 16, *mask)],
                                          stamp=stamps[lane], early=stamps[lane] + 8)(seq[lane])
        for c in lanes:
            c.synchronize()
        word = np.zeros(2, dtype=np.uint64)
        for lane, c in enumerate(lanes):
            assert c.lib.ma_dev_download(c.handle, word.ctypes.data, stamps[lane], 16) == 0
            assert [int(w) for w in word] == [4, 4]  # final and early stamp both at the lane's last sequence
        for step in range(8):
            a, f, bits, *_ = cols[step % 4]
            w = _read(recs[step], 0)
            if step >= 4:
                want_s, want_c = oracle.masked_sum(a, bits, 13)
                valid = np.unpackbits(bits, bitorder="little")[13:13 + n].astype(bool)
                exact = math.fsum(f[valid].tolist())
            else:
                want_s, want_c = oracle.sum_scalar(a), n
                exact = math.fsum(f.tolist())
            assert int(w[0]) == want_s & M64 and int(w[1]) == want_c == int(w[4]), step
            hi, lo = (float(x) for x in w[2:4].view(np.float64))
            assert abs((hi + lo) - exact) <= math.ulp(exact), step
    finally:
        for c in lanes:
            c.set_async(False)
        for c, st in zip(lanes, stamps):
            c.stamp_free(st)
        other.close()


@pytest.mark.parametrize("mode", range(8) if tuning_build() else [5])  # the other triggers are tuning forms
@pytest.mark.parametrize("rows", [50_000, 3_000_017])
def test_every_trigger_of_the_early_stamp_stores_it_once_and_leaves_the_results_alone(ctx, oracle, mode, rows):
    """FusedArgs::early_mode (ma_reduce_fused.hip): the early stamp stored by the first workgroup to finish (0), by the arrival that
    completes 1/4, 1/2, 3/4 of a ticket shard (1-3), or when 1 / 2 / 4 / 6 whole shards have arrived (4-7; 5 is the default) — ctx
    variant bits 19-21 = mode + 1. Whatever the trigger (and on a grid of up to 96 workgroups, where the first workgroup stores it
    whatever was asked for), after the launch both words hold the sequence and the sums are the column's."""
    rng = np.random.default_rng(1000 + mode)
    a = rng.integers(-(1 << 62), 1 << 62, size=rows, dtype=np.int64)
    f = rng.standard_normal(rows) * 1e3
    da, df = ctx.to_device(a, 64), ctx.to_device(f, 64)
    rec = _records(ctx, 1)
    stamp = ctx.stamp_alloc()
    ctx.set_variant((mode + 1) << 19 if tuning_build() else 0)  # 0: the library's trigger (mode 5)
    try:
        word = np.zeros(2, dtype=np.uint64)
        for seq in (1, 2, 3):  # the tickets are ready for the next launch each time
            ctx.prepare_sum_fused([("l", da, rows, rec.ptr), ("g", df, rows, rec.ptr + 16)], stamp=stamp, early=stamp + 8)(seq)
            ctx.synchronize()
            assert ctx.lib.ma_dev_download(ctx.handle, word.ctypes.data, stamp, 16) == 0
            assert [int(x) for x in word] == [seq, seq]
            w = _read(rec, 0)
            assert int(w[0]) == oracle.sum_scalar(a) & M64 and int(w[1]) == rows == int(w[4])
            hi, lo = (float(x) for x in w[2:4].view(np.float64))
            exact = math.fsum(f.tolist())
            assert abs((hi + lo) - exact) <= math.ulp(exact)
    finally:
        ctx.set_variant(0)
        ctx.stamp_free(stamp)
